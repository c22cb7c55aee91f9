#include "./dabgpu_shared_context.h"

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

#include "dabgpu.h"

dabgpu_ctx* dabgpu_shared_context() {
    static std::mutex mu;
    static dabgpu_ctx* ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ctx) {
        const char* dev = std::getenv("DABGPU_DEVICE");
        const int st = dabgpu_create(&ctx, dev ? std::atoi(dev) : 0, nullptr, nullptr);
        if (st != DABGPU_OK)
            throw std::runtime_error(std::string("dabgpu: no usable MI355X context (no CPU fallback): ") + dabgpu_strerror(st) +
                                     " -- " + dabgpu_last_error());
    }
    return ctx;
}

dabgpu_ctx* dabgpu_private_context() {
    (void)dabgpu_shared_context();                  // (fails first, with the same message, when there is no device)
    const char* dev = std::getenv("DABGPU_DEVICE");
    dabgpu_ctx* ctx = nullptr;
    const int st = dabgpu_create(&ctx, dev ? std::atoi(dev) : 0, nullptr, nullptr);
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("dabgpu: cannot create a decoder context: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    return ctx;
}

int dabgpu_tie_rule_from_env() {
    const char* t = std::getenv("DABGPU_TIE_RULE");
    return t ? std::atoi(t) : 0;
}
