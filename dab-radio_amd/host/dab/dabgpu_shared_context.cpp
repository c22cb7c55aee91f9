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

int dabgpu_core_model_from_env() {
    if (const char* c = std::getenv("DABGPU_VITERBI_CORE")) {
        const std::string v = c;
        if (v == "simd" || v == "1") return 1;
        if (v == "scalar" || v == "0") return 0;
        throw std::runtime_error("DABGPU_VITERBI_CORE must be 'scalar' or 'simd', not '" + v + "'");
    }
    if (const char* t = std::getenv("DABGPU_TIE_RULE")) return std::atoi(t) ? 1 : 0;
    // what src/dab/algorithms/dab_viterbi_decoder.cpp:51-73 selects when the reference is built on this host with its default preset
#if defined(__x86_64__) || defined(__i386__)
    return (__builtin_cpu_supports("avx2") || __builtin_cpu_supports("sse4.1")) ? 1 : 0;
#elif defined(__aarch64__)
    return 1;
#else
    return 0;
#endif
}
