// utility/observable.h -- callback list with the interface of the reference's Observable<T...>
// (src/utility/observable.h:6-22): Attach(std::function) and Notify(args...), called in attach order.
#pragma once
#include <functional>
#include <utility>
#include <vector>

template <typename... Args>
class Observable {
public:
    using Callback = std::function<void(Args...)>;
    void Attach(const Callback& cb) { m_callbacks.push_back(cb); }
    void Notify(Args... args) {
        for (auto& cb : m_callbacks) cb(args...);
    }
private:
    std::vector<Callback> m_callbacks;
};
