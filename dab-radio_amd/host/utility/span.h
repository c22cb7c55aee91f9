// utility/span.h -- minimal dynamic-extent span in namespace tcb.
//
// The reference's headers take and return `tcb::span<T>` (its src/utility/span.h is Tristan Brindle's C++20 span
// back-port).  This repository does not redistribute that file; this is an independent, much smaller class that
// offers the subset of the interface the hot-path classes and their callers use.  When the mirror classes are
// dropped into the reference tree, the reference's own utility/span.h is used instead (same names, same semantics).
#pragma once
#include <array>
#include <cstddef>
#include <type_traits>
#include <vector>

namespace tcb {

template <typename T>
class span {
public:
    using element_type = T;
    using value_type = typename std::remove_cv<T>::type;
    using size_type = std::size_t;
    using pointer = T*;
    using reference = T&;
    using iterator = T*;

    constexpr span() noexcept : m_ptr(nullptr), m_len(0) {}
    constexpr span(T* ptr, size_type len) noexcept : m_ptr(ptr), m_len(len) {}
    constexpr span(T* first, T* last) noexcept : m_ptr(first), m_len(static_cast<size_type>(last - first)) {}
    template <std::size_t N>
    constexpr span(T (&arr)[N]) noexcept : m_ptr(arr), m_len(N) {}
    template <typename U, std::size_t N, typename = typename std::enable_if<std::is_convertible<U (*)[], T (*)[]>::value>::type>
    constexpr span(std::array<U, N>& a) noexcept : m_ptr(a.data()), m_len(N) {}
    template <typename U, std::size_t N, typename = typename std::enable_if<std::is_convertible<const U (*)[], T (*)[]>::value>::type>
    constexpr span(const std::array<U, N>& a) noexcept : m_ptr(a.data()), m_len(N) {}
    // any contiguous container with data()/size() whose element pointer converts (std::vector, std::string, ...)
    template <typename C, typename = typename std::enable_if<
                              !std::is_array<C>::value &&
                              std::is_convertible<typename std::remove_pointer<decltype(std::declval<C&>().data())>::type (*)[], T (*)[]>::value>::type>
    constexpr span(C& c) : m_ptr(c.data()), m_len(c.size()) {}
    template <typename C, typename = typename std::enable_if<
                              !std::is_array<C>::value &&
                              std::is_convertible<typename std::remove_pointer<decltype(std::declval<const C&>().data())>::type (*)[], T (*)[]>::value>::type>
    constexpr span(const C& c) : m_ptr(c.data()), m_len(c.size()) {}
    template <typename U, typename = typename std::enable_if<std::is_convertible<U (*)[], T (*)[]>::value>::type>
    constexpr span(const span<U>& o) noexcept : m_ptr(o.data()), m_len(o.size()) {}

    constexpr pointer data() const noexcept { return m_ptr; }
    constexpr size_type size() const noexcept { return m_len; }
    constexpr size_type size_bytes() const noexcept { return m_len * sizeof(T); }
    constexpr bool empty() const noexcept { return m_len == 0; }
    constexpr reference operator[](size_type i) const { return m_ptr[i]; }
    constexpr reference front() const { return m_ptr[0]; }
    constexpr reference back() const { return m_ptr[m_len - 1]; }
    constexpr iterator begin() const noexcept { return m_ptr; }
    constexpr iterator end() const noexcept { return m_ptr + m_len; }

    constexpr span first(size_type n) const { return span(m_ptr, n); }
    constexpr span last(size_type n) const { return span(m_ptr + (m_len - n), n); }
    constexpr span subspan(size_type offset, size_type count = static_cast<size_type>(-1)) const {
        return span(m_ptr + offset, count == static_cast<size_type>(-1) ? m_len - offset : count);
    }

private:
    T* m_ptr;
    size_type m_len;
};

template <typename C>
span(C&) -> span<typename std::remove_pointer<decltype(std::declval<C&>().data())>::type>;
template <typename T>
span(T*, std::size_t) -> span<T>;

}  // namespace tcb
