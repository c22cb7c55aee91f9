// app_helpers/app_io_buffers.h -- the two stream interfaces the reference's example apps are built from
// (examples/app_helpers/app_io_buffers.h:13-23).  A tree that already has the reference's header keeps it.
#pragma once
#include <stddef.h>
#include "utility/span.h"

template <typename T>
struct InputBuffer {
    virtual ~InputBuffer() {}
    virtual size_t read(tcb::span<T> dest) = 0;
};

template <typename T>
struct OutputBuffer {
    virtual ~OutputBuffer() {}
    virtual size_t write(tcb::span<const T> src) = 0;
};
