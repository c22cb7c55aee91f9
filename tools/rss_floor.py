#!/usr/bin/env python3
"""Where a receiver's host memory goes: resident set of a process after (a) loading libdabgpu.so, (b) one dabgpu_create (HIP runtime initialised, this
library's code objects loaded, one stream), (c) one receiver (dabgpu_receiver_create: two contexts, a frame session, three staging buffers), (d) a second
receiver -- without torch in the process.  The difference between (b) and the loader's baseline is the ROCm runtime's floor, which no design of this
library changes; (c) - (b) and (d) - (c) are what a receiver itself costs.

    python tools/rss_floor.py        (on the GPU box)
"""
import ctypes as C, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rss_mb():
    return int(re.search(r"VmRSS:\s+(\d+)", open("/proc/self/status").read()).group(1)) / 1024.0


out = {"python_only_MB": round(rss_mb(), 1)}
L = C.CDLL(os.path.join(ROOT, "dab-radio_amd", "libdabgpu.so"))
out["library_loaded_MB"] = round(rss_mb(), 1)
ctx = C.c_void_p()
L.dabgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
assert L.dabgpu_create(C.byref(ctx), 0, None, None) == 0
out["after_dabgpu_create_MB"] = round(rss_mb(), 1)
L.dabgpu_receiver_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rx = [C.c_void_p(), C.c_void_p()]
assert L.dabgpu_receiver_create(C.byref(rx[0]), 0, 1, None, None) == 0
out["after_one_receiver_MB"] = round(rss_mb(), 1)
assert L.dabgpu_receiver_create(C.byref(rx[1]), 0, 1, None, None) == 0
out["after_two_receivers_MB"] = round(rss_mb(), 1)
# members of the receiver bank (dabgpu_receiver_create_banked): no streams or contexts of their own -- the first one pays for the bank
L.dabgpu_receiver_create_banked.argtypes = [C.POINTER(C.c_void_p), C.c_int]
bk = [C.c_void_p() for _ in range(9)]
assert L.dabgpu_receiver_create_banked(C.byref(bk[0]), 0) == 0
out["after_first_banked_receiver_MB"] = round(rss_mb(), 1)
for k in range(1, 9):
    assert L.dabgpu_receiver_create_banked(C.byref(bk[k]), 0) == 0
out["after_nine_banked_receivers_MB"] = round(rss_mb(), 1)
out["per_further_banked_receiver_MB"] = round((out["after_nine_banked_receivers_MB"] - out["after_first_banked_receiver_MB"]) / 8.0, 1)
free = C.c_size_t(); total = C.c_size_t()
hip = C.CDLL("libamdhip64.so")
hip.hipMemGetInfo(C.byref(free), C.byref(total))
out["device_used_MB_at_end"] = round((total.value - free.value) / 2**20, 1)
out["what"] = __doc__.split("\n\n")[0]
print(json.dumps(out))
