"""SURVEY 8b threading: FIC_Decoder / MSC_Decoder objects are single-threaded each, but different objects run concurrently on
BasicThreadPool workers (src/basic_radio/basic_radio.cpp:51-60) -- and the mirror classes share ONE device context
(host/dab/dabgpu_shared_context.cpp).  The host-side entry points therefore serialise on the context; this test hammers one context from
several threads (ctypes releases the GIL during the calls) and compares every result with the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_host_sync_entry_points_from_concurrent_threads(oracle):
    import dabgpu
    L = dabgpu.lib()
    L.dabgpu_fic_decode_group_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_int]
    ctx = dabgpu.Context(0)
    n_threads, n_iter = 6, 40
    rng = np.random.default_rng(3)
    work = []
    for t in range(n_threads):
        items = []
        for k in range(n_iter):
            fibs = rng.integers(0, 256, 90, dtype=np.uint8)
            tx = oracle.fic_encode_group(fibs)                                           # 2304 punctured bits 0/1
            soft = np.clip(np.rint((2.0 * tx - 1.0) * 50.0 + rng.normal(0, 30.0, tx.size)), -127, 127).astype(np.int8)
            items.append((soft, oracle.fic_decode_group(soft, 0)))
        work.append(items)
    # a second kind of caller on the same context: the single-frame demodulator
    bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
    frame = oracle.tx_to_frame_buffer(oracle.apply_pll(oracle.modulate_frame(bits), -1.0e-4, 0.0))
    exp_frame = oracle.demod_frame(frame, 1.0e-4)["bits"]
    errors = []

    def fic_worker(items):
        try:
            for soft, (eb, em, ee) in items:
                out = np.zeros(96, np.uint8)
                mask, err = C.c_uint32(0), C.c_uint64(0)
                st = L.dabgpu_fic_decode_group_host_sync(ctx._h, soft.ctypes.data, out.ctypes.data, C.byref(mask), C.byref(err), 0)
                assert st == 0
                assert np.array_equal(out, eb) and mask.value == em and err.value == ee
        except Exception as e:                                                           # noqa: BLE001
            errors.append(repr(e))

    def demod_worker():
        try:
            for _ in range(6):
                got, _, _ = ctx.ofdm_demod_frames_host(frame, np.array([1.0e-4], np.float32))
                assert np.array_equal(got[0], exp_frame)
        except Exception as e:                                                           # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=fic_worker, args=(w,)) for w in work] + [threading.Thread(target=demod_worker)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
