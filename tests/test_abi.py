"""CPU-side checks of the C ABI library: it loads, exports every symbol include/dabgpu.h declares, its
host-side tables equal the oracle's, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dabgpu():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "dab-radio_amd", "libdabgpu.so")):
        g.build()
    import dabgpu
    return dabgpu


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dabgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dabgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(dabgpu):
    L = dabgpu.lib()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/dabgpu.h but not exported"
    assert sorted(dabgpu.ABI_SYMBOLS) == syms
    assert L.dabgpu_abi_version() == 4


def test_host_tables_match_oracle(dabgpu, oracle):
    prs, mapper, tw = dabgpu.host_tables()
    assert np.array_equal(prs.view(np.uint32), oracle.prs_fft().view(np.uint32))
    assert np.array_equal(mapper, oracle.mapper())
    assert np.array_equal(tw.view(np.uint32), oracle.twiddles().view(np.uint32))


def test_unsupported_mode_and_null_args(dabgpu):
    L = dabgpu.lib()
    buf = np.zeros(4096, np.float32)
    assert L.dabgpu_get_prs_fft_ref(2, buf.ctypes.data) == 0          # modes II-IV have tables too
    assert L.dabgpu_get_prs_fft_ref(5, buf.ctypes.data) == 2          # DABGPU_ERR_INVALID_ARG: no such transmission mode
    assert L.dabgpu_get_prs_fft_ref(1, None) == 2                      # DABGPU_ERR_INVALID_ARG
    assert L.dabgpu_ofdm_demod_frames(None, None, 1, None, None, None, None, None, 0, 0, None) == 2


def test_no_cpu_fallback(dabgpu):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert dabgpu.device_count() == 0
    with pytest.raises(dabgpu.DabGpuError):
        dabgpu.Context(0)
