"""The reference's own callers must compile against the mirror headers unchanged (`basic_radio` links unchanged).
A scratch COPY of /root/reference/src gets the six mirror headers laid over it, then every caller of the hot path is run through
`g++ -fsyntax-only`: src/basic_radio/basic_radio.cpp (:41-65 FIC/MSC slicing), basic_fic_runner.cpp (:7,20,44-48 FIC_Decoder),
basic_audio_channel.cpp (:6,12 MSC_Decoder), basic_{dab,dab_plus,data_packet}_channel.cpp (DecodeCIF, AAC_Frame_Processor) and
examples/app_helpers/app_ofdm_blocks.h (:25-58 OFDM_Demod).  Skipped where /root/reference is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "dab-radio_amd", "host")
MIRROR_HEADERS = ["ofdm/ofdm_demodulator.h", "dab/fic/fic_decoder.h", "dab/msc/msc_decoder.h", "dab/msc/cif_deinterleaver.h",
                  "dab/algorithms/dab_viterbi_decoder.h", "dab/audio/aac_frame_processor.h"]
CALLERS = ["basic_radio/basic_radio.cpp", "basic_radio/basic_fic_runner.cpp", "basic_radio/basic_audio_channel.cpp",
           "basic_radio/basic_dab_channel.cpp", "basic_radio/basic_dab_plus_channel.cpp", "basic_radio/basic_data_packet_channel.cpp"]


def _fmt_include():
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "include")
        if os.path.exists(os.path.join(p, "fmt", "format.h")):
            return p
    except Exception:
        pass
    return None


@pytest.fixture(scope="module")
def overlay(tmp_path_factory):
    if not os.path.isdir(os.path.join(REF, "src")):
        pytest.skip("/root/reference is not present here")
    fmt = _fmt_include()
    if fmt is None:
        pytest.skip("no header-only fmt available (the reference's vendor/fmt submodule is empty)")
    d = tmp_path_factory.mktemp("ref_overlay")
    src = os.path.join(d, "src")
    shutil.copytree(os.path.join(REF, "src"), src)
    for h in MIRROR_HEADERS:
        shutil.copyfile(os.path.join(HOST, h), os.path.join(src, h))
    return src, fmt


def _syntax_only(args):
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DFMT_HEADER_ONLY"] + args, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]


@pytest.mark.parametrize("caller", CALLERS)
def test_basic_radio_callers_compile_against_the_mirror_headers(overlay, caller):
    src, fmt = overlay
    _syntax_only(["-I" + src, "-I" + os.path.join(ROOT, "include"), "-I" + fmt, os.path.join(src, caller)])


def test_app_ofdm_blocks_compiles_against_the_mirror_demodulator(overlay, tmp_path):
    src, fmt = overlay
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "app_helpers/app_ofdm_blocks.h"\nint main() { return 0; }\n')
    _syntax_only(["-I" + src, "-I" + os.path.join(REF, "examples"), "-I" + os.path.join(ROOT, "include"), "-I" + fmt, str(tu)])
