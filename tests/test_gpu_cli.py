"""-m gpu test of dab-radio_amd/host/apps/dabgpu_radio_cli (the basic_radio_app_cli equivalent of SURVEY 8f row N1):
capture file in a raw / wav format -> frame bit files -> FIB / sub-channel byte files, every output file compared
byte for byte with the CPU oracle (tests/stream_model.py on the oracle's own dequantisation of the same capture)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import stream_model as SM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "dab-radio_amd", "host", "apps", "dabgpu_radio_cli")


def run_cli(*args, stdin=None):
    if not os.path.exists(CLI):
        import __graft_entry__ as g
        g.build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([CLI, *[str(a) for a in args]], capture_output=True, env=env, timeout=300, input=stdin)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    return res


@pytest.fixture(scope="module")
def capture(oracle, tmp_path_factory):
    """u8 and PCM16-wav captures of one synthetic ensemble + the oracle's expected outputs for each"""
    import stream_model as SM
    subs = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(200, 52, is_uep=True, uep_index=20)]
    stream, truth = SM.make_ensemble_stream(oracle, 7, subs, seed=77)
    x = np.stack([stream.real, stream.imag], axis=-1).reshape(-1)
    x = x / np.abs(x).max()
    d = tmp_path_factory.mktemp("capture")
    u8 = np.clip(np.rint(x * 127.0 + 127.5), 0, 255).astype(np.uint8)
    (d / "iq.u8").write_bytes(u8.tobytes())
    s16 = np.clip(np.rint(x * 20000.0), -32768, 32767).astype("<i2")
    payload = s16.tobytes()
    body = (b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 2048000, 2048000 * 4, 4, 16)
            + b"LIST" + struct.pack("<I", 4) + b"INFO" + b"data" + struct.pack("<I", len(payload)) + payload)
    (d / "iq.wav").write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)

    def expected(raw, fmt, block):
        iq = oracle.iq_convert(raw, fmt).view(np.complex64)
        model = SM.StreamModel(oracle)
        for k in range(0, iq.size, block):
            model.process(iq[k:k + block])
        return model
    return dict(dir=d, subs=subs, truth=truth, u8=u8, s16=s16.view(np.uint8), expected=expected)


def expected_decode(oracle, frames_bits, subs):
    fibs = bytearray()
    msc = [bytearray() for _ in subs]
    deint = [oracle.Deinterleaver(s.length * 8) for s in subs]
    for bits in frames_bits:
        for g in range(4):
            eb, em, _ = oracle.fic_decode_group(bits[g * 2304:(g + 1) * 2304], SM.mirror_core_model())
            for i in range(3):
                if em & (1 << i):
                    fibs += eb[32 * i:32 * i + 30].tobytes()
        for c in range(4):
            cif = bits[9216 + c * 55296:9216 + (c + 1) * 55296]
            for si, s in enumerate(subs):
                deint[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                lf = deint[si].deinterleave()
                if lf is not None:
                    msc[si] += oracle.msc_decode_logical(s, lf, SM.mirror_core_model())[0].tobytes()
    return bytes(fibs), [bytes(m) for m in msc]


def test_ofdm_configuration_writes_the_oracle_frame_bits(oracle, capture):
    d = capture["dir"]
    model = capture["expected"](capture["u8"], 0, 65536)
    exp = np.concatenate([f["bits"] for f in model.out_frames])
    assert len(model.out_frames) >= 6
    res = run_cli("-i", d / "iq.u8", "--configuration", "ofdm", "--ofdm-enable-output", "--ofdm-output", d / "bits.bin")
    assert (d / "bits.bin").read_bytes() == exp.tobytes()
    assert f"frames_read={len(model.out_frames)} frames_desync={model.frames_desync}".encode() in res.stderr
    # hard bytes, to stdout, from stdin, with a different block size
    model2 = capture["expected"](capture["u8"], 0, 10000)
    exp2 = np.concatenate([f["bits"] for f in model2.out_frames])
    res = run_cli("--configuration", "ofdm", "--ofdm-enable-output", "--ofdm-output-hard-bytes", "--ofdm-block-size", 10000,
                  stdin=capture["u8"].tobytes())
    assert res.stdout == oracle.soft_bits_to_hard_bytes(exp2).tobytes()
    (d / "hard.bin").write_bytes(res.stdout)


def test_dab_configuration_decodes_bit_files(oracle, capture):
    d = capture["dir"]
    subs = capture["subs"]
    if not (d / "bits.bin").exists():
        pytest.skip("needs the ofdm test's output")
    sub_args = ["--radio-subchannel", "0,48,3,A", "--radio-subchannel", "200,52,uep,20"]
    bits = np.frombuffer((d / "bits.bin").read_bytes(), np.int8).reshape(-1, oracle.NB_FRAME_BITS)
    fibs, msc = expected_decode(oracle, bits, subs)
    run_cli("-i", d / "bits.bin", "--configuration", "dab", "--radio-fib-output", d / "fibs.bin", "--radio-msc-output", d / "msc_", *sub_args)
    assert (d / "fibs.bin").read_bytes() == fibs and len(fibs) >= 30 * 12 * (len(bits) - 1)
    for k in range(2):
        assert (d / f"msc_{k}.bin").read_bytes() == msc[k] and len(msc[k]) > 0
    # transmitted payload comes back (logical frames that draw only on CIFs after the acquisition frame)
    pay = capture["truth"]["payload"][0]
    got = np.frombuffer(msc[0], np.uint8).reshape(-1, pay.shape[1])
    assert sum(int(np.array_equal(got[j], pay[j])) for j in range(got.shape[0])) >= got.shape[0] - 4
    # packed hard bytes in: the decoder sees +-127
    hard = np.frombuffer((d / "hard.bin").read_bytes(), np.uint8)
    hb = oracle.hard_bytes_to_soft_bits(hard).reshape(-1, oracle.NB_FRAME_BITS)
    fibs_h, msc_h = expected_decode(oracle, hb, subs)
    run_cli("-i", d / "hard.bin", "--configuration", "dab", "--radio-input-hard-bytes", "--radio-fib-output", d / "fibs_h.bin",
            "--radio-msc-output", d / "msch_", *sub_args)
    assert (d / "fibs_h.bin").read_bytes() == fibs_h
    assert (d / "msch_1.bin").read_bytes() == msc_h[1]


def test_dab_plus_ofdm_from_wav(oracle, capture):
    import dabgpu
    d = capture["dir"]
    model = capture["expected"](capture["s16"], dabgpu.IQ_FORMATS.index("wav_pcm16"), 65536)
    bits = [f["bits"] for f in model.out_frames]
    fibs, msc = expected_decode(oracle, bits, capture["subs"])
    res = run_cli("-i", d / "iq.wav", "--ofdm-input-mode", "wav", "--radio-fib-output", d / "fibs_w.bin", "--radio-msc-output", d / "mscw_",
                  "--radio-subchannel", "0,48,3,A", "--ofdm-enable-output", "--ofdm-output", d / "bits_w.bin")
    assert (d / "bits_w.bin").read_bytes() == np.concatenate(bits).tobytes()
    assert (d / "fibs_w.bin").read_bytes() == fibs
    assert (d / "mscw_0.bin").read_bytes() == msc[0]
    assert f"radio: frames={len(bits)}".encode() in res.stderr


def test_bad_inputs_fail_loudly(capture):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    d = capture["dir"]
    r = subprocess.run([CLI, "-i", str(d / "iq.u8"), "--ofdm-input-mode", "wav"], capture_output=True, env=env, timeout=120)
    assert r.returncode == 1 and b"Failed to parse OFDM IQ file" in r.stderr
    r = subprocess.run([CLI, "-i", str(d / "iq.u8"), "--ofdm-input-mode", "raw_s24l"], capture_output=True, env=env, timeout=120)
    assert r.returncode == 1 and b"Unknown iq file format" in r.stderr
