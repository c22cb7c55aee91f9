"""The FIG-valid synthetic ensemble the reference's callers are executed on (tests/test_reference_callers_run.py, tests/test_gpu_fig_ensemble.py,
tests/golden/make_golden_fig_ensemble.py): one capture, regenerated from its seed wherever it is needed (the capture itself is 6 MB and not
committed; its SHA-256 is, so a fixture can never be compared with another capture).

Multiplex: tools/dabsynth.py::mixed_layout (14 sub-channels: EEP 3-A / 2-B / 2-A, three UEP table rows incl. one with padding bits), described
to the receiver by tools/dabfig.py FIGs: sub-channel 13 (8 CU) is a packet-mode data service with FEC, 9 a stream-mode data component (the
reference creates no decoder for it), 10 is organised in FIG 0/1 but no service refers to it.  The organisation is spread over the first frames
(three sub-channels per frame, their services one frame later) so that the reference creates its decoders at five different frames."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SEED = 2601
N_FRAMES = 17                    # transmitted; the last one's end is not followed by a NULL symbol and does not come out
PACKET_SUB, STREAM_DATA_SUB, ORPHAN_SUB = 13, 9, 10
BLOCK = 65536                    # --ofdm-block-size default of the reference's apps (basic_radio_app.cpp:84)


def layout():
    import dabsynth                                     # (imports torch for its generator half; only the layout tables are used here)
    return dabsynth.mixed_layout()


class _unshifted:
    """the capture of a committed fixture is data: under DAB_FUZZ_OFFSET (tests/conftest.py shifts every integer seed handed to
    numpy.random.default_rng, tools/fuzz_suite.sh) it is regenerated with the plain generator"""
    def __enter__(self):
        self.saved = np.random.default_rng
        np.random.default_rng = getattr(self.saved, "plain", self.saved)
    def __exit__(self, *a):
        np.random.default_rng = self.saved


def description(seed=SEED):
    import dabfig
    with _unshifted():
        return dabfig.describe(layout(), seed=seed, packet_sub=PACKET_SUB, stream_data_sub=STREAM_DATA_SUB, orphan_sub=ORPHAN_SUB)


def make_capture(O, SM, seed=SEED, n_frames=N_FRAMES):
    """-> (raw_u8 capture bytes [2 x samples], truth): truth = dict(desc, carousel, fib_data [n_frames][4][3][30], payload per sub-channel
    [n_cif][nbytes], subs (oracle SubChannel structs in layout order), sha256)"""
    import dabfig
    desc = description(seed)
    car = dabfig.Carousel(desc)
    fib_data = car.frames(n_frames)
    subs = [O.subchannel(s["start"], s["length"], eep_level=s["eep_level"], eep_type=s["eep_type"], is_uep=bool(s["is_uep"]), uep_index=s["uep_index"])
            for s in desc["subchannels"]]
    with _unshifted():
        stream, truth = SM.make_ensemble_stream(O, n_frames, subs, seed=seed, cfo=1.3e-3, noise=2.0, amplitude=1.0, fib_data=fib_data)
    comp = stream.view(np.float32)
    full = float(np.quantile(np.abs(comp[::13]), 0.999))                # an RTL-SDR style 8-bit capture, the strongest components clip
    u8 = np.clip(np.rint(comp / full * 127.5 + 127.5), 0, 255).astype(np.uint8)
    truth.update(desc=desc, carousel=car, fib_data=fib_data, subs=subs, sha256=hashlib.sha256(u8.tobytes()).hexdigest())
    return u8, truth


def read_fibs(path):
    """fibs.bin {u32 frame, 30 bytes} -> list of (frame, bytes)"""
    raw = open(path, "rb").read() if os.path.exists(path) else b""
    return [(int.from_bytes(raw[p:p + 4], "little"), raw[p + 4:p + 34]) for p in range(0, len(raw), 34)]


def read_msc(path):
    """msc_<id>.bin {u32 frame, u32 cif, u32 n, n bytes} -> list of (frame, cif, bytes)"""
    raw = open(path, "rb").read() if os.path.exists(path) else b""
    out, p = [], 0
    while p < len(raw):
        f, c, n = (int.from_bytes(raw[p + 4 * i:p + 4 * i + 4], "little") for i in range(3))
        out.append((f, c, raw[p + 12:p + 12 + n]))
        p += 12 + n
    return out


def read_created(path):
    """created.txt -> list of dict(frame, id, kind, start, length, is_uep, uep_index, eep_level, eep_type, fec)"""
    out = []
    for line in open(path).read().splitlines():
        t = line.split()
        out.append(dict(frame=int(t[0]), id=int(t[1]), kind=t[2], start=int(t[3]), length=int(t[4]), is_uep=int(t[5]), uep_index=int(t[6]), eep_level=int(t[7]),
                        eep_type=int(t[8]), fec=int(t[9])))
    return out


def transmitted_fibs(truth, n_out_frames):
    """(frame, 30 bytes) of every FIB the first n_out_frames frames carry, in order"""
    return [(f, truth["fib_data"][f, g, i].tobytes()) for f in range(n_out_frames) for g in range(4) for i in range(3)]


def transmitted_bytes(truth, k, first_frame, n_out_frames):
    """what a decoder of sub-channel index k that first sees frame `first_frame` must produce: a list of (frame, cif, bytes) -- empty while
    its time de-interleaver holds fewer than 16 CIFs, then the payload of the CIF 15 earlier (clause 12)"""
    pay = truth["payload"][k]
    out = []
    for f in range(first_frame, n_out_frames):
        for c in range(4):
            seen = 4 * (f - first_frame) + c + 1
            out.append((f, c, pay[4 * f + c - 15].tobytes() if seen >= 16 else b""))
    return out
