"""-m gpu: the frame session of the C ABI (dabgpu_frame_session_*, include/dabgpu.h) -- one batched device decode per transmission frame,
results fetched by (generation, FIB group / sub-channel, CIF).  Seven frames of random soft bits pushed one by one; every FIB group
(bytes, CRC mask, path error) and, from the frame that completes 16 CIFs on, every sub-channel's CIF (bytes, path error) must be what the
oracle decodes (oracle/: fic_decode_group, Deinterleaver + msc_decode_logical); generations older than 8 frames, sub-channels registered
later and FIC-less pushes answer NOT_READY (None)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_frame_session_against_the_oracle(oracle):
    import dabgpu
    rng = np.random.default_rng(77)
    subs_o = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(120, 27, eep_level=0, eep_type=1)]
    subs = [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(120, 27, False, 0, 0, 1)]
    late = dabgpu.SubChannel(400, 8, False, 0, 1, 0)
    n_frames = 11
    frames = rng.integers(-127, 128, (n_frames, 230400), dtype=np.int8)
    fs = dabgpu.FrameSession(0)
    try:
        fs.set_subchannels(subs)
        gens = []
        for k in range(n_frames):
            if k == 9:
                fs.set_subchannels(subs + [late])
            gens.append(fs.push_frame(frames[k], decode_fic=(k != 5)))
        assert gens == list(range(n_frames))
        # the last 8 generations are available, older ones are gone
        assert fs.fetch_fib_group(gens[2], 0) is None and fs.fetch_cif(gens[2], subs[0], 0) is None
        assert fs.fetch_fib_group(gens[5], 1) is None                      # pushed without the FIC
        assert fs.fetch_cif(gens[8], late, 0) is None                      # registered after that frame was pushed
        deint = [oracle.Deinterleaver(s.length * 8) for s in subs_o]
        logical = {}
        for k in range(n_frames):
            for c in range(4):
                cif = frames[k, 9216 + c * 55296: 9216 + (c + 1) * 55296]
                for si, s in enumerate(subs_o):
                    deint[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                    logical[(k, c, si)] = deint[si].deinterleave()
        checked_f = checked_m = 0
        for k in range(n_frames - 8, n_frames):
            if k != 5:
                for g in range(4):
                    got = fs.fetch_fib_group(gens[k], g)
                    eb, em, ee = oracle.fic_decode_group(frames[k, g * 2304:(g + 1) * 2304], 0)
                    assert got is not None and np.array_equal(got[0], eb) and got[1] == em and got[2] == ee, (k, g)
                    checked_f += 1
            for c in range(4):
                for si, s in enumerate(subs_o):
                    got = fs.fetch_cif(gens[k], subs[si], c)
                    lf = logical[(k, c, si)]
                    assert got is not None
                    if lf is None:                                         # fewer than 16 CIFs so far: the decoder's output is undefined upstream too
                        continue
                    dec, err = oracle.msc_decode_logical(s, lf, 0)
                    assert np.array_equal(got[0], dec) and got[1] == err, (k, c, si)
                    checked_m += 1
        # frames 3 .. 10: FIC of all but frame 5; logical frames from CIF 15 on (frame 3 holds CIFs 12 .. 15: three of its four have none yet)
        assert checked_f == 4 * 7 and checked_m == 2 * (4 * 8 - 3)
    finally:
        fs.close()
