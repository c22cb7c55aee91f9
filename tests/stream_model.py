"""Test infrastructure: a sequential restatement of OFDM_Demod's framing state machine
(reference: src/ofdm/ofdm_demodulator.cpp:235-358, :550-577, :922-950) composed from the CPU oracle's numeric
functions, plus a synthetic DAB ensemble generator (FIC with CRC-valid FIBs, EEP sub-channels through the
16-CIF time interleaver).  Used by the -m gpu tests as the expected-output side for the C++ mirror classes."""
import os
import platform

import numpy as np

F32 = np.float32


def mirror_core_model():
    """The Viterbi core model the C++ mirror classes decode with when nothing is said (dab-radio_amd/host/dab/dabgpu_shared_context.cpp::
    dabgpu_core_model_from_env): DABGPU_VITERBI_CORE=scalar|simd, else DABGPU_TIE_RULE=0|1, else the core the reference's own build selects on this
    host (dab_viterbi_decoder.cpp:51-73 under -march=native): AVX2 / SSE4.1 / AArch64 -> 1 (SIMD cores), otherwise 0 (scalar core).  The expected
    side of every test that drives the classes passes this to the oracle."""
    c = os.environ.get("DABGPU_VITERBI_CORE")
    if c:
        return 1 if c in ("simd", "1") else 0
    t = os.environ.get("DABGPU_TIE_RULE")
    if t is not None:
        return 1 if int(t) else 0
    m = platform.machine().lower()
    if m in ("aarch64", "arm64"):
        return 1
    if m in ("x86_64", "amd64", "i686", "i386"):
        try:
            flags = next(line for line in open("/proc/cpuinfo") if line.startswith("flags")).split()
        except (OSError, StopIteration):
            return 0
        return 1 if ("avx2" in flags or "sse4_1" in flags) else 0
    return 0


def l1_average(block):
    """CalculateL1Average (:922-932): sequential float32 accumulation"""
    t = (np.abs(block.real).astype(F32) + np.abs(block.imag).astype(F32)).astype(F32)
    return F32(np.cumsum(t, dtype=F32)[-1] / F32(t.size))


class StreamModel:
    """states: 0 FINDING_NULL_POWER_DIP, 1 READING_NULL_AND_PRS, 2/3 sync, 4 READING_SYMBOLS"""

    def __init__(self, oracle, mode=1):
        self.O = oracle
        self.mode = mode
        self.g = oracle.geometry(mode)
        self.cfg = oracle.sync_cfg_default()
        self.conj_ref, self.time_ref = oracle.sync_refs_mode(mode)
        self.mapper = oracle.mapper_n(self.g.nb_fft, self.g.nb_carriers)
        self.state = 0
        self.signal_avg = F32(0)
        self.null_start = False
        self.null_end = False
        self.ring = np.zeros(self.g.nb_null_period, np.complex64)
        self.ring_index = 0
        self.ring_length = 0
        self.corr = np.zeros(self.g.nb_null_period + self.g.nb_symbol_period, np.complex64)
        self.corr_length = 0
        self.frame = np.zeros(self.g.nb_frame_samples, np.complex64)
        self.frame_length = 0
        self.sync = oracle.SyncState(0.0, 0.0, 0, 0, 0, 0)
        self.fine_time_offset = 0
        self.frames_read = 0
        self.frames_desync = 0
        self.out_frames = []        # dict(bits, coarse, fine, offset, desync, fft, cp)

    def reset(self):                                             # :277-289
        self.state = 0
        self.corr_length = 0
        self.frames_desync += 1
        self.sync = self.O.SyncState(0.0, 0.0, 0, 0, 0, 0)
        self.fine_time_offset = 0

    def update_signal_average(self, block):                      # :934-950
        n, k = block.size, 100
        if n < k:
            return
        beta = F32(0.95)
        for i in range(0, n - k, 500):
            self.signal_avg = F32(beta * self.signal_avg + F32(F32(1.0) - beta) * l1_average(block[i:i + k]))

    def find_null(self, buf):                                    # :291-347
        n, k = buf.size, 100
        start_t = F32(self.signal_avg * F32(0.35))
        end_t = F32(self.signal_avg * F32(0.75))
        nb_read = n
        for i in range(0, n - k, k):
            l1 = l1_average(buf[i:i + k])
            if self.null_start:
                if l1 > end_t:
                    self.null_end = True
                    nb_read = i + k
                    break
            elif l1 < start_t:
                self.null_start = True
        cap = self.ring.size
        keep = min(nb_read, cap)                                 # sequential writes: only the last `cap` samples survive
        first = nb_read - keep
        idx = (self.ring_index + first + np.arange(keep)) % cap
        self.ring[idx] = buf[first:nb_read]
        self.ring_index = (self.ring_index + nb_read) % cap
        self.ring_length = min(cap, self.ring_length + nb_read)
        if not self.null_end:
            return nb_read
        L = self.ring_length
        self.corr[:L] = self.ring[(np.arange(L) + self.ring_index) % cap]
        self.corr_length = L
        self.null_start = self.null_end = False
        self.ring_length = 0
        self.state = 1
        return nb_read

    def read_null_prs(self, buf):                                # :349-358
        take = min(self.corr.size - self.corr_length, buf.size)
        self.corr[self.corr_length:self.corr_length + take] = buf[:take]
        self.corr_length += take
        if self.corr_length == self.corr.size:
            self.state = 2
        return take

    def run_sync(self):                                          # :360-548
        O, g = self.O, self.g
        prs_sym = self.corr[g.nb_null_period:g.nb_null_period + g.nb_fft]
        O.coarse_freq_sync_mode(self.mode, prs_sym, self.sync, self.cfg, self.time_ref)
        f = F32(F32(self.sync.freq_coarse) + F32(self.sync.freq_fine))
        ok, off, _ = O.fine_time_sync_mode(self.mode, prs_sym, f, self.cfg, self.conj_ref)
        if not ok:
            self.reset()
            return
        start = g.nb_null_period + off
        count = self.corr.size - start
        self.frame[:count] = self.corr[start:]
        self.frame_length = count
        self.corr_length = 0
        self.fine_time_offset = off
        self.state = 4

    def read_symbols(self, buf):                                 # :550-577
        O, g = self.O, self.g
        take = min(self.frame.size - self.frame_length, buf.size)
        self.frame[self.frame_length:self.frame_length + take] = buf[:take]
        self.frame_length += take
        if self.frame_length < self.frame.size:
            return take
        null_at = g.nb_frame_symbols * g.nb_symbol_period
        self.corr[:g.nb_null_period] = self.frame[null_at:null_at + g.nb_null_period]
        self.corr_length = g.nb_null_period
        f = F32(F32(self.sync.freq_coarse) + F32(self.sync.freq_fine))
        r = O.demod_frame_mode(self.mode, self.frame, f, want_fft=True, m=self.mapper)
        self.sync.freq_fine = float(O.update_fine_freq_mode(self.mode, self.sync.freq_fine, r["total_phase"], 0.9))
        self.frames_read += 1
        self.out_frames.append(dict(bits=r["bits"].copy(), coarse=F32(self.sync.freq_coarse), fine=F32(self.sync.freq_fine),
                                    offset=self.fine_time_offset, desync=self.frames_desync, fft=r["fft"].copy()))
        self.frame_length = 0
        self.state = 1
        return take

    def process(self, buf):                                      # :235-275
        buf = np.ascontiguousarray(buf, dtype=np.complex64)
        self.update_signal_average(buf)
        pos = 0
        while pos < buf.size:
            rest = buf[pos:]
            if self.state == 0:
                pos += self.find_null(rest)
            elif self.state == 1:
                pos += self.read_null_prs(rest)
            elif self.state in (2, 3):
                self.run_sync()
            else:
                pos += self.read_symbols(rest)


def make_ensemble_stream(oracle, n_frames, subs, seed, cfo=1.8e-3, timing_pad=1234, noise=3.0, amplitude=1.0 / 39.2, payload=None, fib_data=None):
    """n_frames transmission frames: FIC = 4 groups of 3 CRC-valid FIBs (random bytes, or fib_data [n_frames][4][3][30] -- e.g. the FIGs of
    tools/dabfig.py -- to which the CRC is added by the encoder), MSC = the listed EEP / UEP sub-channels
    (random payload, time interleaved across CIFs), other capacity units random bits; then CFO, noise, a lead-in of
    noise-only samples so that the NULL detector has a level to compare with.
    Returns (stream c64, dict with the transmitted fib bytes [n_frames][4][90] and payload per sub-channel [n_cif][nbytes])."""
    rng = np.random.default_rng(seed)
    n_cif = 4 * n_frames
    plans = [oracle.subchannel_plan(s) for s in subs]
    if payload is None:
        payload = [rng.integers(0, 256, (n_cif, p[2]), dtype=np.uint8) for p in plans]
    cif_bits = rng.integers(0, 2, (n_cif, oracle.NB_CIF_BITS), dtype=np.uint8)
    for s, p, pay in zip(subs, plans, payload):
        lf = np.stack([oracle.msc_encode_logical(s, pay[t]) for t in range(n_cif)])
        tx = oracle.time_interleave(lf)
        cif_bits[:, s.start_address * 64:(s.start_address + s.length) * 64] = tx
    fibs = rng.integers(0, 256, (n_frames, 4, 90), dtype=np.uint8)
    if fib_data is not None:
        fibs = np.ascontiguousarray(fib_data, dtype=np.uint8).reshape(n_frames, 4, 90)
    frames = []
    for f in range(n_frames):
        bits = np.empty(oracle.NB_FRAME_BITS, np.uint8)
        for g in range(4):
            bits[g * 2304:(g + 1) * 2304] = oracle.fic_encode_group(fibs[f, g])
        bits[9216:] = cif_bits[4 * f:4 * f + 4].reshape(-1)
        frames.append(oracle.modulate_frame(bits))
    tx = np.concatenate(frames)
    tx = oracle.apply_pll(tx, cfo, 0.37)
    # lead-in: a stretch of ordinary signal so the L1 level is established before the first NULL symbol arrives
    stream = np.concatenate([tx[oracle.NB_NULL_PERIOD:oracle.NB_NULL_PERIOD + 30000 + timing_pad], tx])
    stream = stream + noise * (rng.standard_normal(stream.size) + 1j * rng.standard_normal(stream.size))
    return (stream * amplitude).astype(np.complex64), dict(fibs=fibs, payload=payload, plans=plans)


def make_offair_like_capture(oracle, n_frames, subs, seed, cfo=2.3e-3, ppm=20.0, dropouts=((14, -0.03, 14000), (33, -0.03, 60000), (45, 0.2, 30000)),
                             noise=2.0, clip_fraction=0.004):
    """A stand-in for the reference's off-air recording (README.md:41 is a release asset, there is no network here): one ensemble of
    n_frames transmission frames as an RTL-SDR-style 8-bit capture with the impairments a real reception has and the plain generator
    above lacks:
      * multipath: two echoes inside the cyclic prefix (37 and 180 samples late, -6 dB and -12 dB, rotated),
      * sample-clock error of `ppm` parts per million (the fine time offset walks ~4 samples per 100 ms: windowed-sinc resampling),
      * carrier offset `cfo` (cycles per sample), a DC offset and an IQ gain / phase imbalance (1.05, 3 degrees) of the tuner,
      * additive noise, signal drop-outs (transmission frame, position inside it, samples): the first two wipe a NULL + phase reference
        symbol and force a failed synchronisation, a NULL search and a re-acquisition; the third only ruins data symbols,
      * u8 quantisation with clipping of the strongest `clip_fraction` of the components.
    Returns (capture bytes uint8 [2 x samples], truth dict of make_ensemble_stream)."""
    clean, truth = make_ensemble_stream(oracle, n_frames, subs, seed, cfo=0.0, noise=0.0, amplitude=1.0)
    rng = np.random.default_rng(seed + 1)
    x = clean.astype(np.complex128)
    y = x.copy()
    for delay, gain, phase in ((37, 0.5, 0.7), (180, 0.25, -2.1)):
        y[delay:] += gain * np.exp(1j * phase) * x[:-delay]
    # resample at (1 + ppm 1e-6): output sample n sits at input time n (1 + ppm 1e-6); 8-tap Hann-windowed sinc
    n_out = int((y.size - 16) / (1.0 + ppm * 1e-6))
    t = np.arange(n_out, dtype=np.float64) * (1.0 + ppm * 1e-6) + 4.0
    i0 = np.floor(t).astype(np.int64)
    frac = t - i0
    z = np.zeros(n_out, np.complex128)
    for k in range(-3, 5):
        u = k - frac
        w = np.sinc(u) * (0.5 + 0.5 * np.cos(np.pi * u / 4.0))
        z += w * y[i0 + k]
    n = np.arange(n_out, dtype=np.float64)
    z *= np.exp(2j * np.pi * (cfo * n + 0.37))
    rms = np.sqrt(np.mean(np.abs(z) ** 2))
    for fr, where, length in dropouts:
        a = int((fr + where) * oracle.NB_FRAME_SAMPLES) + 31234          # (the lead-in of make_ensemble_stream)
        z[a:a + length] = 0.0
    z += noise * (rng.standard_normal(n_out) + 1j * rng.standard_normal(n_out))
    z += (0.03 - 0.02j) * rms
    g, th = 1.05, np.deg2rad(3.0)
    i_, q_ = z.real, g * (z.imag * np.cos(th) + z.real * np.sin(th))
    comp = np.stack([i_, q_], axis=-1).reshape(-1)
    full = np.quantile(np.abs(comp[::17]), 1.0 - clip_fraction)           # the strongest components clip
    u8 = np.clip(np.rint(comp / full * 127.5 + 127.5), 0, 255).astype(np.uint8)
    return u8, truth


def expected_decode(oracle, frames_bits, subs):
    """what FIC_Decoder + MSC_Decoder of basic_radio produce from a sequence of frames: (FIB bytes of every CRC-valid FIB, decoded
    bytes per sub-channel), composed from the oracle's functions"""
    fibs = bytearray()
    msc = [bytearray() for _ in subs]
    deint = [oracle.Deinterleaver(s.length * 8) for s in subs]
    for bits in frames_bits:
        for g in range(4):
            eb, em, _ = oracle.fic_decode_group(bits[g * 2304:(g + 1) * 2304], mirror_core_model())
            for i in range(3):
                if em & (1 << i):
                    fibs += eb[32 * i:32 * i + 30].tobytes()
        for c in range(4):
            cif = bits[9216 + c * 55296:9216 + (c + 1) * 55296]
            for si, s in enumerate(subs):
                deint[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                lf = deint[si].deinterleave()
                if lf is not None:
                    msc[si] += oracle.msc_decode_logical(s, lf, mirror_core_model())[0].tobytes()
    return bytes(fibs), [bytes(m) for m in msc]
