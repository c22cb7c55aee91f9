"""Test infrastructure: a sequential restatement of OFDM_Demod's framing state machine
(reference: src/ofdm/ofdm_demodulator.cpp:235-358, :550-577, :922-950) composed from the CPU oracle's numeric
functions, plus a synthetic DAB ensemble generator (FIC with CRC-valid FIBs, EEP sub-channels through the
16-CIF time interleaver).  Used by the -m gpu tests as the expected-output side for the C++ mirror classes."""
import numpy as np

F32 = np.float32


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


def make_ensemble_stream(oracle, n_frames, subs, seed, cfo=1.8e-3, timing_pad=1234, noise=3.0, amplitude=1.0 / 39.2, payload=None):
    """n_frames transmission frames: FIC = 4 groups of 3 CRC-valid random FIBs, MSC = the listed EEP sub-channels
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
