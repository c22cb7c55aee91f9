"""Transmit-side generator used by bench.py, tools/bench_decode.py and the full-size -m gpu tests (untimed plumbing, torch on
the device): ETSI EN 300 401 channel coding (energy dispersal 10, convolutional code 11.1, puncturing 11.1.2 / 11.2 / 11.3,
FIB CRC 5.2.1) and the Mode-I OFDM modulator (14.5-14.7).  It is the inverse of the path under test, written independently
of it; nothing here is part of the product."""
import numpy as np
import torch

TAPS = [[0, 2, 3, 5, 6], [0, 1, 2, 3, 6], [0, 1, 4, 6], [0, 2, 3, 5, 6]]        # polynomials 133,171,145,133 (octal)
NB_FRAME_SAMPLES = 196608
N_SUB, SUB_BYTES, SUB_CU = 18, 192, 48                 # 18 sub-channels x 48 CU, EEP 3-A, 192 bytes / CIF
FIC_STEPS_PER_FRAME = 4 * 774
MSC_STEPS_PER_FRAME = 4 * N_SUB * 1542


def canonical_layout():
    """18 x 48 CU EEP 3-A (PI_8 x 45 blocks, PI_7 x 3 blocks, 192 bytes per CIF) filling all 864 CU"""
    return [dict(start=SUB_CU * s, length=SUB_CU, is_uep=0, uep_index=0, eep_level=2, eep_type=0, segments=[(8, 45), (7, 3)], nbytes=SUB_BYTES)
            for s in range(N_SUB)]


def mixed_layout():
    """A multiplex as they are on air: 14 sub-channels of 8 sizes and 4 protection families on 832 of the 864 capacity units --
    eight DAB+ services EEP 3-A (3 x 48 CU = 64 kbit/s, 3 x 60 CU = 80 kbit/s, 2 x 72 CU = 96 kbit/s), two EEP 2-B (42 CU = 64 kbit/s), three
    MP2 services on UEP (table rows 35 / 38 / 43: 128 kbit/s level 3 = 96 CU with 4 padding bits, 160 kbit/s level 5 = 80 CU, 192 kbit/s
    level 5 = 96 CU) and one 8 CU EEP 2-A data service (the n = 1 special case of msc_decoder.cpp:77-94 / subchannel_protection_tables.h:88-139).
    segments = [(PI, L blocks of 128 mother bits)], nbytes = decoded bytes per CIF: ETSI EN 300 401 tables 7, 9, 10, written out here for
    the generator (the product derives them from its own copy of the tables; check_layout compares the two)."""
    subs, at = [], 0

    def add(length, segments, nbytes, **kw):
        nonlocal at
        d = dict(start=at, length=length, is_uep=0, uep_index=0, eep_level=0, eep_type=0, segments=segments, nbytes=nbytes)
        d.update(kw)
        subs.append(d)
        at += length
    for n6, count in ((8, 3), (10, 3), (12, 2)):                     # EEP 3-A, length = 6 n: L1 = 6n - 3 blocks of PI_8, L2 = 3 of PI_7, 24 n bytes
        for _ in range(count):
            add(6 * n6, [(8, 6 * n6 - 3), (7, 3)], 24 * n6, eep_level=2)
    for _ in range(2):                                               # EEP 2-B, length = 21 n (n = 2): L1 = 24n - 3 of PI_6, L2 = 3 of PI_5, 96 n bytes
        add(42, [(6, 45), (5, 3)], 192, eep_level=1, eep_type=1)
    add(96, [(16, 11), (9, 22), (6, 60), (10, 3)], 384, is_uep=1, uep_index=35)     # 128 kbit/s, protection level 3, 4 padding bits
    add(80, [(5, 11), (4, 19), (2, 87), (4, 3)], 480, is_uep=1, uep_index=38)       # 160 kbit/s, level 5
    add(96, [(6, 11), (4, 20), (2, 110), (5, 3)], 576, is_uep=1, uep_index=43)      # 192 kbit/s, level 5
    add(8, [(13, 5), (12, 1)], 24, eep_level=1)                                     # EEP 2-A, n = 1
    return subs


def check_layout(layout, dabgpu):
    """the generator's plans against the product's (dabgpu_subchannel_plan): a row typed wrongly above must not pass as a decoder bug"""
    import ctypes as C
    L = dabgpu.lib()
    for d in layout:
        sc = dabgpu.SubChannel(d["start"], d["length"], d["is_uep"], d["uep_index"], d["eep_level"], d["eep_type"])
        pi, lx, nb = (C.c_int * 4)(), (C.c_int * 4)(), C.c_int(0)
        n = L.dabgpu_subchannel_plan(C.byref(sc), pi, lx, C.byref(nb))
        got = [(pi[k], lx[k]) for k in range(n) if lx[k] > 0]
        assert got == [sg for sg in d["segments"] if sg[1] > 0] and nb.value == d["nbytes"], (d, got, nb.value)
        kept = sum(4 * L_ * (8 + p) for p, L_ in d["segments"]) + 12
        assert kept <= d["length"] * 64, (d, kept)


def prbs_bytes(n):
    reg, out = 0xFFFF, np.empty(n, np.uint8)
    for k in range(n):
        b = 0
        for i in range(8):
            v = ((reg >> 8) ^ (reg >> 4)) & 1
            b |= v << (7 - i)
            reg = ((reg << 1) | v) & 0xFFFF
        out[k] = b
    return out


def kept_index(segments):
    """mother-code bit indices that survive puncturing for [(PI, L blocks)...] + the PI_X tail"""
    order = [0, 4, 2, 6, 1, 5, 3, 7]
    idx, m = [], 0
    for pi, L in segments + [(8, None)]:
        cnt = [1] * 8
        for e in range(pi):
            cnt[order[e % 8]] += 1
        n_groups = 6 if L is None else 32 * L
        for g in range(n_groups):
            idx += [m + 4 * g + r for r in range(cnt[g % 8])]
        m += 4 * n_groups
    return np.array(idx, dtype=np.int64)


def bytes_to_bits(x):
    sh = torch.arange(7, -1, -1, device=x.device, dtype=torch.uint8)
    return ((x.unsqueeze(-1) >> sh) & 1).reshape(*x.shape[:-1], -1)


def conv_encode(bits):
    """bits [..., n] (0/1 uint8) -> mother code [..., 4*(n+6)]"""
    n = bits.shape[-1]
    x = torch.nn.functional.pad(bits, (6, 6))
    outs = []
    for taps in TAPS:
        acc = torch.zeros(bits.shape[:-1] + (n + 6,), dtype=torch.uint8, device=bits.device)
        for d in taps:
            acc ^= x[..., 6 - d: 6 - d + n + 6]
        outs.append(acc)
    return torch.stack(outs, dim=-1).reshape(*bits.shape[:-1], -1)


def crc16(data):
    """data [..., nbytes] uint8 -> [..., 2] bytes (poly 0x1021, init/xorout 0xFFFF)"""
    bits = bytes_to_bits(data).to(torch.int32)
    crc = torch.full(data.shape[:-1], 0xFFFF, dtype=torch.int32, device=data.device)
    for i in range(bits.shape[-1]):
        msb = ((crc >> 15) & 1) ^ bits[..., i]
        crc = ((crc << 1) & 0xFFFF) ^ (msb * 0x1021)
    crc ^= 0xFFFF
    return torch.stack([(crc >> 8) & 0xFF, crc & 0xFF], dim=-1).to(torch.uint8)


def modulate(frame_bits, prs, mapper, freq=None, out=None, chunk=32, scale=1.0 / 39.2):
    """frame_bits [n,75,3072] uint8 on the device -> iq [n,196608] complex64 in frame-buffer layout (76 symbols, then the NULL
    period left at zero): QPSK on de-interleaved positions, differential modulation from the PRS, IFFT, cyclic prefix,
    optional per-frame carrier offset freq [n] (cycles per sample)."""
    device = frame_bits.device
    n_frames = frame_bits.shape[0]
    mp = torch.from_numpy(mapper.astype(np.int64)).to(device)
    bins = torch.where(mp < 768, mp + (2048 - 768), mp - 768 + 1)            # carrier index -> FFT bin
    prs_t = torch.from_numpy(prs).to(device)
    a = 0.70710678
    iq = out if out is not None else torch.zeros((n_frames, NB_FRAME_SAMPLES), dtype=torch.complex64, device=device)
    n = torch.arange(NB_FRAME_SAMPLES, device=device, dtype=torch.float64) if freq is not None else None
    for k0 in range(0, n_frames, chunk):
        k1 = min(k0 + chunk, n_frames)
        m = k1 - k0
        b = frame_bits[k0:k1]
        z = torch.complex((1.0 - 2.0 * b[:, :, :1536].float()) * a, (1.0 - 2.0 * b[:, :, 1536:].float()) * a)
        spec = torch.zeros((m, 76, 2048), dtype=torch.complex64, device=device)
        spec[:, 0] = prs_t
        cur = prs_t[bins].expand(m, -1).clone()
        for s_ in range(75):                                  # differential modulation, symbol by symbol
            cur = cur * z[:, s_]
            spec[:, s_ + 1, bins] = cur
        t = torch.fft.ifft(spec, dim=2) * 2048.0
        body = iq[k0:k1, : 76 * 2552].view(m, 76, 2552)
        body[:, :, 504:] = t
        body[:, :, :504] = t[:, :, 2048 - 504:]
        iq[k0:k1, 76 * 2552:] = 0
        if freq is not None:
            ph = (-2.0 * np.pi) * freq[k0:k1, None].double() * n[None, :]
            iq[k0:k1] *= torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
    iq *= scale                                                                 # unit-ish RMS like a normalised capture
    return iq


def random_frames(n_frames, seed, device, mapper, prs, chunk=32):
    """configs[1] input: random-payload frames with a +-5 kHz carrier offset per frame (so the PLL does real work).
    Returns (iq [n,196608] complex64, bits [n,75,3072] uint8, freq [n] float32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    freq = ((torch.rand(n_frames, generator=g, device=device) * 2 - 1) * (5000.0 / 2.048e6)).float().contiguous()
    bits = torch.empty((n_frames, 75, 3072), dtype=torch.uint8, device=device)
    for k0 in range(0, n_frames, chunk):
        k1 = min(k0 + chunk, n_frames)
        bits[k0:k1] = torch.randint(0, 2, (k1 - k0, 75, 3072), generator=g, device=device, dtype=torch.uint8)
    iq = modulate(bits, prs, mapper, freq=freq, chunk=chunk)
    return iq, bits, freq


# ---- DAB+ audio super frames (ETSI TS 102 563 clauses 5.2, 6): generator side only ----
_GF_EXP = np.zeros(512, np.int64)
_GF_LOG = np.zeros(256, np.int64)
_x = 1
for _i in range(255):
    _GF_EXP[_i] = _GF_EXP[_i + 255] = _x
    _GF_LOG[_x] = _i
    _x <<= 1
    if _x & 0x100:
        _x ^= 0x11D


def _gmul(a, b):
    return 0 if a == 0 or b == 0 else int(_GF_EXP[_GF_LOG[a] + _GF_LOG[b]])


def rs_parity(data):
    """RS(120,110) parity of 110 data bytes: generator with roots alpha^0..alpha^9 over GF(2^8), p(x) = x^8+x^4+x^3+x^2+1"""
    g = [1] + [0] * 10
    for i in range(10):
        for j in range(i + 1, 0, -1):
            g[j] = g[j - 1] ^ _gmul(g[j], int(_GF_EXP[i]))
        g[0] = _gmul(g[0], int(_GF_EXP[i]))
    rem = [0] * 10
    for d in data:
        fb = int(d) ^ rem[9]
        for i in range(9, 0, -1):
            rem[i] = rem[i - 1] ^ _gmul(fb, g[i])
        rem[0] = _gmul(fb, g[0])
    return rem[::-1]


def _crc16(data, poly, init, xorout):
    crc = init
    for b in data:
        crc ^= int(b) << 8
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ xorout


def make_superframe(rng, n):
    """one audio super frame of 5 logical frames of n bytes (48 kHz, SBR: 3 access units with valid CRCs, fire code, RS parity)"""
    n_rs = 5 * n // 120
    data_len = 110 * n_rs
    sf = np.zeros(5 * n, np.uint8)
    sf[2] = 0x60
    first = 3 + 3
    cuts = [first, first + (data_len - first) // 3, first + 2 * (data_len - first) // 3, data_len]
    bits = []
    for v in cuts[1:3]:
        bits += [(v >> (11 - b)) & 1 for b in range(12)]
    sf[3:6] = np.packbits(np.array(bits, np.uint8))
    for i in range(3):
        a, b = cuts[i], cuts[i + 1]
        sf[a:b - 2] = rng.integers(0, 256, b - a - 2, dtype=np.uint8)
        c = _crc16(sf[a:b - 2], 0x1021, 0xFFFF, 0xFFFF)
        sf[b - 2], sf[b - 1] = c >> 8, c & 0xFF
    fc = _crc16(sf[2:11], 0x782F, 0, 0)
    sf[0], sf[1] = fc >> 8, fc & 0xFF
    for i in range(n_rs):
        sf[i + 110 * n_rs::n_rs] = rs_parity(sf[i:i + 110 * n_rs:n_rs])
    return sf


class Multiplex:
    """The canonical multiplex of SURVEY 8(d) config 4 for `n` distinct ensembles: CRC-valid FIBs and 18 x 192-byte sub-channel
    payloads per ensemble, channel coded and TIME INTERLEAVED (clause 12): the payload changes with every CIF, with a period of
    `period` CIFs = period / 4 transmission frames, so that a sequence of period / 4 stored frames repeats a legal, steady-state
    transmission in which every decoded logical frame tells which CIFs it was assembled from (a wrong ring slot or age, or two frames
    in flight in the wrong order, decode to another CIF's payload or to garbage -- with one repeated frame they would not show).
    Decoded CIF r (counted from the first frame) carries payload[(r - 15) mod period]; frame j carries fibs[j mod (period / 4)]."""

    def __init__(self, n, seed, device, period=8, superframes=False, rs_errors=0, layout=None, fig=False):
        """superframes: every sub-channel carries DAB+ audio super frames (5 logical frames each; period must be a multiple of 20 CIFs so
        that the stored frames repeat whole super frames) drawn from 32 generated ones, each RS codeword with rs_errors damaged symbols
        -- the payload the DAB+ outer code downstream of the channel decoder has real work on.
        layout: the multiplex (canonical_layout() by default, mixed_layout(), ...): capacity units no sub-channel occupies carry random bits
        fig: the FIBs carry the ensemble's Fast Information Groups (tools/dabfig.py: FIG 0/0 with the CIF count, 0/1 sub-channel organisation, 0/2
        services, 0/9, 0/17, labels -- a carousel over the stored frames) instead of random bytes: what the reference's FIG_Processor /
        DAB_Database_Updater can build its database from (tests/test_reference_callers_run.py does that on the single-receiver capture)"""
        assert period % 4 == 0 and period >= 4
        assert not superframes or period % 20 == 0
        self.layout = layout if layout is not None else canonical_layout()
        canonical = layout is None
        assert canonical or not superframes
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.n, self.period, self.n_frames = n, period, period // 4
        nf = self.n_frames
        fib_data = torch.randint(0, 256, (n, nf, 4, 3, 30), generator=g, device=device, dtype=torch.uint8)
        if fig:
            import dabfig
            lay = self.layout
            self.fig_descriptions = [dabfig.describe(lay, seed=seed + e, eid=0xE000 | ((37 * seed + e) & 0xFFF)) for e in range(n)]
            packed = [dabfig.repeating_frames(d, nf) for d in self.fig_descriptions]
            self.fig_groups_dropped = sum(p[1] for p in packed)       # labels that did not fit into nf x 12 FIBs (the MCI always does from nf = 2 on)
            fib_data = torch.from_numpy(np.stack([p[0] for p in packed])).to(device)
        self.fibs = torch.cat([fib_data, crc16(fib_data)], dim=-1).reshape(n, nf, 4, 96)
        pr96 = torch.from_numpy(prbs_bytes(96)).to(device)
        fic_mother = conv_encode(bytes_to_bits(self.fibs ^ pr96))
        fic_tx = fic_mother[..., torch.from_numpy(kept_index([(16, 21), (15, 3)])).to(device)]        # [n,nf,4,2304]
        if superframes:
            rng = np.random.default_rng(seed + 17)
            base = np.stack([make_superframe(rng, SUB_BYTES) for _ in range(32)])                  # [32][5 * 192]
            self.superframes_clean = base
            n_rs = 5 * SUB_BYTES // 120
            pick = rng.integers(0, 32, (n, N_SUB, period // 5))
            sfs = base[pick].copy()                                                                 # [n][sub][q][960]
            for k in range(rs_errors):         # distinct symbols of every codeword, away from the fire-code bytes at the start
                col = (rng.integers(2, 24, pick.shape + (n_rs,)) + 24 * k) * n_rs + np.arange(n_rs)
                np.put_along_axis(sfs, col, np.take_along_axis(sfs, col, axis=-1) ^ rng.integers(1, 256, col.shape, dtype=np.uint8), axis=-1)
            self.superframe_pick = pick
            pl = sfs.reshape(n, N_SUB, period // 5, 5, SUB_BYTES).reshape(n, N_SUB, period, SUB_BYTES).transpose(0, 2, 1, 3)
            self.payload = torch.from_numpy(np.ascontiguousarray(pl)).to(device)
            torch.randint(0, 256, (1,), generator=g, device=device)                                # (keeps the generator's stream position simple)
        elif canonical:
            self.payload = torch.randint(0, 256, (n, period, N_SUB, SUB_BYTES), generator=g, device=device, dtype=torch.uint8)
        if canonical:
            self.payloads = [self.payload[:, :, s_] for s_ in range(N_SUB)]                          # per sub-channel [n][period][bytes]
        else:
            self.payloads = [torch.randint(0, 256, (n, period, d["nbytes"]), generator=g, device=device, dtype=torch.uint8) for d in self.layout]
        self.sub_bytes = [d["nbytes"] for d in self.layout]
        self.cif_out_bytes = sum(self.sub_bytes)
        self.msc_steps_per_frame = 4 * sum(sum(32 * L for _, L in d["segments"]) + 6 for d in self.layout)
        # the CIFs as transmitted: random bits where no sub-channel lives, every sub-channel coded, punctured and time interleaved
        tx = torch.randint(0, 2, (n, period, 55296), generator=g, device=device, dtype=torch.uint8) if not canonical else \
            torch.empty((n, period, 55296), dtype=torch.uint8, device=device)
        plans = {}
        for d, pay in zip(self.layout, self.payloads):
            key = (tuple(d["segments"]), d["nbytes"])
            if key not in plans:
                plans[key] = (torch.from_numpy(prbs_bytes(d["nbytes"])).to(device), torch.from_numpy(kept_index(list(d["segments"]))).to(device))
            pr, kidx = plans[key]
            nk = kidx.numel()
            assert nk <= d["length"] * 64
            logical = torch.empty((n, period, nk), dtype=torch.uint8, device=device)
            for e0 in range(0, n, 64):
                logical[e0:e0 + 64] = conv_encode(bytes_to_bits(pay[e0:e0 + 64] ^ pr))[..., kidx]
            # time interleaver: transmitted CIF s carries bit i of the logical frame that is bitrev4(i mod 16) CIFs older
            # (the receiver takes bit i from the CIF that is 15 - bitrev4(i mod 16) CIFs old, cif_deinterleaver.cpp:57-68); the
            # sub-channel's padding bits (beyond the code word) keep their random filling
            sub = tx[:, :, d["start"] * 64:d["start"] * 64 + nk]
            for k in range(16):
                dd = int("{:04b}".format(k)[::-1], 2)
                for s_ in range(period):
                    sub[:, s_, k::16] = logical[:, (s_ - dd) % period, k::16]
        cifs = tx.reshape(n, nf, 4 * 55296)
        self.frame_bits = torch.cat([fic_tx.reshape(n, nf, 9216), cifs], dim=2).reshape(n, nf, 75, 3072)

    def subchannels(self, dabgpu):
        return [dabgpu.SubChannel(d["start"], d["length"], d["is_uep"], d["uep_index"], d["eep_level"], d["eep_type"]) for d in self.layout]

    def expected_cif(self, cif_index):
        """decoded bytes of CIF number cif_index (counted from the first stored frame), all sub-channels back to back: [n][cif_out_bytes]"""
        return torch.cat([p[:, (cif_index - 15) % self.period] for p in self.payloads], dim=1)


def ensemble_iq(n_ensembles, n_distinct, seed, device, mapper, prs, noise=0.05, period=8, superframes=False, rs_errors=0, layout=None, fig=False):
    """IQ of n_ensembles ensembles built from n_distinct (<= 64) seeded multiplexes: ensemble e carries
    multiplex e % n_distinct; every ensemble gets its own noise realisation.  Returns (iq [period / 4][E][196608] complex64 -- the
    period / 4 transmission frames that repeat -- and the Multiplex)."""
    mux = Multiplex(n_distinct, seed, device, period, superframes=superframes, rs_errors=rs_errors, layout=layout, fig=fig)
    nf = mux.n_frames
    iq = torch.empty((nf, n_ensembles, NB_FRAME_SAMPLES), dtype=torch.complex64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    for f in range(nf):
        base = modulate(mux.frame_bits[:, f].contiguous(), prs, mapper)
        for e0 in range(0, n_ensembles, n_distinct):
            m = min(n_distinct, n_ensembles - e0)
            iq[f, e0:e0 + m] = base[:m]
            if noise:
                nz = torch.randn((m, NB_FRAME_SAMPLES, 2), generator=g, dtype=torch.float32, device=device)
                iq[f, e0:e0 + m] += noise * torch.view_as_complex(nz)
        del base
    return iq, mux


def add_noise_(iq, noise, seed, chunk=64):
    """in place: every receiver's own noise realisation on frame-aligned iq [nf][E][196608] built with noise = 0"""
    g = torch.Generator(device=iq.device)
    g.manual_seed(seed)
    for f in range(iq.shape[0]):
        for e0 in range(0, iq.shape[1], chunk):
            e1 = min(e0 + chunk, iq.shape[1])
            iq[f, e0:e1] += noise * torch.view_as_complex(torch.randn((e1 - e0, iq.shape[2], 2), generator=g, dtype=torch.float32, device=iq.device))
    return iq


def ensemble_slices(iq, n_distinct, seed, lead, stride, max_cfo_hz=5000.0, max_toff=100, noise=0.05, chunk=64):
    """The sync-enabled configurations of SURVEY 8(d): from the frame-aligned iq [nf][E][196608] of ensemble_iq (noise-free there: pass the
    noise here) build per-receiver slices [nf][E][stride] in which frame f of receiver e begins at sample lead + toff[e] (toff uniform in
    [-max_toff, max_toff]: the receiver expects the PRS at `lead`), rotated by a carrier offset cfo[e] (uniform in +-max_cfo_hz, cycles
    per sample, continuous over the repeating frames), with the receiver's own noise over the whole slice (the NULL symbol included).
    Returns (slices, cfo [E] float32, toff [E] int32)."""
    nf, E, L = iq.shape[0], iq.shape[1], 76 * 2552
    device = iq.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cfo = ((torch.rand(E, generator=g, device=device) * 2 - 1) * (max_cfo_hz / 2.048e6)).double()
    # the stored frames repeat: a carrier offset that is a whole number of cycles per repetition keeps the phase continuous
    rep = nf * NB_FRAME_SAMPLES
    cfo = torch.round(cfo * rep) / rep
    toff = torch.randint(-max_toff, max_toff + 1, (E,), generator=g, device=device, dtype=torch.int64)
    out = torch.empty((nf, E, stride), dtype=torch.complex64, device=device)
    n = torch.arange(stride, device=device, dtype=torch.int64)
    for f in range(nf):
        for e0 in range(0, E, chunk):
            e1 = min(e0 + chunk, E)
            src = n[None, :] - (lead + toff[e0:e1, None])                      # frame sample that lands on slice sample n
            inside = (src >= 0) & (src < L)
            x = torch.gather(iq[f, e0:e1], 1, src.clamp(0, L - 1))
            ph = (2.0 * np.pi) * cfo[e0:e1, None] * (src + f * NB_FRAME_SAMPLES).double()
            x = x * torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
            x = torch.where(inside, x, torch.zeros_like(x))
            if noise:
                x = x + noise * torch.view_as_complex(torch.randn((e1 - e0, stride, 2), generator=g, dtype=torch.float32, device=device))
            out[f, e0:e1] = x
    return out, cfo.float(), toff.int()


def ensemble_streams_u8(iq, seed, block, max_cfo_hz=5000.0, noise=0.05, chunk=32):
    """Unsynchronised capture streams for the device-resident chain (stream bank -> ring -> decoders -> DAB+): from the frame-aligned
    iq [nf][E][196608] (noise-free) build, per receiver, the repeating transmission -- [76 symbols | NULL] x nf -- starting at a random
    sample (the receiver knows nothing about where frames begin), with its own carrier offset and noise, quantised like an RTL-SDR
    capture (raw_u8: (x / full_scale) * 127.5 + 127.5, rounded, clipped).  The buffer repeats its first `block` samples at the end so that
    any block of `block` samples starting inside one repetition is contiguous.
    Returns (raw [E][nf * 196608 + block][2] uint8, start [E] int64 = stream sample at which frame 0's PRS begins, cfo [E])."""
    nf, E = iq.shape[0], iq.shape[1]
    device = iq.device
    rep = nf * NB_FRAME_SAMPLES
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cfo = ((torch.rand(E, generator=g, device=device) * 2 - 1) * (max_cfo_hz / 2.048e6)).double()
    cfo = torch.round(cfo * rep) / rep
    start = torch.randint(0, rep, (E,), generator=g, device=device, dtype=torch.int64)
    raw = torch.empty((E, rep + block, 2), dtype=torch.uint8, device=device)
    flat = iq.permute(1, 0, 2).reshape(E, rep) if nf > 1 else iq[0]         # [E][rep]: frames back to back (a view when nf == 1)
    n = torch.arange(rep, device=device, dtype=torch.int64)
    full_scale = 4.0 * float(iq[0, :min(E, 8)].abs().square().mean().sqrt().item()) + 4.0 * noise
    for e0 in range(0, E, chunk):
        e1 = min(e0 + chunk, E)
        src = (n[None, :] - start[e0:e1, None]) % rep
        if nf > 1:
            x = torch.gather(iq.permute(1, 0, 2)[e0:e1].reshape(e1 - e0, rep), 1, src)
        else:
            x = torch.gather(flat[e0:e1], 1, src)
        ph = (2.0 * np.pi) * cfo[e0:e1, None] * n[None, :].double()
        x = x * torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
        x = torch.view_as_real(x)
        if noise:
            x = x + noise * torch.randn(x.shape, generator=g, dtype=torch.float32, device=device)
        q = torch.clamp(torch.round(x * (127.5 / full_scale) + 127.5), 0, 255).to(torch.uint8)
        raw[e0:e1, :rep] = q
        raw[e0:e1, rep:] = q[:, :block]
    return raw, start, cfo.float()
