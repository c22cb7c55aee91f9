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


class Multiplex:
    """The canonical multiplex of SURVEY 8(d) config 4 for `n` distinct ensembles: CRC-valid FIBs and 18 x 192-byte sub-channel
    payloads per ensemble, channel coded and TIME INTERLEAVED (clause 12): the payload changes with every CIF, with a period of
    `period` CIFs = period / 4 transmission frames, so that a sequence of period / 4 stored frames repeats a legal, steady-state
    transmission in which every decoded logical frame tells which CIFs it was assembled from (a wrong ring slot or age, or two frames
    in flight in the wrong order, decode to another CIF's payload or to garbage -- with one repeated frame they would not show).
    Decoded CIF r (counted from the first frame) carries payload[(r - 15) mod period]; frame j carries fibs[j mod (period / 4)]."""

    def __init__(self, n, seed, device, period=8):
        assert period % 4 == 0 and period >= 4
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.n, self.period, self.n_frames = n, period, period // 4
        nf = self.n_frames
        fib_data = torch.randint(0, 256, (n, nf, 4, 3, 30), generator=g, device=device, dtype=torch.uint8)
        self.fibs = torch.cat([fib_data, crc16(fib_data)], dim=-1).reshape(n, nf, 4, 96)
        pr96 = torch.from_numpy(prbs_bytes(96)).to(device)
        fic_mother = conv_encode(bytes_to_bits(self.fibs ^ pr96))
        fic_tx = fic_mother[..., torch.from_numpy(kept_index([(16, 21), (15, 3)])).to(device)]        # [n,nf,4,2304]
        self.payload = torch.randint(0, 256, (n, period, N_SUB, SUB_BYTES), generator=g, device=device, dtype=torch.uint8)
        pr192 = torch.from_numpy(prbs_bytes(SUB_BYTES)).to(device)
        kidx = torch.from_numpy(kept_index([(8, 45), (7, 3)])).to(device)                               # EEP 3-A, n = 8
        logical = torch.empty((n, period, N_SUB, 3072), dtype=torch.uint8, device=device)
        for e0 in range(0, n, 64):
            logical[e0:e0 + 64] = conv_encode(bytes_to_bits(self.payload[e0:e0 + 64] ^ pr192))[..., kidx]
        # time interleaver: transmitted CIF s carries bit i of the logical frame that is bitrev4(i mod 16) CIFs older
        # (the receiver takes bit i from the CIF that is 15 - bitrev4(i mod 16) CIFs old, cif_deinterleaver.cpp:57-68)
        tx = torch.empty_like(logical)
        for k in range(16):
            d = int("{:04b}".format(k)[::-1], 2)
            for s_ in range(period):
                tx[:, s_, :, k::16] = logical[:, (s_ - d) % period, :, k::16]
        cifs = tx.reshape(n, nf, 4 * 55296)
        self.frame_bits = torch.cat([fic_tx.reshape(n, nf, 9216), cifs], dim=2).reshape(n, nf, 75, 3072)

    def subchannels(self, dabgpu):
        return [dabgpu.SubChannel(SUB_CU * s, SUB_CU, 0, 0, 2, 0) for s in range(N_SUB)]


def ensemble_iq(n_ensembles, n_distinct, seed, device, mapper, prs, noise=0.05, period=8):
    """IQ of n_ensembles ensembles built from n_distinct (<= 64) seeded multiplexes: ensemble e carries
    multiplex e % n_distinct; every ensemble gets its own noise realisation.  Returns (iq [period / 4][E][196608] complex64 -- the
    period / 4 transmission frames that repeat -- and the Multiplex)."""
    mux = Multiplex(n_distinct, seed, device, period)
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
