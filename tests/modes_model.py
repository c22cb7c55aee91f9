"""Test infrastructure: a mode-generic DQPSK OFDM frame generator (transmission modes I-IV geometries) used by the mode
II-IV tests.  The first symbol is a random unit-magnitude reference (the differential demodulator needs no PRS), then
75 / 152 data symbols carrying the given bits through the frequency interleaver, cyclic prefix, NULL symbol at the end
(the demodulator's frame-buffer layout)."""
import numpy as np


def make_tx_frame(oracle, mode, bits01, rng):
    """transmission order (NULL first, then PRS and the data symbols), PRS = the mode's phase reference symbol"""
    g = oracle.geometry(mode)
    f = make_frame(oracle, mode, bits01, rng, prs=True)
    body = f[:g.nb_frame_symbols * g.nb_symbol_period]
    return np.concatenate([np.zeros(g.nb_null_period, np.complex64), body])


def make_frame(oracle, mode, bits01, rng, prs=False):
    g = oracle.geometry(mode)
    N, NC, M = g.nb_fft, g.nb_carriers, g.nb_carriers // 2
    mapper = oracle.mapper_n(N, NC)
    n_data = g.nb_frame_symbols - 1
    b = np.asarray(bits01, np.uint8).reshape(n_data, 2 * NC)
    carriers = np.arange(NC)
    k = np.where(carriers < M, carriers - M, carriers - M + 1)
    bins = (N + k) % N                                          # carrier index c -> FFT bin
    spec = np.zeros((g.nb_frame_symbols, N), np.complex128)
    if prs:
        cur = oracle.prs_fft_mode(mode)[bins].astype(np.complex128)
    else:
        cur = np.exp(2j * np.pi * rng.integers(0, 4, NC) / 4)   # reference symbol, carrier order
    spec[0, bins] = cur
    a = np.sqrt(0.5)
    for s in range(n_data):
        # soft bit n of the symbol belongs to carrier mapper[n]: re from bits[n], im from bits[n + NC]
        z = np.zeros(NC, np.complex128)
        z[mapper] = (1.0 - 2.0 * b[s, :NC]) * a + 1j * (1.0 - 2.0 * b[s, NC:]) * a
        cur = cur * z
        spec[s + 1, bins] = cur
    t = np.fft.ifft(spec, axis=1) * (N / np.sqrt(NC))
    frame = np.zeros(g.nb_frame_samples, np.complex64)
    body = frame[:g.nb_frame_symbols * g.nb_symbol_period].reshape(g.nb_frame_symbols, g.nb_symbol_period)
    body[:, g.nb_cp:] = t
    body[:, :g.nb_cp] = t[:, N - g.nb_cp:]
    return frame
