"""Test infrastructure: a DAB+ audio super-frame generator (ETSI TS 102 563 clauses 5.2 and 6: header, access units with
CRC, fire code, RS(120,110) parity, byte interleaving across 5 logical frames) built on the oracle's encoder, plus helpers
to damage the result.  Used by the CPU pin tests and the -m gpu parity tests."""
import numpy as np


def crc16_ccitt(data):
    crc = 0xFFFF
    for b in data:
        crc ^= int(b) << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def make_superframe(oracle, rng, n, dac_rate=1, sbr_flag=1, channel_mode=1, ps_flag=0, mpeg_config=0, bad_au_crc=()):
    """n = bytes per logical frame (multiple of 24) -> uint8[5 n] super frame, list of AU payloads"""
    n_rs = 5 * n // 120
    assert n_rs * 120 == 5 * n
    data_len = 110 * n_rs
    num_aus = {(0, 1): 2, (1, 1): 3, (0, 0): 4, (1, 0): 6}[(dac_rate, sbr_flag)]
    nb_start_bytes = (12 * (num_aus - 1) + 7) // 8
    first = 3 + nb_start_bytes
    # split the remaining data bytes into num_aus access units (each >= 3 bytes: payload + CRC)
    room = data_len - first
    cuts = np.sort(rng.choice(np.arange(3, room - 3), num_aus - 1, replace=False)) if num_aus > 1 else np.array([], int)
    while num_aus > 1 and np.any(np.diff(np.concatenate([[0], cuts, [room]])) < 3):
        cuts = np.sort(rng.choice(np.arange(3, room - 3), num_aus - 1, replace=False))
    starts = [first] + [first + int(c) for c in cuts] + [data_len]
    sf = np.zeros(5 * n, np.uint8)
    sf[2] = (dac_rate << 6) | (sbr_flag << 5) | (channel_mode << 4) | (ps_flag << 3) | mpeg_config
    bits = []
    for v in starts[1:num_aus]:
        bits += [(v >> (11 - b)) & 1 for b in range(12)]
    bits += [0] * (8 * nb_start_bytes - len(bits))
    sf[3:3 + nb_start_bytes] = np.packbits(np.array(bits, np.uint8))
    aus = []
    for i in range(num_aus):
        a, b = starts[i], starts[i + 1]
        payload = rng.integers(0, 256, b - a - 2, dtype=np.uint8)
        crc = crc16_ccitt(payload)
        if i in bad_au_crc:
            crc ^= 0x0101
        sf[a:b - 2] = payload
        sf[b - 2] = crc >> 8
        sf[b - 1] = crc & 0xFF
        aus.append(payload)
    fc = oracle.firecode_crc(sf[2:11])
    sf[0] = fc >> 8
    sf[1] = fc & 0xFF
    # RS(120,110) over the interleaved columns: codeword i = sf[i + j n_rs], j = 0..119
    for i in range(n_rs):
        col = sf[i:i + 110 * n_rs:n_rs]
        sf[i + 110 * n_rs::n_rs] = oracle.rs120_encode(col)
    return sf, aus, starts


def corrupt(rng, sf, n_rs_errors_per_codeword):
    """flip the given number of distinct symbols in each RS codeword (dict codeword index -> count, or int for all)"""
    sf = sf.copy()
    n_rs = sf.size // 120
    for i in range(n_rs):
        k = n_rs_errors_per_codeword.get(i, 0) if isinstance(n_rs_errors_per_codeword, dict) else n_rs_errors_per_codeword
        for j in rng.choice(120, k, replace=False):
            sf[i + int(j) * n_rs] ^= rng.integers(1, 256)
    return sf
