"""Generates tests/golden/decoder_vectors.npz with the REFERENCE's own FIC_Decoder::DecodeFIBGroup (src/dab/fic/fic_decoder.cpp:53-117),
MSC_Decoder::DecodeCIF / DecodeEEP / DecodeUEP (src/dab/msc/msc_decoder.cpp:46-154) and CIF_Deinterleaver (src/dab/msc/cif_deinterleaver.cpp) --
compiled in place into oracle/_ref/libdab_ref_decoders.so -- executed over the inputs of tests/decoder_vectors.py (build container only).

    python tests/golden/make_golden_decoders.py

LABEL (stored in the fixture): "reference control code over the RESTATED Viterbi core".  The class those decoders call, DAB_Viterbi_Decoder, is
declared by the reference's header but defined in oracle/ref_harness_decoders.cpp over oracle/dab_oracle_decode.c, because the reference's own
definition needs the absent vendor/viterbi_decoder.  The vectors therefore pin which symbols reach the core, with which puncturing vectors, segment
lengths and padding, in which order, and what happens to the decoded bytes (rows a20-a22 of SURVEY 8) to executed reference code; the
add-compare-select core and the de-puncturing loop stay restated ("parity unpinned").  Both core models (scalar / SIMD) are stored; the second as
XOR against the first (they differ on exact metric ties only)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
LABEL = "reference control code (fic_decoder.cpp, msc_decoder.cpp, cif_deinterleaver.cpp compiled in place) over the RESTATED Viterbi core " \
        "(oracle/dab_oracle_decode.c behind the reference's dab_viterbi_decoder.h): pins rows a20-a22, not the add-compare-select core"


def run_reference(O, R, DV, model, soft_fic, cs, cifs):
    """-> (n valid FIBs per group, FIB bytes [groups][90], per case the list of DecodeCIF results (bytes objects, b'' while the de-interleaver fills))"""
    R.ref_dec_set_core_model(model)
    fic = R.ref_fic_create(2304, 3)
    n = np.zeros(len(soft_fic), np.int32)
    fibs = np.zeros((len(soft_fic), 90), np.uint8)
    for g, s in enumerate(soft_fic):
        s = np.ascontiguousarray(s)
        n[g] = R.ref_fic_decode_group(fic, s.ctypes.data, s.size, g % 4, fibs[g].ctypes.data, 90)
    R.ref_fic_destroy(fic)
    msc = []
    for c in cs:
        h = R.ref_msc_create(c["index"] % 64, c["start"], c["length"], c["is_uep"], c["uep_index"], c["eep_level"], c["eep_type"])
        out = np.zeros(c["length"] * 8, np.uint8)
        rec = []
        for t in range(cifs.shape[1]):
            cif = np.ascontiguousarray(cifs[c["mux"], t])
            k = R.ref_msc_decode_cif(h, cif.ctypes.data, cif.size, out.ctypes.data, out.size)
            assert k >= 0
            rec.append(out[:k].tobytes())
        R.ref_msc_destroy(h)
        msc.append(rec)
    return n, fibs, msc


def make(O, R):
    import decoder_vectors as DV
    cs, cifs, payload = DV.msc_multiplexes(O)
    soft_fic, _ = DV.fic_groups(O)
    d = dict(label=np.array(LABEL), n_cifs=np.int64(DV.N_CIFS), fic_inputs_sha256=np.array(DV.sha(soft_fic)), msc_inputs_sha256=np.array(DV.sha(cifs)),
             case_names=np.array([c["name"] for c in cs]),
             case_table=np.array([[c["mux"], c["start"], c["length"], c["is_uep"], c["uep_index"], c["eep_level"], c["eep_type"], c["setting"]] for c in cs], np.int32))
    res = [run_reference(O, R, DV, m, soft_fic, cs, cifs) for m in (0, 1)]
    d["fic_n_valid_scalar"], d["fic_fibs_scalar"] = res[0][0], res[0][1]
    d["fic_n_valid_simd"], d["fic_fibs_simd_xor_scalar"] = res[1][0], res[1][1] ^ res[0][1]
    lens = np.array([[len(b) for b in rec] for rec in res[0][2]], np.int32)                      # [case][cif] bytes returned
    assert np.array_equal(lens, np.array([[len(b) for b in rec] for rec in res[1][2]], np.int32))
    d["msc_lengths"] = lens
    blob0 = np.frombuffer(b"".join(b for rec in res[0][2] for b in rec), np.uint8)
    blob1 = np.frombuffer(b"".join(b for rec in res[1][2] for b in rec), np.uint8)
    d["msc_bytes_scalar"], d["msc_bytes_simd_xor_scalar"] = blob0, blob0 ^ blob1
    return d


if __name__ == "__main__":
    import oracle as O
    R = O.ref_decoders()
    assert R is not None, "oracle/_ref/libdab_ref_decoders.so missing: needs /root/reference (make -C oracle ref)"
    d = make(O, R)
    np.savez_compressed(os.path.join(HERE, "decoder_vectors.npz"), **d)
    print("cases", len(d["case_names"]), "valid FIBs", int(d["fic_n_valid_scalar"].sum()), "MSC bytes", d["msc_bytes_scalar"].size,
          "bytes that differ between the core models", int(np.count_nonzero(d["msc_bytes_simd_xor_scalar"])),
          "->", os.path.getsize(os.path.join(HERE, "decoder_vectors.npz")), "bytes")
