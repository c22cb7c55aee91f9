"""Builds executables out of the REFERENCE's own callers + this repository's mirror classes, in the build container only.

INTEGRATION.md's recipe, mechanised: a scratch COPY of /root/reference/src gets the mirror's headers and sources laid over the files they
replace and everything is compiled THERE, against the reference's own utility/span.h, observable.h, database entities and constants.
Nothing of the reference is copied into the repository, and nothing built here travels (the GPU box has no /root/reference): what
travels is the data these executables produce, committed as fixtures under tests/golden/ together with the scripts that made them.

`link(..., backend="oracle")` links against tests/cpp/fake_dabgpu_oracle.cpp -- the C ABI of include/dabgpu.h implemented by the CPU oracle --
so the reference's control code can be EXECUTED without a device; backend="gpu" links against libdabgpu.so (link check only, here)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "dab-radio_amd", "host")
CSRC = os.path.join(ROOT, "dab-radio_amd", "csrc")
ORACLE = os.path.join(ROOT, "oracle")
MIRROR_HEADERS = ["ofdm/ofdm_demodulator.h", "dab/fic/fic_decoder.h", "dab/msc/msc_decoder.h", "dab/msc/cif_deinterleaver.h",
                  "dab/algorithms/dab_viterbi_decoder.h", "dab/audio/aac_frame_processor.h"]
MIRROR_SOURCES = ["ofdm/ofdm_demodulator.cpp", "dab/fic/fic_decoder.cpp", "dab/msc/msc_decoder.cpp", "dab/msc/cif_deinterleaver.cpp",
                  "dab/algorithms/dab_viterbi_decoder.cpp", "dab/audio/aac_frame_processor.cpp",
                  "dab/dabgpu_frame_batcher.cpp", "dab/dabgpu_frame_batcher.h", "dab/dabgpu_shared_context.cpp", "dab/dabgpu_shared_context.h"]
ORACLE_SRCS = ["dab_oracle_ofdm.c", "dab_oracle_decode.c", "dab_oracle_io.c", "dab_oracle_dabplus.c", "dab_oracle_chain.c"]
# the reference's own translation units behind BasicFICRunner and OFDM_Block (compiled from the overlay, i.e. from the reference's text)
REF_FIC_SIDE = ["basic_radio/basic_fic_runner.cpp", "dab/fic/fig_processor.cpp", "dab/radio_fig_handler.cpp", "dab/database/dab_database_updater.cpp",
                "dab/dab_logging.cpp", "dab/constants/charsets.cpp"]
REF_OFDM_SIDE = ["ofdm/dab_ofdm_params_ref.cpp", "ofdm/dab_prs_ref.cpp", "ofdm/dab_mapper_ref.cpp"]


def fmt_include():
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "include")
        if os.path.exists(os.path.join(p, "fmt", "format.h")):
            return p
    except Exception:
        pass
    return None


def available():
    """(ok, reason)"""
    if not os.path.isdir(os.path.join(REF, "src")):
        return False, "/root/reference is not present here"
    if fmt_include() is None:
        return False, "no header-only fmt available (the reference's vendor/fmt submodule is empty)"
    return True, ""


class Overlay:
    def __init__(self, workdir):
        self.dir = str(workdir)
        self.src = os.path.join(self.dir, "src")
        self.out = os.path.join(self.dir, "obj")
        self.fmt = fmt_include()
        shutil.copytree(os.path.join(REF, "src"), self.src)
        for h in MIRROR_HEADERS + MIRROR_SOURCES:
            shutil.copyfile(os.path.join(HOST, h), os.path.join(self.src, h))
        os.makedirs(self.out, exist_ok=True)
        self._objs = {}

    def compile(self, path, extra=(), opt="-O1"):
        """path relative to the overlay's src/ or absolute"""
        full = path if os.path.isabs(path) else os.path.join(self.src, path)
        key = (full, tuple(extra))
        if key in self._objs:
            return self._objs[key]
        obj = os.path.join(self.out, "%03d_" % len(self._objs) + os.path.basename(full).rsplit(".", 1)[0] + ".o")
        cc = ["gcc", "-std=gnu11", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w", "-mavx2", "-mbmi2", "-mfma"] if full.endswith(".c") else \
             ["g++", "-std=c++17", opt, "-DFMT_HEADER_ONLY", "-I" + self.src, "-I" + os.path.join(REF, "examples"), "-I" + os.path.join(ROOT, "include"), "-I" + self.fmt,
              "-I" + CSRC, "-I" + ORACLE]
        res = subprocess.run(cc + list(extra) + ["-c", full, "-o", obj], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (path, res.stderr[-4000:])
        self._objs[key] = obj
        return obj

    def mirror_objects(self, exclude=()):
        return [self.compile(f) for f in MIRROR_SOURCES if f.endswith(".cpp") and f not in exclude]

    def oracle_backend_objects(self):
        return [self.compile(os.path.join(ORACLE, s)) for s in ORACLE_SRCS] + \
               [self.compile(os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"), opt="-O2"), self.compile(os.path.join(CSRC, "dabgpu_host_logic.cpp"), opt="-O2")]

    def link(self, objs, name, backend="oracle"):
        exe = os.path.join(self.dir, name)
        if backend == "oracle":
            cmd = ["g++", *objs, *self.oracle_backend_objects(), "-pthread", "-lm", "-o", exe]
        else:
            lib_dir = os.path.join(ROOT, "dab-radio_amd")
            cmd = ["g++", *objs, "-L" + lib_dir, "-ldabgpu", "-Wl,-rpath," + lib_dir, "-pthread", "-o", exe]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, "unresolved symbols between the reference's callers and the mirror classes:\n" + res.stderr[-4000:]
        return exe

    def callers_driver(self, backend="oracle"):
        """tests/cpp/ref_callers_driver.cpp: app_iq_readers -> OFDM_Block -> ThreadedRingBuffer -> BasicFICRunner + FIG stack + database -> lazily created MSC decoders"""
        objs = [self.compile(f) for f in REF_FIC_SIDE + REF_OFDM_SIDE]
        objs += [self.compile(os.path.join(ROOT, "tests", "cpp", "ref_callers_driver.cpp"))] + self.mirror_objects(
            exclude=("dab/audio/aac_frame_processor.cpp",))      # not on this driver's path (the oracle-backed ABI has no DAB+ bank)
        return self.link(objs, "ref_callers_driver_" + backend, backend)
