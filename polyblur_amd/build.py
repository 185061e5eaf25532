"""Build libpolyblur_hip.so in-tree with hipcc for gfx950 (no cmake, no torch extension).

    python -m polyblur_amd.build [--force] [--experimental]

The library lands in polyblur_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libpolyblur_hip.so"
SOURCES = ["api.hip", "comm.hip", "conv.hip", "conv_fft.hip", "conv_big.hip", "conv_wfft.hip", "conv_w128.hip", "estimate.hip", "lines_fixed.hip", "filters.hip", "nc.hip"]
# measured experiments that are not part of the product (NOTEBOOK.md): python -m polyblur_amd.build --experimental
EXPERIMENTAL_SOURCES = ["conv_strip.hip", "conv_xt.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experimental: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = hipcc_path()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "polyblur_hip.h"))
    jobs = []
    objs = []
    extra = os.environ.get("PB_EXTRA_FLAGS", "").split()
    flags = FLAGS + extra
    sources = SOURCES + (EXPERIMENTAL_SOURCES if experimental else [])
    if experimental:
        flags = flags + ["-DPB_EXPERIMENTAL"]
    # objects of another flavour (debug / trace / experimental flags) never meet the default library's: one object
    # directory per flag set, and the library is relinked whenever the flavour of the last link differs
    flavour = hashlib.sha256(" ".join(flags + sources).encode()).hexdigest()[:10]
    objdir = os.path.join(LIBDIR, "obj" if not (extra or experimental) else "obj-" + flavour)
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(LIBDIR, "flavour.txt")
    relink = not os.path.exists(stamp) or open(stamp).read().strip() != flavour
    for src in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    out = lib_path()
    if jobs or relink or not os.path.exists(out):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", out] + objs + ["-ldl"])
        with open(stamp, "w") as f:
            f.write(flavour + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experimental="--experimental" in sys.argv))
