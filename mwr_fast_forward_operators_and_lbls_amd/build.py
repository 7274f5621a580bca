"""Build recipe for libmwrt.so (hipcc, gfx950 only, in-tree so the .so travels with the repo).

Four translation units -- the C-ABI host side (csrc/mwrt.hip) and the kernel instantiations of each
frequency-chunk width (csrc/mwrt_inst.hip with -DMWRT_INST_NFC=8|14|16) -- are compiled in parallel and
linked into one shared library."""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "mwrt.hip")
INST = os.path.join(CSRC, "mwrt_inst.hip")
DEPS = [SRC, INST, os.path.join(CSRC, "mwrt_kernels.hip.h"), os.path.join(CSRC, "mwrt_inst.hip.h"),
        os.path.join(ROOT, "include", "mwrt.h")]
LIB = os.path.join(HERE, "libmwrt.so")
OBJ_DIR = os.path.join(HERE, "build")
INST_WIDTHS = (8, 14, 16)


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; no other compiler can build the gfx950 kernels)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_native(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile the translation units for gfx950 and link libmwrt.so (or ``out``).  Returns the library path."""
    lib = out or LIB
    if not force and out is None and not is_stale():
        return LIB
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = str(os.getpid())
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              *extra_flags]
    jobs = [(SRC, os.path.join(OBJ_DIR, f"mwrt.{tag}.o"), [])]
    jobs += [(INST, os.path.join(OBJ_DIR, f"mwrt_inst{n}.{tag}.o"), [f"-DMWRT_INST_NFC={n}"]) for n in INST_WIDTHS]

    def compile_one(job):
        src, obj, defs = job
        cmd = common + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, jobs))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp", *objs]
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.run(link, check=True)
        os.replace(lib + ".tmp", lib)
    finally:
        for _, obj, _ in jobs:
            if os.path.exists(obj):
                os.remove(obj)
    return lib


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
