"""Build recipe for libmwrt.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mwrt.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "mwrt_kernels.hip.h"), os.path.join(ROOT, "include", "mwrt.h")]
LIB = os.path.join(HERE, "libmwrt.so")


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


def build_native(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile csrc/mwrt.hip -> libmwrt.so for gfx950.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), *extra_flags, "-o", LIB + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
