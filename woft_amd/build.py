"""Build libwoft_hip.so (gfx950) in-tree with hipcc.  `python -m woft_amd.build [--force]`."""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIBDIR = HERE / "lib"
LIB = LIBDIR / "libwoft_hip.so"
STAMP = LIBDIR / "libwoft_hip.stamp"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
         "-Wno-c++20-extensions"]


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest():
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "woft_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    LIBDIR.mkdir(exist_ok=True)
    dig = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB
    objs = []
    procs = []
    for src in _sources():
        obj = LIBDIR / (src.stem + ".o")
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    STAMP.write_text(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
