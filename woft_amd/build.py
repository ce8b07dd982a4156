"""Build libwoft_hip.so (gfx950) in-tree with hipcc.  `python -m woft_amd.build [--force]`."""
import hashlib
import os
import subprocess
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIBDIR = HERE / "lib"
LIB = LIBDIR / "libwoft_hip.so"
STAMP = LIBDIR / "libwoft_hip.stamp"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
         "-Wno-c++20-extensions"]


# Translation units.  The two template-heavy sources are compiled once per arithmetic mode (-DWOFT_ONLY_PREC = the precision code
# of woft_conv_params: 1 bf16x3, 2 bf16 (+ the exact-fp32 kernels), 3 fp16, 4 f16mx8 (conv_regb.hip only)): hipcc compiles a unit on
# ONE core (conv.hip as a single unit: 9+ minutes), the parts run side by side.  Every part exports its own dispatcher symbol
# (woft_conv_dispatch_p<k>, woft_conv_regb_launch_p<k>); the C ABI entry points live in part 1 of conv.hip.
PARTS = {"conv.hip": [("p1", ["-DWOFT_ONLY_PREC=1"]), ("p2", ["-DWOFT_ONLY_PREC=2"]), ("p3", ["-DWOFT_ONLY_PREC=3"])],
         "conv_regb.hip": [(f"p{k}", [f"-DWOFT_ONLY_PREC={k}"]) for k in (1, 2, 3, 4)]}


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _units():
    """-> [(source, object, extra flags)]"""
    out = []
    for src in _sources():
        for tag, flags in PARTS.get(src.name, [("", [])]):
            out.append((src, LIBDIR / (src.stem + ("." + tag if tag else "") + ".o"), flags))
    return out


def _digest():
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "woft_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PARTS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    LIBDIR.mkdir(exist_ok=True)
    dig = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB
    objs = []
    units = _units()
    wanted = {u[1] for u in units}
    for old in LIBDIR.glob("*.o"):            # (objects of an earlier unit list must not be linked)
        if old not in wanted:
            old.unlink()
    # per-unit stamps: a unit is recompiled when its source, any header or its flags changed (an edit of conv_regb.hip leaves the
    # three 9-minute parts of conv.hip alone)
    hdr = hashlib.sha256()
    for h in sorted(list(CSRC.glob("*.h")) + [HERE.parent / "include" / "woft_hip.h"]):
        hdr.update(h.name.encode())
        hdr.update(h.read_bytes())

    def unit_hash(src, extra):
        u = hashlib.sha256(hdr.digest())
        u.update(src.read_bytes())
        u.update(" ".join(FLAGS + extra).encode())
        return u.hexdigest()
    # longest compiles first (measured on this image: a part of conv.hip ~4 min on one core, of conv_regb.hip ~2,
    # conv_1x1.hip ~2, everything else under a minute); at most `jobs` compilers at a time
    jobs = int(os.environ.get("WOFT_BUILD_JOBS", os.cpu_count() or 4))
    weight = {"conv.hip": 8, "conv_regb.hip": 6, "conv_1x1.hip": 20}
    units.sort(key=lambda u: -(u[0].stat().st_size * weight.get(u[0].name, 1)))
    pending, running = [], []
    for src, obj, extra in units:
        objs.append(obj)
        uh = unit_hash(src, extra)
        tag = Path(str(obj) + ".hash")
        if not force and obj.exists() and tag.exists() and tag.read_text().strip() == uh:
            continue
        pending.append((src, obj, extra, uh))
    while pending or running:
        while pending and len(running) < jobs:
            src, obj, extra, uh = pending.pop(0)
            tag = Path(str(obj) + ".hash")
            if tag.exists():
                tag.unlink()
            cmd = [HIPCC, *FLAGS, *extra, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((src, subprocess.Popen(cmd), tag, uh))
        time.sleep(0.2)
        for ent in list(running):
            rc = ent[1].poll()
            if rc is None:
                continue
            running.remove(ent)
            if rc != 0:
                for other in running:
                    other[1].kill()
                raise RuntimeError(f"hipcc failed on {ent[0]}")
            ent[2].write_text(ent[3])
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    STAMP.write_text(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
