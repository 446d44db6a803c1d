"""Build libwmar_hip.so (gfx950) in-tree with hipcc.  `python -m wmar_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwmar_hip.so")

# (source, extra flags).  watermark.hip carries the pinned sampling arithmetic
# (include/wmar_math.h): no fp contraction there.
SOURCES = [
    ("keytable.cpp", []),
    ("comm.cpp", []),
    ("watermark.hip", ["-ffp-contract=off"]),
    ("gumbel.hip", ["-ffp-contract=off"]),
    ("gpt.hip", []),
    ("rar.hip", []),
    ("cham.hip", []),
    ("vqgan.hip", []),
    ("augment.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-x", "hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libwmar_hip.so)")


def _deps(src: str):
    d = [os.path.join(CSRC, src)]
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            d.append(os.path.join(CSRC, f))
    inc = os.path.join(HERE, "..", "include")
    for f in os.listdir(inc):
        d.append(os.path.join(inc, f))
    return d


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    # development builds only (tuning scripts under scripts/): environment-variable knobs compiled in
    dev = ["-DWMAR_DEV_KNOBS"] if os.environ.get("WMAR_DEV_KNOBS") else []
    dev += os.environ.get("WMAR_EXTRA_HIPCC_FLAGS", "").split()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [(s, f) for s, f in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for src, flags in srcs:
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc] + COMMON + flags + dev + ["-c", os.path.join(CSRC, src), "-o", obj]
        stamp = obj + ".cmd"       # the command line the object was built with: a change of flags (dev knobs) rebuilds it
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        stale = force or not same_cmd or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        if stale:
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(cmd[-1] + ".cmd", "w") as fh:
            fh.write(" ".join(cmd))
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s, _ in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
