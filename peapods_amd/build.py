"""Builds libpeapods_amd.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m peapods_amd.build            # hipcc cross-compiles without a GPU

The library is the only compiled artefact of the product; the parity oracle is
built separately by oracle/Makefile and never linked here.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpeapods_amd.so")
SOURCES = ["host_util.cpp", "kernels.hip", "kernels_general.hip", "kernels_cluster.hip", "sim.hip"]
HEADERS = ["engine.hpp", "host_util.hpp", "device_util.hpp", os.path.join("..", "..", "include", "peapods_amd.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",  # no fused multiply-add anywhere: the reference never fuses
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-result",
]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the peapods_amd extension cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, *extra_flags, "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
