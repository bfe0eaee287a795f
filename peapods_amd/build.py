"""Builds libpeapods_amd.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m peapods_amd.build            # hipcc cross-compiles without a GPU

Every translation unit is compiled to its own object (in parallel, only when it or a header
changed), then linked.  The library is the only compiled artefact of the product; the parity
oracle is built separately by oracle/Makefile and never linked here.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libpeapods_amd.so")
SOURCES = ["host_util.cpp", "kernels.hip", "kernels_general.hip", "kernels_cluster.hip", "kernels_reg.hip", "sim.hip", "comm.hip", "bandwidth.hip"]
HEADERS = ["engine.hpp", "host_util.hpp", "device_util.hpp", "bitcount.hpp", os.path.join("..", "..", "include", "peapods_amd.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # no fused multiply-add anywhere: the reference never fuses
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-result",
]
# RCCL is bound at run time (dlopen in comm.hip), so a single-GPU box without it still loads the library
LINK_FLAGS = ["-shared", "-ldl", "-lpthread"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the peapods_amd extension cannot be built")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    built = os.path.getmtime(target)
    return any(os.path.getmtime(d) > built for d in deps)


def _header_paths():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def source_hash():
    """sha256 (12 hex digits) over the product sources in a fixed order: what libpeapods_amd.so was built from.
    The library reports it (pa_version), profiles/counters.json records it at collection time, and bench.py prints
    counter-derived figures only when the two agree."""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:12]


STAMP = os.path.join(OBJ, "build_stamp.txt")


def _stamp(extra_flags):
    return source_hash() + " " + " ".join(extra_flags)


def needs_build(extra_flags=()):
    """The library is missing, older than a source, or was built from other sources / with other flags (a
    variant build with extra_flags leaves its stamp, so the next plain build() relinks the plain library)."""
    if _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + _header_paths()):
        return True
    try:
        with open(STAMP) as fh:
            return fh.read() != _stamp(extra_flags)
    except OSError:
        return True


def build(force=False, verbose=False, extra_flags=()):
    extra_flags = tuple(extra_flags)
    if not force and not needs_build(extra_flags):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    tag = "".join(c if c.isalnum() else "_" for c in "".join(extra_flags))
    digest = source_hash()
    jobs = []
    objects = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        # host_util.cpp carries pa_version(): its object is named after the source hash, so it is rebuilt (3 s)
        # whenever any source changed
        suffix = tag + ("_" + digest if src == "host_util.cpp" else "")
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + suffix + ".o")
        objects.append(obj)
        defines = [f'-DPA_SOURCE_HASH="{digest}"'] if src == "host_util.cpp" else []
        if force or _newer(obj, [path] + _header_paths()):
            jobs.append([cc, *FLAGS, *extra_flags, *defines, "-c", path, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    run([cc, "--offload-arch=gfx950", "-fPIC", *LINK_FLAGS, "-o", LIB, *objects])
    for name in os.listdir(OBJ):  # objects of older source hashes
        if name.startswith("host_util") and name.endswith(".o") and os.path.join(OBJ, name) not in objects:
            os.remove(os.path.join(OBJ, name))
    with open(STAMP, "w") as fh:
        fh.write(_stamp(extra_flags))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
