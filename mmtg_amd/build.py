"""Build libmmtg_hip.so (gfx950 only) with hipcc, in-tree.

    python -m mmtg_amd.build [--force] [--jobs N]

Objects go to mmtg_amd/csrc/_build/*.o, the library to mmtg_amd/libmmtg_hip.so
(git-ignored, but shipped to the GPU box by gpurun).  hipcc cross-compiles
without a GPU present.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libmmtg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
EXTRA = os.environ.get("MMTG_EXTRA_DEFS", "").split()     # (diagnostic builds: -DMMTG_P8_PHASE_TRACE)
# the extra defines are compiled INTO the library (mmtg_build_flags()): hip.lib() refuses a diagnostic build unless
# MMTG_ALLOW_DIAGNOSTIC_BUILD=1, so it cannot be mistaken for the product by a later run, bench.py or a committed profile
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-DNDEBUG", "-fvisibility=hidden", '-DMMTG_BUILD_FLAGS="%s"' % " ".join(EXTRA)] + EXTRA


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_hash(src):
    h = hashlib.sha1()
    for name in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "mmtg_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp = obj + ".sha1"
    digest = _deps_hash(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(digest)
    return obj, True


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did in ex.map(lambda s: _compile(s, force), srcs):
            objs.append(obj)
            rebuilt |= did
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
               "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    build(a.force, a.jobs)
