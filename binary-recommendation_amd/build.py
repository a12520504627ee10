"""Build libbinrec_hip.so (the C-ABI hot-path library) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
Each translation unit is compiled to an object in parallel, then linked.
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
LIB_NAME = "libbinrec_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libbinrec_hip.so cannot be built (ROCm toolchain required)")


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(paths: list[str]) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())     # (not the absolute path: the tree is copied to another root on the GPU box, where a
        with open(p, "rb") as f:                   #  path-dependent digest made every build() there recompile all sources)
            h.update(f.read())
    h.update(" ".join(f for f in FLAGS if not os.path.isabs(f)).encode())
    return h.hexdigest()


def _headers() -> list[str]:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "binrec.h"))
    return hs


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_digest = _digest(_headers())
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        stamp = obj + ".sha"
        want = _digest([src]) + hdr_digest
        objs.append(obj)
        have = open(stamp).read() if os.path.exists(stamp) and os.path.exists(obj) else ""
        if force or have != want:
            jobs.append((src, obj, stamp, want))

    def compile_one(job):
        src, obj, stamp, want = job
        cmd = [hipcc, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
        return src

    if jobs:
        if verbose:
            print(f"[binrec build] compiling {len(jobs)} source(s) for {ARCH} ...", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB_PATH):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[binrec build] linked {LIB_PATH}", file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
