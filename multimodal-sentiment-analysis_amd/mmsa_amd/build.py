"""Compile csrc/*.hip for gfx950 into mmsa_amd/libmmsa_hip.so (in-tree, plain hipcc, no torch headers)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG_DIR), "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmmsa_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
SOURCES = ("frontend.hip", "denoise.hip", "conv.hip", "wgrad.hip", "audio_ew.hip", "text.hip", "dense.hip", "head.hip", "optim.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = tuple(os.path.join(CSRC, h) for h in ("mmsa_common.h", "fft512.h", "conv_rows3.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, (s, *headers)):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        # link inside the object directory (hipcc leaves its unbundled temporaries beside the output), then move
        tmp = os.path.join(OBJ_DIR, "libmmsa_hip.so")
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp])
        os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
