"""Builds the gfx950 shared library (C ABI of include/adunet.h) in-tree with hipcc.

    python -m adunet_amd.build        (or __graft_entry__.build())

hipcc cross-compiles without a GPU; the resulting csrc/libadunet_hip.so travels to the GPU box
with the repository snapshot.  Objects are rebuilt only when a source is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
LIB = os.path.join(CSRC, "libadunet_hip.so")
SOURCES = ["api.hip", "conv.hip", "norm.hip", "resize.hip", "head.hip", "optim.hip", "tier2.hip", "metrics.hip", "comm.hip", "upconv.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "adunet.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = sp[:-4] + ".o"
        newest = max(os.path.getmtime(p) for p in [sp] + deps)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            extra = os.environ.get("AD_CFLAGS", "").split()   # e.g. -DAD_STAMP for the phase-stamp diagnostic build
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"] + extra + ["-c", sp, "-o", obj])
        objs.append(obj)
    if jobs:            # the translation units are independent: compile them side by side (conv.hip alone takes ~40 s)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4, 8)) as pool:
            list(pool.map(run, jobs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
