"""The CPU double of the C ABI's host side (VERDICT r03 "missing" 5): the library built with
AddressSanitizer + UndefinedBehaviorSanitizer on its HOST code, driven without a GPU through every entry point that answers
from host arithmetic -- tile geometry (pick_geo), split-K and weight-gradient split planning, workspace sizes, the 2 GiB
image-run chunking, the gathers' staging rule -- over thousands of arbitrary shapes, plus launch entry points called with
arguments they must reject.  The first run of this test (r04) found a division by zero and three 32-bit overflows in shape
queries (tools/host_sanitizer_fuzz.py lists them)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adaptive-depth-u-net-for-image-super-resolution-segmentation_amd", "csrc")
ASAN = "/opt/rocm/lib/llvm/lib/clang"


def _asan_runtime():
    for root, _, files in os.walk(ASAN):
        if "libclang_rt.asan-x86_64.so" in files:
            return os.path.join(root, "libclang_rt.asan-x86_64.so")
    return None


def test_host_side_of_the_c_abi_under_asan_and_ubsan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rt = _asan_runtime()
    if not (os.path.exists(hipcc) and rt):
        pytest.skip("hipcc or clang's sanitizer runtime is missing")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    flags = [hipcc, "--offload-arch=gfx950", "-O1", "-g1", "-fPIC", "-std=c++17", "-Wno-unused-value",
             "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=undefined"]

    def compile_one(src):
        obj = str(tmp_path / (src[:-4] + ".o"))
        subprocess.run(flags + ["-c", os.path.join(CSRC, src), "-o", obj], check=True, capture_output=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, srcs))
    lib = str(tmp_path / "libadunet_san.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", lib]
                   + objs + ["-ldl"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt, ADUNET_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_sanitizer_fuzz.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    report = [l for l in (run.stdout + run.stderr).splitlines() if "runtime error" in l or "AddressSanitizer" in l]
    assert run.returncode == 0 and not report and run.stdout.strip().endswith("ok"), (report[:3], run.stderr[-1500:])
