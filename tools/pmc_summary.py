#!/usr/bin/env python3
"""Join the two single-counter rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: tools/pmc.sh) into per-kernel HBM traffic.

usage: tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [train steps in the profiled run, default 3]

Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes, so reads are doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
Dispatches of the two passes are matched by order (same deterministic program) and checked by kernel name.
"""
import collections
import csv
import glob
import json
import re
import sys

FAMILIES = {       # kernel name -> op family used by bench.py's roofline ("_ln": launches with the fused LayerNorm epilogue)
    "conv3x3_fwd_wres_kernel": "conv3x3_fwd", "conv3x3_fwd_ws_kernel": "conv3x3_fwd", "conv3x3_fwd_kernel": "conv3x3_fwd",
    "splitk_finalize_kernel": "conv3x3_fwd", "conv3x3_map4_kernel": "conv3x3_fwd", "conv3x3_map1_kernel": "conv3x3_fwd",
    "conv3x3_fwd_wres_kernel_ln": "conv3x3_ln_relu_fwd", "conv3x3_fwd_ws_kernel_ln": "conv3x3_ln_relu_fwd",
    "conv3x3_fwd_wres_kernel_relugrad": "conv3x3_dgrad_relu", "conv3x3_fwd_wres_kernel_lnbwd": "conv3x3_dgrad_ln_bwd",
    "conv3x3_wgrad_kernel": "conv3x3_wgrad", "conv3x3_wgrad_ws_kernel": "conv3x3_wgrad", "wgrad_reduce_kernel": "conv3x3_wgrad",
    # the dedicated first-layer kernels belong to the families bench.py books them in (r04: traffic and algorithmic bytes of a
    # family are quoted over the same launches)
    "conv3x3_c3_fwd_kernel": "conv3x3_ln_relu_fwd", "conv3x3_c3_wgrad_kernel": "conv3x3_wgrad",
    "pw_gemm_lds_kernel": "upconv_bank_gemms",
    "pw_gemm_kernel": "upconv_bank_gemms", "pw_wgrad_kernel": "upconv_bank_gemms", "pw_wgrad_reduce_kernel": "upconv_bank_gemms",
}
HELPERS = {"splitk_finalize_kernel", "wgrad_reduce_kernel", "pw_wgrad_reduce_kernel"}   # counted with the launch they finish


def short(name):
    if re.search(r"conv3x3_fwd_w(res|s)_kernel", name) and ("Li2EEE" in name or re.search(r", 2>", name)):
        return re.search(r"conv3x3_fwd_w(?:res|s)_kernel", name).group(0) + "_ln"
    if re.search(r"conv3x3_fwd_wres_kernel", name) and ("Li3EEE" in name or re.search(r", 3>", name)):
        return "conv3x3_fwd_wres_kernel_relugrad"
    if re.search(r"conv3x3_fwd_wres_kernel", name) and ("Li4EEE" in name or re.search(r", 4>", name)):
        return "conv3x3_fwd_wres_kernel_lnbwd"
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)", name)
    if m:
        return m.group(1)
    m = re.match(r"_Z\d+([a-z0-9_]+?_kernel)", name)
    if m:
        return m.group(1)
    return re.sub(r"[<(].*", "", name)


def load(d):
    return list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))


def main():
    fetch, write, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    f, w = load(fetch), load(write)
    assert len(f) == len(w), "the two passes dispatched different numbers of kernels"
    per = collections.OrderedDict()
    for a, b in zip(f, w):
        assert a["Kernel_Name"] == b["Kernel_Name"] and a["Counter_Name"] == "FETCH_SIZE" and b["Counter_Name"] == "WRITE_SIZE"
        k = per.setdefault(short(a["Kernel_Name"]), {"launches": 0, "fetch_kib_raw": 0.0, "write_kib": 0.0})
        k["launches"] += 1
        k["fetch_kib_raw"] += float(a["Counter_Value"])
        k["write_kib"] += float(b["Counter_Value"])
    fam = collections.OrderedDict()
    for name, k in per.items():
        k["hbm_bytes_per_launch"] = (2.0 * k["fetch_kib_raw"] + k["write_kib"]) * 1024.0 / k["launches"]
        if name in FAMILIES:
            g = fam.setdefault(FAMILIES[name], {"launches": 0, "hbm_bytes": 0.0})
            if name not in HELPERS:
                g["launches"] += k["launches"]          # helpers count towards the bytes of the launch they finish
            g["hbm_bytes"] += (2.0 * k["fetch_kib_raw"] + k["write_kib"]) * 1024.0
    for g in fam.values():
        g["hbm_bytes_per_launch"] = g["hbm_bytes"] / g["launches"]
    # the kernel sources these passes were taken on: bench.py quotes the figure only while that stamp still matches
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # whole-step traffic: every dispatch of the run except torch's own fill / copy kernels of the set-up, over the steps
    skip = ("at::native", "__amd_rocclr", "pack_kernel")
    step_bytes = sum((2.0 * k["fetch_kib_raw"] + k["write_kib"]) * 1024.0 for n, k in per.items() if not n.startswith(skip)) / steps
    json.dump({"correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950, MI355X_MICROARCH.md)",
               "kernel_source_stamp": bench.kernel_source_stamp(), "steps_profiled": steps, "hbm_bytes_per_step": step_bytes,
               "kernels": per, "families": fam}, open(out, "w"), indent=1)
    print(f"HBM traffic per train step: {step_bytes / 1e9:.2f} GB ({steps} steps profiled)")
    for name, k in sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{name:34s} launches={k['launches']:4d}  {k['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
