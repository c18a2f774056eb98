#!/bin/bash
# After `gpurun -- tools/evidence.sh <tag>` has merged gpurun_out/<tag>/ back: copy what is judged into profiles/<tag>_*.
# usage (in the container): tools/collect_evidence.sh r03
set -e
tag=${1:-r03}
o=gpurun_out/$tag
cp $o/bench.json profiles/${tag}_bench.json
cp $o/op_breakdown.txt profiles/${tag}_op_breakdown.txt
cp "$(ls -t $o/prof/*/*_kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats.csv
cp $o/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp "$(ls -t $o/pmc_fetch_size/*/*_counter_collection.csv | head -1)" profiles/${tag}_pmc_fetch_size.csv
cp "$(ls -t $o/pmc_write_size/*/*_counter_collection.csv | head -1)" profiles/${tag}_pmc_write_size.csv
cp $o/pmc_summary.txt profiles/${tag}_pmc_summary.txt
cp $o/pmc_traffic.json profiles/${tag}_pmc_traffic.json
for w in E2s06 E2s07; do
  cp $o/bench_$w.json profiles/${tag}_bench_$w.json
  cp $o/op_breakdown_$w.txt profiles/${tag}_op_breakdown_$w.txt
done
for f in bench_K5.json bench_feed.json layers_K2p.txt inkernel_clock.json inkernel_clock.txt timeline_K2p.txt timeline_K3.txt \
         bench_K3.json bench_K3d4.json bench_K3ln.json op_breakdown_K3.txt op_breakdown_K3d4.txt op_breakdown_K3ln.txt kernel_stats_K3d4.csv \
         layers_E2s07.txt layers_s06_d5_b8.txt; do
  [ -f $o/$f ] && cp $o/$f profiles/${tag}_$f
done
python3 - <<PY
import json, bench
t = json.load(open("profiles/${tag}_pmc_traffic.json"))["kernel_source_stamp"]
print("stamp", bench.kernel_source_stamp(), t, "OK" if t == bench.kernel_source_stamp() else "MISMATCH")
d = json.load(open("profiles/${tag}_bench.json")); r = d["roofline"]
print("K2p %.0f img/s %.2f ms frac_step %.3f  %s %.3f (%.0f TFLOP/s)  hbm %.1f GB" % (d["value"], d["ms_per_step"], r["frac_step"], r["family"], r["frac"], r["achieved"], r["hbm_gb_per_step"]))
for k, v in r["families"].items():
    print("  %-20s %.3f ms  frac %.3f  at-clock %s" % (k, v["ms_per_step"], v["frac"], v.get("frac_at_clock")))
print("  micro %.3f" % d["micro"]["fwd_dgrad_wgrad"]["frac"])
for w in ("E2s06", "E2s07"):
    e = json.load(open("profiles/${tag}_bench_%s.json" % w)); print(w, "%.0f img/s %.2f ms frac_step %.3f" % (e["value"], e["ms_per_step"], e["roofline"]["frac_step"]))
PY
