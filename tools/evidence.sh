#!/bin/bash
# Evidence for profiles/ in one gpurun call (on the GPU box): tools/evidence.sh <tag>   e.g. tools/evidence.sh r03
#   1. two single-counter PMC passes (FETCH_SIZE, WRITE_SIZE) over 3 eager train steps of the headline workload
#   2. the plain bench line with the per-op-family table (un-profiled; this is the number that counts)
#   3. the same bench under rocprofv3 --kernel-trace --stats
#   4. breakdown tables of the reference's own Experiment-2 shapes
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/ afterwards.
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$out/pmc_$(echo $ctr | tr 'A-Z' 'a-z')
    rm -rf "$d"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$d" -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-micro > "$out/pmc_$ctr.log" 2>&1 || exit 1
done
python3 tools/pmc_summary.py "$out/pmc_fetch_size" "$out/pmc_write_size" "$out/pmc_traffic.json" 3 > "$out/pmc_summary.txt" || exit 1
cp "$out/pmc_traffic.json" profiles/${tag}_pmc_traffic.json          # bench.py quotes it (stamped) from here on
python3 bench.py --breakdown > "$out/bench.json" 2> "$out/op_breakdown.txt" || exit 1
rm -rf "$out/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --no-cpu-baseline --no-micro > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err" || exit 1
cp "$(ls $out/prof/*/*_kernel_stats.csv | head -1)" "$out/kernel_stats.csv"
for w in E2s06 E2s07; do
    python3 bench.py --workload $w --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-micro > "$out/bench_$w.json" 2> "$out/op_breakdown_$w.txt" || exit 1
done
#   5. r04: BASELINE config 5 (build-defined mixed stream), the fed rate beside the resident one, the per-launch table of K2p,
#      and the in-kernel clock table when the diagnostic build ab/clock.so (tools/build_variant.sh clock -DAD_CLOCK) is present
python3 bench.py --workload K5 --steps 8 --warmup 2 > "$out/bench_K5.json" 2> "$out/bench_K5.err" || exit 1
python3 bench.py --feed loader --no-cpu-baseline --no-micro --steps 40 > "$out/bench_feed.json" 2> "$out/bench_feed.err" || exit 1
python3 tools/layer_table.py --workload K2p > "$out/layers_K2p.txt" 2>&1 || exit 1
if [ -f ab/clock.so ]; then
    ADUNET_LIB=ab/clock.so python3 tools/inkernel_clock.py --json "$out/inkernel_clock.json" > "$out/inkernel_clock.txt" 2>&1 || exit 1
fi
#   6. r05: BASELINE config 3 as the reference's source defines it (segmentation U-Nets, bench.py SEG_WORKLOADS): bench lines with
#      the op table, the BatchNorm model under rocprofv3 --kernel-trace --stats; and the per-launch tables that size the round
#      quantisation of the small-batch Experiment-2 rows
for w in K3 K3d4 K3ln; do
    python3 bench.py --workload $w --no-cpu-baseline --breakdown > "$out/bench_$w.json" 2> "$out/op_breakdown_$w.txt" || exit 1
done
rm -rf "$out/prof_K3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_K3" -- python3 bench.py --workload K3d4 --no-cpu-baseline > "$out/bench_K3d4_under_rocprof.json" 2> "$out/bench_K3d4_under_rocprof.err" || exit 1
cp "$(ls $out/prof_K3/*/*_kernel_stats.csv | head -1)" "$out/kernel_stats_K3d4.csv"
python3 tools/layer_table.py --workload E2s07 > "$out/layers_E2s07.txt" 2>&1 || exit 1
python3 tools/layer_table.py --workload 0.6,5,256,8 > "$out/layers_s06_d5_b8.txt" 2>&1 || exit 1
#   7. r05 (second half): the per-dispatch timeline of ONE graph-replayed step (tools/graph_timeline.py): what every launch costs
#      inside the graph -- eager HIP-event timings overstate launches under ~100 us
for w in K2p K3; do
    rm -rf "$out/trace_$w"
    rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$w" -- python3 bench.py --workload $w --no-cpu-baseline --no-micro --steps 6 --warmup 2 > /dev/null 2> "$out/trace_$w.err" || exit 1
    python3 tools/graph_timeline.py "$out/trace_$w" --which -6 > "$out/timeline_$w.txt" || exit 1
    rm -rf "$out/trace_$w"
done
tail -3 "$out/pmc_summary.txt"; cat "$out/bench.json" | head -c 600; echo
