"""Timeline of ONE graph-replayed train step from a rocprofv3 --kernel-trace CSV: every dispatch in start order with its
duration and the gap to the previous dispatch's end.  Picks the last complete step (from one adam_kernel to the next).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-micro --no-families
    python tools/graph_timeline.py DIR [--anchor adam_kernel] [--which -2]"""
import argparse
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        rest = name[m.end():]
        base, tail = rest[:n], rest[n:]
        digits = re.findall(r"Li(\d+)E|Lb([01])E", tail)
        args = ",".join(a or b for a, b in digits)
        return f"{base}<{args}>" if args else base
    return name.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--anchor", default="adam_kernel")
    ap.add_argument("--which", type=int, default=-2, help="index of the anchor occurrence that starts the step (default: second to last)")
    a = ap.parse_args()
    files = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if a.anchor in r[2]]
    if len(anchors) < 2:
        sys.exit("fewer than two anchors")
    i0, i1 = anchors[a.which] + 1, anchors[a.which + 1] + 1 if a.which + 1 != 0 else len(rows)
    step = rows[i0:i1]
    t0 = step[0][0]
    prev_end = None
    tot_k = tot_gap = 0.0
    print(f"{'#':>4} {'start us':>10} {'dur us':>9} {'gap us':>8}  kernel")
    for k, (s, e, n) in enumerate(step):
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{k:>4} {(s - t0) / 1e3:>10.1f} {(e - s) / 1e3:>9.1f} {gap:>8.1f}  {short(n)}")
        tot_k += (e - s) / 1e3
        tot_gap += max(gap, 0.0)
        prev_end = max(prev_end or e, e)
    print(f"dispatches {len(step)}, span {(step[-1][1] - t0) / 1e3:.1f} us, kernel time {tot_k:.1f} us, positive gaps {tot_gap:.1f} us")


if __name__ == "__main__":
    main()
