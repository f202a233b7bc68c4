"""Dev tool (GPU box): read a rocprofv3 kernel_trace.csv of bench.py and split every kernel's durations by what ran right before it
(K1 after a GEMM = inside the qlinear step; K1 after K1 = the K1-alone replays; ...), plus the idle gap between consecutive kernels.
usage: python tools/trace_instep.py <kernel_trace.csv>"""
import collections
import csv
import statistics
import sys


def short(name):
    for k, v in (("gemm_s8", "GEMM"), ("quant_rowwise", "K1")):
        if k in name:
            return v
    return "other"


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", "")))
rows.sort()
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for (s0, e0, k0, g0), (s1, e1, k1, g1) in zip(rows, rows[1:]):
    if s1 - e0 > 50_000:          # a host-side pause between graph replays, not a kernel boundary
        continue
    key = f"{k1}[grid {g1}] after {k0}"
    dur[key].append((e1 - s1) / 1e3)
    gap[key].append((s1 - e0) / 1e3)
for k in sorted(dur):
    d, g = sorted(dur[k]), sorted(gap[k])
    if len(d) < 20:
        continue
    print(f"{k:44s} n={len(d):6d}  duration median {statistics.median(d):7.2f} us  mean {sum(d) / len(d):7.2f}  p10 {d[len(d) // 10]:7.2f}  p90 {d[9 * len(d) // 10]:7.2f}"
          f"   gap to predecessor median {statistics.median(g):5.2f} us  mean {sum(g) / len(g):5.2f}")
