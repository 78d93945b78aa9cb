#!/usr/bin/env python3
"""Where a step's time goes, from a rocprofv3 --kernel-trace CSV of `bench.py`: per step (letterbox to
letterbox) the wall span of the forward pass on the compute stream, the sum of its kernels' durations, and
the idle gaps between consecutive kernels -- the price of ~47 dependent launches per step.
    python tools/trace_gaps.py gpurun_out/trace/*/*_kernel_trace.csv [steps]
"""
import collections
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
lb = [i for i, r in enumerate(rows) if "letterbox" in r[2] or "stem_fused" in r[2]]      # first launch of a step
lb = lb[-steps - 1:]
span = busy = 0
gaps = []
per = collections.defaultdict(lambda: [0, 0.0, 0.0])       # kernel -> calls, us, gap-before us
n = 0
for a, b in zip(lb[:-1], lb[1:]):
    # forward-pass kernels of this step: everything on the letterbox's queue between the two letterboxes
    q = rows[a][3]
    ks = [r for r in rows[a:b] if r[3] == q and "nms" not in r[2] and "tracker" not in r[2] and "copyBuffer" not in r[2]]
    if len(ks) < 10:
        continue
    n += 1
    span += ks[-1][1] - ks[0][0]
    prev_end = None
    for s, e, name, _, _ in ks:
        busy += e - s
        k = name.replace("(anonymous namespace)::", "").replace("void rtmodt::", "").replace("rtmodt::", "")
        k = (k.split("(")[0] if "<" in k.split("(")[0] or not k.startswith("_Z") else k)[:60]
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
        if prev_end is not None:
            g = max(0, s - prev_end)
            gaps.append(g)
            per[k][2] += g / 1e3
        prev_end = max(prev_end or 0, e)
print(f"steps analysed {n}; kernels per step {len(gaps) / max(n, 1) + 1:.1f}")
print(f"forward span per step {span / n / 1e3:8.1f} us")
print(f"kernel busy per step  {busy / n / 1e3:8.1f} us  ({100.0 * busy / span:.1f} % of the span)")
print(f"idle between kernels  {sum(gaps) / n / 1e3:8.1f} us  (mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
print(f"{'kernel':62s} {'calls/step':>10s} {'us/step':>9s} {'avg us':>8s} {'gap before, us/step':>20s}")
for k, (c, us, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} {c / n:10.1f} {us / n:9.1f} {us / c:8.1f} {g / n:20.1f}")
