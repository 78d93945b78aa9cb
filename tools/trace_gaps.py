#!/usr/bin/env python3
"""Where a step's time goes, from a rocprofv3 --kernel-trace CSV of `bench.py`.  The forward pass runs as one chain of
dependent launches per sub-batch (or per stage of the net), each on its own queue (the queues that carry a stem or a
head_final launch); over the last
`steps` steps this prints, per chain, the sum of its kernels' durations and the idle gaps between consecutive
kernels, and for the device the time during which at least one / at least two forward-pass kernels were running.
    python tools/trace_gaps.py gpurun_out/trace/*/*_kernel_trace.csv [steps]
"""
import collections
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
is_stem = lambda name: "letterbox" in name or "stem_fused" in name or "front_fused" in name      # first launch of a chain's step
stem_q = sorted({r[3] for r in rows if is_stem(r[2])})
queues = stem_q + sorted({r[3] for r in rows if "conv" in r[2] or "bottleneck" in r[2] or "c2f32" in r[2] or "head_final" in r[2]} - set(stem_q))   # (+ the later stages' queues of the staged engine)
first = [r for r in rows if is_stem(r[2]) and r[3] == queues[0]]
assert len(first) > steps, "trace shorter than the requested number of steps"
t0, t1 = first[-steps - 1][0], first[-1][0]
skip = ("nms", "tracker", "copyBuffer", "zones")
fwd = [r for r in rows if r[3] in queues and t0 <= r[0] < t1 and not any(s in r[2] for s in skip)]


def short(name):
    k = name.replace("(anonymous namespace)::", "").replace("void rtmodt::", "").replace("rtmodt::", "")
    return (k.split("(")[0] if "<" in k.split("(")[0] or not k.startswith("_Z") else k)[:60]


print(f"steps analysed {steps}; step period {(t1 - t0) / steps / 1e3:.1f} us; forward-pass chains (queues) {len(queues)}")
per = collections.defaultdict(lambda: [0, 0.0, 0.0])       # kernel -> calls, us, gap-before us
for q in queues:
    ks = [r for r in fwd if r[3] == q]
    busy = sum(e - s for s, e, _, _ in ks)
    gaps, prev_end = [], None
    for s, e, name, _ in ks:
        k = short(name)
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
        if prev_end is not None:
            g = max(0, s - prev_end)
            gaps.append(g)
            per[k][2] += g / 1e3
        prev_end = max(prev_end or 0, e)
    print(f"chain on queue {q}: {len(ks) / steps:5.1f} kernels/step, kernel time {busy / steps / 1e3:8.1f} us/step "
          f"({100.0 * busy / (t1 - t0):.1f} % of the period), idle between its kernels {sum(gaps) / steps / 1e3:6.1f} us/step "
          f"(mean gap {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
# device view: how long at least one / at least two forward-pass kernels were in flight
ev = sorted([(s, 1) for s, e, _, _ in fwd] + [(e, -1) for s, e, _, _ in fwd])
depth, last, ge1, ge2 = 0, t0, 0, 0
for t, dlt in ev:
    t = min(max(t, t0), t1)
    if depth >= 1:
        ge1 += t - last
    if depth >= 2:
        ge2 += t - last
    depth += dlt
    last = t
tot = sum(e - s for s, e, _, _ in fwd)
print(f"device: >= 1 forward kernel running {100.0 * ge1 / (t1 - t0):.1f} % of the period, >= 2 running {100.0 * ge2 / (t1 - t0):.1f} %; "
      f"sum of kernel durations {tot / steps / 1e3:.1f} us/step = {tot / (t1 - t0):.2f} x the period")
print(f"{'kernel':62s} {'calls/step':>10s} {'us/step':>9s} {'avg us':>8s} {'gap before, us/step':>20s}")
for k, (c, us, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} {c / steps:10.1f} {us / steps:9.1f} {us / c:8.1f} {g / steps:20.1f}")
