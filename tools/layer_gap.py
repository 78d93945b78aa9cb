#!/usr/bin/env python3
"""Per launch: measured time (tools/profile_layers.py output, kernels alone on the device) against the time the same FLOPs
take at 40 % of the dense fp16 MFMA peak (north_star's target, 1000 TFLOP/s), and the marginal rate between two batch sizes.
    python tools/layer_gap.py profiles/r02/layers_chains1.txt [profiles/r02/layers_chains1_F8.txt]"""
import re
import sys


def load(path):
    rows = []
    for line in open(path):
        m = re.match(r"(.{72}) +([\d.]+) +([\d.]+) +([\d.]+)$", line.rstrip("\n"))
        if m:
            rows.append((m.group(1).strip(), float(m.group(2)), float(m.group(3))))
    return rows


a = load(sys.argv[1])
b = load(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"{'launch':58s} {'us':>7s} {'TFLOP/s':>8s} {'us @40%':>8s} {'x over':>7s} {'share of the gap':>17s}" + ("   marginal TFLOP/s (second file)" if b else ""))
tot = sum(u for _, u, _ in a)
tgt = [u * t / 1000.0 for _, u, t in a]          # us * TFLOP/s = MFLOP; / 1000 TFLOP/s = us at 40 %
gap = sum(max(u - g, 0) for (_, u, _), g in zip(a, tgt))
for i, ((n, u, t), g) in enumerate(zip(a, tgt)):
    extra = ""
    if b and i < len(b):
        u2, t2 = b[i][1], b[i][2]
        marg = (u2 * t2 - u * t) / (u2 - u) if u2 > u else 0.0
        extra = f"   {marg:8.1f}"
    over = f"{u / g:7.2f}" if g > 0 else "      -"
    print(f"{n[:58]:58s} {u:7.1f} {t:8.1f} {g:8.1f} {over} {100 * max(u - g, 0) / gap:16.1f}%" + extra)
print(f"total {tot:.1f} us; at 40 % of peak the same FLOPs take {sum(tgt):.1f} us; launches without FLOPs (pool, head_final's decode) count in full")
