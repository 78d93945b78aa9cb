#!/usr/bin/env python3
"""Per-launch roofline of the forward pass from a `tools/profile_layers.py` listing (plain engine: every launch alone on the device):
for every conv its FLOPs and its COMPULSORY bytes layer by layer (input once + output + weights, fp16; a tensor kept in LDS by a fused
launch is not counted), the MFMA floor (2 500 TFLOP/s dense fp16) and the HBM floor (6.3 TB/s, the streaming rate tools/probes measure),
the larger of the two against the measured time.  The sum of the floors is what layer-by-layer execution with perfect kernels would take.
    python tools/layer_roofline.py profiles/r03/layers_chains1.txt [frames=32] > profiles/r03/layer_roofline.txt        (no GPU needed)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03/layers_chains1.txt"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
PEAK, HBM, SIZE = 2500e12, 6.3e12, 640


def out_hw(name):
    top = int(name.split(".")[0])
    if top == 22:
        return SIZE // (8, 16, 32)[int(name.split(".")[2])]
    return SIZE // {0: 2, 1: 4, 2: 4, 3: 8, 4: 8, 5: 16, 6: 16, 7: 32, 8: 32, 9: 32, 12: 16, 15: 8, 16: 16, 18: 16, 19: 32, 21: 32}[top]


rows = []
for line in open(path):
    m = re.match(r"(.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
    if m and not line.startswith("total") and not line.startswith("launch"):
        rows.append([m.group(1).strip(), float(m.group(2))])
keys = [r[0].split(" ")[0] for r in rows]


def row_of(name):
    if name in keys:
        return keys.index(name)
    if name.startswith("22."):
        return keys.index("22.stage" + name.split(".")[3])
    stem = name.rsplit(".", 1)[0]                       # 4.m.0.cv1 -> 4.m.0
    if stem in keys:
        return keys.index(stem)
    raise KeyError(name)


acc = {}
for c in pkg.weights.spec("s"):
    ho = out_hw(c.name)
    M = B * ho * ho
    K = c.cin * c.k * c.k
    flops = 2.0 * M * c.cout * K
    by_in, by_out, by_w = M * c.stride * c.stride * c.cin * 2.0, M * c.cout * 2.0, c.cout * K * 2.0
    if c.name == "0":
        by_in = B * SIZE * SIZE * 3.0                   # the stem reads the frame bytes
    a = acc.setdefault(row_of(c.name), dict(convs=[], flops=0.0, bytes=0.0))
    a["convs"].append((c.name, by_in, by_out))
    a["flops"] += flops
    a["bytes"] += by_in + by_out + by_w
# tensors that never reach HBM: the intermediate of a fused Bottleneck launch, the input of a 1x1 conv run as the tail of its producer
for i, (name, _) in enumerate(rows):
    if i in acc and "fused bottleneck" in name:
        first = acc[i]["convs"][0]
        acc[i]["bytes"] -= 2 * first[2]                 # .cv1's output: written and read back in LDS
    if "runs as the tail of the previous launch" in name and i in acc and i - 1 in acc:
        acc[i]["bytes"] -= acc[i]["convs"][0][1]        # the tail's input ...
        acc[i - 1]["bytes"] -= acc[i - 1]["convs"][-1][2] if "fused bottleneck" not in rows[i - 1][0] else 0.0   # ... is its producer's output
# one line per LAUNCH: a conv that runs as the tail of its producer, or inside head_final, joins that launch
groups = []
pending = None
for i, (name, us) in enumerate(rows):
    a = acc.get(i, dict(convs=[], flops=0.0, bytes=0.0))
    if "runs as the tail of the previous launch" in name and groups:
        g = groups[-1]
        g["flops"] += a["flops"]; g["bytes"] += a["bytes"]; g["us"] += us; g["name"] += " + " + name.split(" ")[0]
    elif "runs inside head_final" in name:
        pending = dict(name=name.split(" [")[0], flops=a["flops"], bytes=a["bytes"], us=us)
    else:
        g = dict(name=name, flops=a["flops"], bytes=a["bytes"], us=us)
        if pending:
            g["flops"] += pending["flops"]; g["bytes"] += pending["bytes"]; g["us"] += pending["us"]; pending = None
        groups.append(g)
print(f"{'launch':60s} {'GFLOP':>7s} {'MB':>7s} {'FLOP/B':>7s} {'mfma us':>8s} {'hbm us':>7s} {'floor':>7s} {'meas.':>7s} {'floor/meas':>10s}")
tf = tm = tmf = thf = 0.0
for g in groups:
    tm += g["us"]
    if g["flops"] == 0:
        print(f"{g['name'][:60]:60s} {'':7s} {'':7s} {'':7s} {'':8s} {'':7s} {'':7s} {g['us']:7.1f}")
        continue
    t_m, t_h = g["flops"] / PEAK * 1e6, g["bytes"] / HBM * 1e6
    fl = max(t_m, t_h)
    tf += fl; tmf += t_m; thf += t_h
    print(f"{g['name'][:60]:60s} {g['flops'] / 1e9:7.2f} {g['bytes'] / 1e6:7.1f} {g['flops'] / g['bytes']:7.0f} {t_m:8.1f} {t_h:7.1f} {fl:7.1f} {g['us']:7.1f} {fl / g['us'] if g['us'] else 0:10.2f}")
F = sum(a["flops"] for a in acc.values())
print(f"\n{B} frames: {F / 1e9:.1f} GFLOP, compulsory bytes layer by layer {sum(a['bytes'] for a in acc.values()) / 1e6:.0f} MB")
print(f"sum of MFMA floors {tmf:.0f} us, of HBM floors {thf:.0f} us, of max(floor) per launch {tf:.0f} us; measured launches {tm:.0f} us")
print(f"=> layer-by-layer execution with perfect kernels: {F / (tf * 1e-6) / 1e12:.0f} TFLOP/s = {F / (tf * 1e-6) / PEAK * 100:.1f} % of the MFMA peak; "
      f"measured, every launch alone on the device (plain engine) {F / (tm * 1e-6) / 1e12:.0f} TFLOP/s = {F / (tm * 1e-6) / PEAK * 100:.1f} %")
