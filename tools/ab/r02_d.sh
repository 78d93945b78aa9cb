#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 2 > gpurun_out/layers_F2.txt 2> /dev/null
RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 8 > gpurun_out/layers_F8.txt 2> /dev/null
python - <<'PY'
import re
def load(p):
    rows=[]
    for l in open(p):
        m=re.match(r"(.{72}) +([\d.]+) +([\d.]+) +([\d.]+)$", l.rstrip("\n"))
        if m: rows.append((m.group(1).strip(), float(m.group(2)), float(m.group(3))))
    return rows
a=load("gpurun_out/layers_F2.txt"); b=load("gpurun_out/layers_F8.txt")
print(f"{'layer':40s} {'us@16':>8s} {'us@64':>8s} {'TF@16':>7s} {'TF@64':>7s} {'marg TF':>8s} {'fixed us':>8s}")
ta=tb=0
for (n1,u1,t1),(n2,u2,t2) in zip(a,b):
    fl16=t1*u1  # TF*us = MFLOP... relative units
    fl64=t2*u2
    marg=(fl64-fl16)/(u2-u1) if u2>u1 else 0
    fixed=u1-(u2-u1)/3.0
    ta+=u1; tb+=u2
    print(f"{n1.split(' [')[0][:40]:40s} {u1:8.1f} {u2:8.1f} {t1:7.1f} {t2:7.1f} {marg:8.1f} {fixed:8.1f}   {n2.split('[')[-1][:40]}")
print("total", ta, tb)
PY
