#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "weight_stationary" > gpurun_out/t_ws.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_ws.log
RTMODT_TUNE_WS=1 RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/layers_o.txt 2> gpurun_out/layers_o.err
python - <<'PY'
import re, collections
best=collections.OrderedDict()
for l in open('gpurun_out/layers_o.err'):
    m=re.match(r"\[tune\] (\S+)\s+(\S+)\s+([\d.]+) us", l)
    if not m: continue
    n,t,us=m.group(1),m.group(2),float(m.group(3))
    d=best.setdefault(n,{})
    d[t]=us
for n,d in best.items():
    ws={k:v for k,v in d.items() if k.startswith('ws:')}
    if not ws: continue
    other=min(v for k,v in d.items() if not k.startswith('ws:'))
    ob=[k for k,v in d.items() if v==other][0]
    print(f"{n:12s} best other {ob:18s} {other:7.2f} us | " + "  ".join(f"{k} {v:7.2f}" for k,v in ws.items()))
PY
grep -E "ws:|^total" gpurun_out/layers_o.txt
