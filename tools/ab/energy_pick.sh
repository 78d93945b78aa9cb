#!/bin/bash
# Does picking a layer's LOWEST-ENERGY tile (kernel_probe energy) instead of its fastest one move the staged bench?
# Variant B rewrites the tune cache of variant A for the 3x3 convs 128 -> 128 at 40x40 (6.m / 12.m / 18.m) to k64:256x128s2/8w.
mkdir -p gpurun_out/ep
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
export RTMODT_TUNE_CACHE=/tmp/ep_a.txt; rm -f /tmp/ep_a.txt /tmp/ep_b.txt
python bench.py --steps 100 --warmup 10 $Q > gpurun_out/ep/gen.json 2>/dev/null || exit 1
cp /tmp/ep_a.txt gpurun_out/ep/cache_a.txt
python - <<'PY'
import re
lines = open('/tmp/ep_a.txt').read().splitlines()
out = []
n = 0
for l in lines:
    if l.startswith('#'):
        out.append(l); continue
    key, val = l.split('\t')
    t0, t1, fused = val.split()
    if re.match(r'^(6|12|18)\.m\.', key) and '|128>128|k3s1' in key:
        # Bottleneck ops: t0 / t1 are the tiles of the two convs when not fused
        t0 = t1 = '35'; fused = '0'; n += 1
    out.append(f"{key}\t{t0} {t1} {fused}")
open('/tmp/ep_b.txt', 'w').write('\n'.join(out) + '\n')
print('rewritten', n)
PY
cp /tmp/ep_b.txt gpurun_out/ep/cache_b.txt
for rep in 1 2 3; do
  for v in a b; do
    RTMODT_TUNE_CACHE=/tmp/ep_$v.txt timeout -k 10 200 python bench.py --steps 300 --warmup 20 $Q > gpurun_out/ep/${v}_$rep.json 2>/dev/null || exit 1
  done
done
for f in gpurun_out/ep/[ab]_?.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
grep -E "^(6|12|18)\.m\." /tmp/ep_a.txt | head -8
