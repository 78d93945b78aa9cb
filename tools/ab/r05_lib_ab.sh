#!/bin/bash
# same-box A/B of two builds of the library: tools/ab/lib_old.so, tools/ab/lib_new.so copied over the package's library in turn (the box's copy of the repo is scratch)
# usage (on the GPU box): bash tools/ab/r05_lib_ab.sh OUTDIR [rounds]
O=${1:-gpurun_out/r05/lib_ab}; N=${2:-3}
mkdir -p $O
L=real-time-multi-object-detection---tracking-system_amd/lib/librtmodt_hip.so
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --prewarm 0.2 --steps 300 --warmup 30"
for r in $(seq 1 $N); do
  for v in old new; do
    cp tools/ab/lib_$v.so $L || exit 1
    timeout -k 10 200 $B > $O/${v}_$r.json 2> $O/${v}_$r.err || exit 1
    python3 -c "import json; d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1]); print('$v', $r, d['value'], d['ms_per_step'], d['roofline']['library_build']['csrc_sha256'][:8])"
  done
done
cp tools/ab/lib_new.so $L
