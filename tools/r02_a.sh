#!/bin/bash
# round 2, first GPU pass: bench-shape parity test, the driver's bench command vs a long run, host-frame upload variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -x -q -m gpu -k "benchmarked_shape" > gpurun_out/t_bshape.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/t_bshape.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b20.json 2> gpurun_out/b20.err; echo "b20 rc=$?"
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify"
timeout -k 10 200 python bench.py --steps 300 --warmup 30 $Q > gpurun_out/b300.json 2> gpurun_out/b300.err; echo "b300 rc=$?"
for mode in 0 1; do
  RTMODT_H2D=$mode timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames > gpurun_out/host_copy_h2d$mode.json 2> gpurun_out/host_copy_h2d$mode.err; echo "host copy h2d=$mode rc=$?"
done
RTMODT_H2D=1 timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames --stages 2 > gpurun_out/host_copy_h2d1_s2.json 2> gpurun_out/host_copy_h2d1_s2.err; echo "host copy h2d=1 stages 2 rc=$?"
timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames --host-mode mapped > gpurun_out/host_mapped.json 2> gpurun_out/host_mapped.err; echo "host mapped rc=$?"
tail -5 gpurun_out/t_bshape.log
for f in gpurun_out/b20.json gpurun_out/b300.json gpurun_out/host_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(j["value"], j["ms_per_step"], j.get("timing",{}), j.get("host_frames"), j.get("verified"), j["roofline"]["frac"])
except Exception as e:
    print("unparsed", e)
PY
done
