#!/bin/bash
# round 2, GPU pass A: parity tests for the new paths, the driver's bench command vs a long run, host-frame variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -x -q -m gpu -k "benchmarked_shape or page_locked or staged_pipeline or free_running" > gpurun_out/t_new.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/t_new.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b20.json 2> gpurun_out/b20.err; echo "b20 rc=$?"
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify"
timeout -k 10 200 python bench.py --steps 300 --warmup 30 $Q > gpurun_out/b300.json 2> gpurun_out/b300.err; echo "b300 rc=$?"
timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames > gpurun_out/host_inplace.json 2> gpurun_out/host_inplace.err; echo "host in place rc=$?"
RTMODT_ZERO_COPY=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames > gpurun_out/host_copy.json 2> gpurun_out/host_copy.err; echo "host copy rc=$?"
RTMODT_ZERO_COPY=0 RTMODT_H2D=1 timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames > gpurun_out/host_copy_h2d1.json 2> gpurun_out/host_copy_h2d1.err; echo "host copy h2d1 rc=$?"
RTMODT_ZERO_COPY=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames --stages 2 > gpurun_out/host_copy_s2.json 2> gpurun_out/host_copy_s2.err; echo "host copy stages 2 rc=$?"
timeout -k 10 200 python bench.py --steps 200 --warmup 30 $Q --host-frames --pageable > gpurun_out/host_pageable.json 2> gpurun_out/host_pageable.err; echo "host pageable rc=$?"
tail -5 gpurun_out/t_new.log
for f in gpurun_out/b20.json gpurun_out/b300.json gpurun_out/host_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t=j.get("timing",{})
    print(j["value"], j["ms_per_step"], "cold", t.get("cold_ms_per_step"), "long", t.get("long_window",{}).get("value"), "p50/max", t.get("step_ms_p50"), t.get("step_ms_max"), "host", j.get("host_frames"), "ver", j.get("verified"), "frac", j["roofline"]["frac"], "1f", j.get("one_frame_per_stream_per_step",{}).get("value"), "lat", j.get("latency_single_stream_ms",{}).get("p50"))
except Exception as e:
    print("unparsed", e)
PY
done
