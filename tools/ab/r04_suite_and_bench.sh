mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/gpu_tests_5.txt 2>&1; rc=$?
tail -3 gpurun_out/r04/gpu_tests_5.txt
[ $rc -eq 0 ] || exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-compare --no-host-leg > gpurun_out/r04/bench_silu4_$i.json 2>/dev/null || exit 1; done
python - <<PY
import json
for i in (1,2):
    d=json.loads(open("gpurun_out/r04/bench_silu4_%d.json"%i).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_kernels_ms_eager"], d["roofline"].get("in_kernel_clock",{}).get("ghz_mean"))
PY
