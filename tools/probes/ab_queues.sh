cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --no-compare $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'])"
}
EXTRA=""
run q4c2 RTMODT_CHAINS=2 && run q8c2 GPU_MAX_HW_QUEUES=8 RTMODT_CHAINS=2 && run q8c4 GPU_MAX_HW_QUEUES=8 RTMODT_CHAINS=4 RTMODT_TUNE_LOG=1 && run q4c4 RTMODT_CHAINS=4 || exit 1
grep streams gpurun_out/ab_q8c4.err | tail -12
EXTRA="--frames-per-stream 4"
run f4q4c2 RTMODT_CHAINS=2 && run f4q8c4 GPU_MAX_HW_QUEUES=8 RTMODT_CHAINS=4 || exit 1
EXTRA="--frames-per-stream 3"
run f3q4c2 RTMODT_CHAINS=2 && run f3q8c3 GPU_MAX_HW_QUEUES=8 RTMODT_CHAINS=3 && run f3q4c3 RTMODT_CHAINS=3 || exit 1
