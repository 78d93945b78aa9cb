cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA="--host-frames"
run hs2 A=1 && run hs3 RTMODT_STAGES=3 && run hs2b A=1 && run hs3b RTMODT_STAGES=3 || exit 1
EXTRA="--host-frames --pageable"
run ps2 A=1 && run ps3 RTMODT_STAGES=3 || exit 1
for S in 2 4; do
EXTRA="--streams $S --frames-per-stream 1"
run n${S}s2 A=1 && run n${S}s3 RTMODT_STAGES=3 || exit 1
done
