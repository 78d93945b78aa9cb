cd $GRAFT_REPO_ROOT
run() { # label, env..., -- args
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --no-compare $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
}
EXTRA=""
run c1 RTMODT_CHAINS=1 && run c2join RTMODT_CHAINS=2 RTMODT_CHAIN_JOIN=1 && run c2free RTMODT_CHAINS=2 && run c1b RTMODT_CHAINS=1 && run c2freeb RTMODT_CHAINS=2 && run c2joinb RTMODT_CHAINS=2 RTMODT_CHAIN_JOIN=1 || exit 1
EXTRA="--frames-per-stream 1"
run f1c1 RTMODT_CHAINS=1 && run f1c2 RTMODT_CHAINS=2 || exit 1
EXTRA="--frames-per-stream 4"
run f4c1 RTMODT_CHAINS=1 && run f4c2 RTMODT_CHAINS=2 && run f4c4 RTMODT_CHAINS=4 || exit 1
