cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA=""
export RTMODT_TUNE_CACHE=/tmp/tc.txt
run base RTMODT_POST_PRIO=0 && run prio RTMODT_POST_PRIO=1 RTMODT_TUNE_LOG=1 && run base2 RTMODT_POST_PRIO=0 && run prio2 RTMODT_POST_PRIO=1 || exit 1
grep streams gpurun_out/ab_prio.err | tail -8
EXTRA="--host-frames --stages 3"
run hbase RTMODT_POST_PRIO=0 && run hprio RTMODT_POST_PRIO=1 || exit 1
EXTRA="--host-frames --stages 2"
run h2base RTMODT_POST_PRIO=0 && run h2prio RTMODT_POST_PRIO=1 || exit 1
