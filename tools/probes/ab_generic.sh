cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA=""
run base A=1 && run s8cv2 RTMODT_SPLIT=8.cv2 && run s9pool RTMODT_SPLIT=9.pool && run s9cv2 RTMODT_SPLIT=9.cv2 && run s12m RTMODT_SPLIT=12.m && run s12cv2 RTMODT_SPLIT=12.cv2 && run s15cv1 RTMODT_SPLIT=15.cv1 && run base2 A=1 || exit 1
