cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA=""
run fold RTMODT_UP_READ=0 && run lo A=1 && run fold2 RTMODT_UP_READ=0 && run lo2 A=1 || exit 1
RTMODT_UP_READ=0 RTMODT_CHAINS=1 python tools/profile_layers.py 2>/dev/null | grep -E "^9.cv2|^12.cv1|^12.cv2|^15.cv1|^total"
RTMODT_CHAINS=1 python tools/profile_layers.py 2>/dev/null | grep -E "^9.cv2|^12.cv1|^12.cv2|^15.cv1|^total"
