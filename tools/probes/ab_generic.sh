cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-compare --no-latency $EXTRA > gpurun_out/ab_$label.json 2>gpurun_out/ab_$label.err || return 1
  python -c "import json; d=json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['chains'], d['roofline'].get('stages'))"
}
EXTRA=""
run base A=1 && run bneck0 RTMODT_BNECK=0 && run nobt RTMODT_NO_BNECK_TAIL=1 && run tail0 RTMODT_TAIL=0 && run nohf RTMODT_NO_HEAD_FINAL=1 && run stem0 RTMODT_STEM_FUSE=0 && run base2 A=1 || exit 1
