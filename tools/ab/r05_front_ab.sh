# Same-box A/B of round 5's front-end work in a DIAGNOSTIC build (make DIAG=1): [a] round-4 launches (RTMODT_FRONT=0 RTMODT_BNECK32=0), [b] the default,
# [c] the two VALU-bound kernels with ONE workgroup per CU (RTMODT_FRONT_WGS=1), each twice, interleaved.   gpurun -- 'bash tools/ab/r05_front_ab.sh'
O=gpurun_out/r05/front_ab; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --steps 300 --warmup 30"
for rep in 1 2; do
  RTMODT_FRONT=0 RTMODT_BNECK32=0 $B > $O/a_r04_launches_$rep.json 2>/dev/null || exit 1
  $B > $O/b_default_$rep.json 2>/dev/null || exit 1
  RTMODT_FRONT_WGS=1 $B > $O/c_one_wg_per_cu_$rep.json 2>/dev/null || exit 1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    r=json.load(open(f)); print(f.split("/")[-1], r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
