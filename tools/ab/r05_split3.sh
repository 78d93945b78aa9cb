# Stage cuts of the three-stage engine after the front end got ~75 us shorter (DIAGNOSTIC build: RTMODT_SPLIT3=<first op of stage 2>,<first op of stage 3>)
O=gpurun_out/r05/split3; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --steps 300 --warmup 30"
for rep in 1 2; do
  for cut in ${CUTS:-"6.m.1,16" "6.cv2,16" "7,16" "6.cv2,18." "7,18." "6.m.1,15." "5,15."}; do
    RTMODT_SPLIT3="$cut" $B > $O/cut_$(echo $cut | tr ',.' '__')_$rep.json 2>/dev/null || echo "failed $cut"
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        r=json.load(open(f)); print(f.split("/")[-1], r["value"], r["ms_per_step"], r["roofline"]["stages"])
    except Exception as e: print(f, "bad", e)
PY
