# Stage cuts of the three-stage engine at ONE frame per stream per step (8 frames per launch set): are F = 4's cuts {6.cv2, 18.} right there too?
# DIAGNOSTIC library tools/ab/lib_diag.so (make DIAG=1 of the same sources) copied over the package's on the box; RTMODT_SPLIT3=<first op of stage 2>,<first op of stage 3>
O=gpurun_out/r05/split3_F1; mkdir -p $O
L=real-time-multi-object-detection---tracking-system_amd/lib/librtmodt_hip.so
cp $L /tmp/lib_keep.so && cp tools/ab/lib_diag.so $L || exit 1
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --steps 400 --warmup 40 --frames-per-stream 1"
for rep in 1 2; do
  for cut in "6.cv2,18." "6.m.1,16" "6.cv2,16" "7,18." "6.cv2,19" "8.,18." "6.m.0,15." "5,16" "7,19"; do
    RTMODT_SPLIT3="$cut" timeout -k 10 200 $B > $O/cut_$(echo $cut | tr ',.' '__')_$rep.json 2>/dev/null || echo "failed $cut"
  done
done
cp /tmp/lib_keep.so $L
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], r["value"], r["ms_per_step"])
    except Exception as e: print(f, "bad", e)
PY
