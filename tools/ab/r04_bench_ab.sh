# Round 4, run BEFORE the tile table was pruned (ids 57..59 + 62 were the ping-pong tap-reuse tiles, 60 / 61 the ping-pong
# plain tiles of that table; today: 22..25 and 26).  RTMODT_TUNE_SKIP / RTMODT_TUNE_LDS_PENALTY are rt_diag() switches now:
# build csrc with `make DIAG=1` to repeat this.  Results: profiles/r04/ab/.
mkdir -p gpurun_out/r04/ab
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-verify > gpurun_out/r04/ab/$name.json 2> gpurun_out/r04/ab/$name.err || exit 1; python - <<PY
import json
d=json.loads(open("gpurun_out/r04/ab/$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("conv_kernels_ms_eager"))
PY
}
run base A=1
run nopp RTMODT_TUNE_SKIP=57,58,59,60,61
run noppt RTMODT_TUNE_SKIP=60,61
run pen10 RTMODT_TUNE_LDS_PENALTY=0.10
run pen25 RTMODT_TUNE_LDS_PENALTY=0.25
run base2 A=1
