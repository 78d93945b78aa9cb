set -o pipefail
mkdir -p gpurun_out/r05/pmc_front
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=${1:-tools/probes/bin/front_probe}
T=${2:-front}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/r05/pmc_$T/a -- $P 32 2 > /dev/null 2> gpurun_out/r05/pmc_$T/a.log || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d gpurun_out/r05/pmc_$T/b -- $P 32 2 > /dev/null 2> gpurun_out/r05/pmc_$T/b.log || exit 1
python3 - <<PY
import csv,glob,collections
for tag in "ab":
    for f in glob.glob("gpurun_out/r05/pmc_$T/%s/*/*counter_collection.csv"%tag):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items():
            if "front_fused" in k or "c2f32" in k or "bottleneck_fused" in k or "stem_fused" in k or "conv_mfma_tail" in k:
                print(k, {a:round(b/1e6,2) for a,b in v.items()})
PY
