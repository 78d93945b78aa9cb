#!/bin/bash
# r03: free-running fp16-vs-fp32 decision audit over 16 frames per engine configuration (tools/diag_e2e.py)
set -e
O=gpurun_out/diag; mkdir -p $O
RTMODT_TUNE_CACHE=$O/tune.txt python3 tools/diag_e2e.py --tag autotune --autotune 1 --out $O > $O/autotune.txt 2>&1
python3 tools/diag_e2e.py --tag default --autotune 0 --out $O > $O/default.txt 2>&1
RTMODT_BNECK=1 python3 tools/diag_e2e.py --tag bneck1 --autotune 0 --out $O > $O/bneck1.txt 2>&1
RTMODT_BNECK=0 RTMODT_TAIL=0 RTMODT_TILE_K64=44 RTMODT_TILE_3X3S1=-1 python3 tools/diag_e2e.py --tag pt44 --autotune 0 --out $O > $O/pt44.txt 2>&1
RTMODT_BNECK=0 python3 tools/diag_e2e.py --tag autotune_bneck0 --autotune 1 --out $O > $O/autotune_bneck0.txt 2>&1
python3 tools/diag_e2e.py --tag hd_autotune --autotune 1 --hw 1080x1920 --seeds 5:9 --out $O > $O/hd_autotune.txt 2>&1
tail -n 1 $O/*.txt
