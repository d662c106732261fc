#!/bin/bash
# SQ counters of the scan kernel chosen by $1 (scan2 / scan5 / ...) under the timing-study knock-out GFT_SCAN_DEBUG=$2 (0: the
# production kernel, 1: filter only, 4: no bucket table, 8: no short terms, 12: both), per document -> gpurun_out/sqd_$1_$2.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
K=$1
O=gpurun_out/tac_$K
rm -rf $O && mkdir -p $O
export GFT_SCAN_KERNEL=$K
export GFT_SCAN_DEBUG=$2
B="python3 tools/probe_scan.py --docs 500000 --unordered --modes $2 --reps 2"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $O/p1 -o run --output-format csv -- $B > $O/p1.log 2>&1
python3 tools/sq_summary.py $O/p1 --docs 500000 | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({k:v for k,v in d.items() if k.startswith('k_scan') and '<' in k},indent=1))" > gpurun_out/sqd_${K}_$2.json
rm -rf $O
