#!/bin/bash
# SQ instruction counters of the scan kernel chosen by $1 (scan2 / scan4 / ...), per document -> gpurun_out/sq_$1_{a,b}.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
K=$1
O=gpurun_out/sqc_$K
rm -rf $O && mkdir -p $O
export GFT_SCAN_KERNEL=$K
B="python3 bench.py --steps 4 --warmup 2 --cpu-docs 0 $2"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq1 -o run --output-format csv -- $B > $O/sq1.log 2>&1
python3 tools/sq_summary.py $O/sq1 --docs 1000000 > gpurun_out/sq_${K}_a.json
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/sq2 -o run --output-format csv -- $B > $O/sq2.log 2>&1
python3 tools/sq_summary.py $O/sq2 --docs 1000000 > gpurun_out/sq_${K}_b.json
rm -rf $O
