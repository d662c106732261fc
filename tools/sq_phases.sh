#!/bin/bash
# SQ counters of the scan kernel per phase (GFT_SCAN_DEBUG knock-outs), per document: run through gpurun from the repo root.
#   tools/sq_phases.sh "<modes>" [extra probe_scan args]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
modes=${1:-"0 1 2 3"}
shift
for mode in $modes; do
  rm -rf gpurun_out/sqp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/sqp -o run --output-format csv -- python3 tools/probe_scan.py --docs 1000000 --unordered --reps 2 --modes $mode "$@" > gpurun_out/sqp_$mode.log 2>&1
  python3 tools/sq_summary.py gpurun_out/sqp --docs 1000000 > gpurun_out/sq_phase_$mode.json
  rm -rf gpurun_out/sqp
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d gpurun_out/sqp -o run --output-format csv -- python3 tools/probe_scan.py --docs 1000000 --unordered --reps 2 --modes $mode "$@" > gpurun_out/sqp2_$mode.log 2>&1
  python3 tools/sq_summary.py gpurun_out/sqp --docs 1000000 > gpurun_out/sq_phase2_$mode.json
  rm -rf gpurun_out/sqp
done
echo done
