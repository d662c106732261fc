#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, FETCH_SIZE / WRITE_SIZE (separate passes), SQ instruction counters, and the default bench line.
# Outputs land in gpurun_out/; copy the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
rocprofv3 --kernel-trace --stats -d gpurun_out/r1_stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-docs 0 > gpurun_out/r1_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-docs 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o runc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-docs 0 > gpurun_out/pmc_write.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r1_pmc_traffic.json --docs 1000000 > gpurun_out/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/sq_final -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-docs 0 > gpurun_out/sq_final.log 2>&1
python3 tools/sq_summary.py gpurun_out/sq_final > gpurun_out/r1_sq_counters_raw.json
for mode in 1 12 4; do
  GFT_SCAN_DEBUG=$mode rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/sq_m$mode -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-docs 0 > gpurun_out/sq_m$mode.log 2>&1
  python3 tools/sq_summary.py gpurun_out/sq_m$mode --docs 1000000 > gpurun_out/sq_mode_$mode.json
  rm -rf gpurun_out/sq_m$mode
done
find gpurun_out/r1_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1_final_kernel_stats.csv \;
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/sq_final gpurun_out/r1_stats
echo profiles collected
