#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, FETCH_SIZE / WRITE_SIZE / TCC hit-miss (separate passes), SQ instruction counters, the default
#   bench line, and the same for the mixed alphabet and the 100 k-term configuration.
# Outputs land in gpurun_out/r2/; copy the summaries into profiles/ afterwards (tools/collect_profiles.sh does not touch it).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
O=gpurun_out/r2
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 5 --warmup 2 --cpu-docs 0"
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/r2_final_bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r2_final_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o runc --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o runc --output-format csv -- $B > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r2_pmc_traffic.json --docs 1000000 > $O/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_tcc -o runc --output-format csv -- $B > $O/pmc_tcc.log 2>&1
python3 tools/sq_summary.py $O/pmc_tcc > $O/r2_tcc_counters.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq1 -o run --output-format csv -- $B > $O/sq1.log 2>&1
python3 tools/sq_summary.py $O/sq1 --docs 1000000 > $O/r2_sq_counters_a.json
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/sq2 -o run --output-format csv -- $B > $O/sq2.log 2>&1
python3 tools/sq_summary.py $O/sq2 --docs 1000000 > $O/r2_sq_counters_b.json
# mixed alphabet (runs on k_scan3)
python3 bench.py --steps 10 --warmup 2 --alphabet mixed > $O/bench_mixed.log 2>&1
tail -1 $O/bench_mixed.log > $O/r2_mixed_bench.json
rocprofv3 --kernel-trace --stats -d $O/stats_mixed -o run --output-format csv -- $B --alphabet mixed > $O/stats_mixed.log 2>&1
find $O/stats_mixed -name "*kernel_stats.csv" -exec cp {} $O/r2_mixed_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq_mixed -o run --output-format csv -- $B --alphabet mixed > $O/sq_mixed.log 2>&1
python3 tools/sq_summary.py $O/sq_mixed --docs 1000000 > $O/r2_mixed_sq_counters.json
# BASELINE configs[4]'s device half: 100 000 terms, 1 000 expressions with INORD, 200 000 documents
C5="--terms 100000 --exprs 1000 --inord 0.5 --docs 200000"
python3 bench.py --steps 5 --warmup 2 $C5 --cpu-docs 2000 > $O/bench_c5.log 2>&1
tail -1 $O/bench_c5.log > $O/r2_c5_bench.json
rocprofv3 --kernel-trace --stats -d $O/stats_c5 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-docs 0 $C5 > $O/stats_c5.log 2>&1
find $O/stats_c5 -name "*kernel_stats.csv" -exec cp {} $O/r2_c5_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq_c5 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-docs 0 $C5 > $O/sq_c5.log 2>&1
python3 tools/sq_summary.py $O/sq_c5 --docs 200000 > $O/r2_c5_sq_counters.json
# the same configuration with 50 % INORD expressions (the solver variant with the position algebra)
python3 bench.py --steps 10 --warmup 2 --inord 0.5 --cpu-docs 20000 > $O/bench_inord.log 2>&1
tail -1 $O/bench_inord.log > $O/r2_inord_bench.json
# phase clocks of the timing-study kernel variants: cycles per phase (solver: per group and wave; scan: per unit)
python3 tools/probe_solve.py --docs 1000000 --modes 8 2>&1 | grep "solve debug\|GFT_SOLVE" | tail -18 > $O/r2_solve_phase_clocks.txt || true
python3 tools/probe_solve.py --docs 1000000 --inord 0.5 --modes 8 2>&1 | grep "solve debug\|GFT_SOLVE" | tail -18 > $O/r2_solve_phase_clocks_inord.txt || true
GFT_SCAN_KERNEL=scan2 python3 tools/probe_scan.py --docs 1000000 --unordered --modes 64,0 --reps 3 2>&1 | grep "scan debug\|GFT_SCAN" | tail -4 > $O/r2_scan_phase_clocks.txt || true
./tools/ubench/ubench > $O/r2_ubench_valu.txt 2>&1 || true
./tools/ubench/fbench > $O/r2_ubench_filter_lds.txt 2>&1 || true
./tools/ubench/gbench > $O/r2_ubench_gather.txt 2>&1 || true
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/sq1 $O/sq2 $O/stats_mixed $O/sq_mixed $O/stats_c5 $O/sq_c5
echo profiles collected
