#!/bin/bash
# Round 3: collects the rocprofv3 evidence on the GPU box (run through gpurun from the repo root), part $1 = a | b.
#   a: the headline configuration on the production kernel (k_scan5) and, for comparison, on k_scan2 (GFT_SCAN_KERNEL=scan2;
#      the streaming kernel's set, r3_scan4_*, was collected the same way earlier in the round):
#      bench lines, kernel-trace stats, FETCH_SIZE / WRITE_SIZE (separate passes), TCC hit/miss/RDREQ, SQ counters, phase clocks
#   b: BASELINE configs[4]'s device half with its traffic counters, the INORD and mixed-alphabet lines, configs[0], the
#      host-memory path
# Outputs land in gpurun_out/r3/; the summaries are copied into profiles/ afterwards (this script does not touch it).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
O=gpurun_out/r3
mkdir -p $O
B="python3 bench.py --steps 5 --warmup 2 --cpu-docs 0"
if [ "$1" = "a" ]; then
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/r3_final_bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r3_final_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o runc --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o runc --output-format csv -- $B > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r3_pmc_traffic.json --docs 1000000 > $O/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_tcc -o runc --output-format csv -- $B > $O/pmc_tcc.log 2>&1
python3 tools/sq_summary.py $O/pmc_tcc > $O/r3_tcc_counters.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq1 -o run --output-format csv -- $B > $O/sq1.log 2>&1
python3 tools/sq_summary.py $O/sq1 --docs 1000000 > $O/r3_sq_counters_a.json
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/sq2 -o run --output-format csv -- $B > $O/sq2.log 2>&1
python3 tools/sq_summary.py $O/sq2 --docs 1000000 > $O/r3_sq_counters_b.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/sq1 $O/sq2
# the one-probe-per-byte kernel (GFT_SCAN_KERNEL=scan2), same configuration
export GFT_SCAN_KERNEL=scan2
python3 bench.py --steps 20 --warmup 3 --cpu-docs 0 > $O/bench_scan2.log 2>&1
tail -1 $O/bench_scan2.log > $O/r3_scan2_bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats2.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r3_scan2_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o runc --output-format csv -- $B > $O/pmc_fetch2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o runc --output-format csv -- $B > $O/pmc_write2.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r3_scan2_pmc_traffic.json --docs 1000000 > $O/pmc_summary2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq1 -o run --output-format csv -- $B > $O/sq12.log 2>&1
python3 tools/sq_summary.py $O/sq1 --docs 1000000 > $O/r3_scan2_sq_counters_a.json
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/sq2 -o run --output-format csv -- $B > $O/sq22.log 2>&1
python3 tools/sq_summary.py $O/sq2 --docs 1000000 > $O/r3_scan2_sq_counters_b.json
unset GFT_SCAN_KERNEL
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/sq1 $O/sq2
python3 tools/probe_scan.py --docs 1000000 --unordered --modes 64,0,1 --reps 3 2>&1 | grep "scan debug\|GFT_SCAN" | tail -6 > $O/r3_scan_phase_clocks.txt || true
GFT_SCAN_KERNEL=scan2 python3 tools/probe_scan.py --docs 1000000 --unordered --modes 64,0,1 --reps 3 2>&1 | grep "scan debug\|GFT_SCAN" | tail -6 > $O/r3_scan2_phase_clocks.txt || true
echo part a collected
else
# BASELINE configs[4]'s device half: 100 000 terms, 1 000 expressions with INORD, 200 000 documents
C5="--terms 100000 --exprs 1000 --inord 0.5 --docs 200000"
python3 bench.py --steps 5 --warmup 2 $C5 --cpu-docs 2000 > $O/bench_c5.log 2>&1
tail -1 $O/bench_c5.log > $O/r3_c5_bench.json
BC="python3 bench.py --steps 3 --warmup 1 --cpu-docs 0 $C5"
rocprofv3 --kernel-trace --stats -d $O/stats_c5 -o run --output-format csv -- $BC > $O/stats_c5.log 2>&1
find $O/stats_c5 -name "*kernel_stats.csv" -exec cp {} $O/r3_c5_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o runc --output-format csv -- $BC > $O/pmc_fetch_c5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o runc --output-format csv -- $BC > $O/pmc_write_c5.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r3_c5_pmc_traffic.json --docs 200000 > $O/pmc_summary_c5.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_tcc -o runc --output-format csv -- $BC > $O/pmc_tcc_c5.log 2>&1
python3 tools/sq_summary.py $O/pmc_tcc > $O/r3_c5_tcc_counters.json
rm -rf $O/stats_c5 $O/pmc_fetch $O/pmc_write $O/pmc_tcc
python3 bench.py --steps 10 --warmup 2 --inord 0.5 --cpu-docs 20000 > $O/bench_inord.log 2>&1
tail -1 $O/bench_inord.log > $O/r3_inord_bench.json
python3 bench.py --steps 10 --warmup 2 --alphabet mixed > $O/bench_mixed.log 2>&1
tail -1 $O/bench_mixed.log > $O/r3_mixed_bench.json
python3 tools/bench_c1.py > $O/r3_c1_bench.json 2> $O/c1.log
python3 tools/bench_latency.py --batch-docs 250000 --reps 50 > $O/r3_host_path.json 2> $O/lat.log
GFT_HOST_TIMING=1 python3 tools/bench_latency.py --batch-docs 250000 --reps 2 2>&1 | grep "host timing" | tail -3 > $O/r3_host_path_phases.txt || true
python3 tools/probe_pcie.py 2>&1 | grep "GB/s" > $O/r3_pcie_probe.txt || true
# the vector-memory path and the instruction cache of the production scan kernel (one pass per counter group)
for grp in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  bash tools/pmc_one.sh scan5 $grp || true
done
echo part b collected
fi
