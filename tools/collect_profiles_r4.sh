#!/bin/bash
# Round 4: collects the rocprofv3 evidence on the GPU box (run through gpurun from the repo root), part $1 = a | b | c.
#   a: the headline configuration on the production kernel (k_scan5): bench line (with launch spread), kernel-trace stats,
#      FETCH_SIZE / WRITE_SIZE (separate passes), TCC hit / miss / RDREQ, SQ counters (two passes), the vector-memory path
#      (TCP pending stall, L1 -> L2 requests and their latency), phase clocks, SQ counters under the knock-outs (SALU by phase)
#   b: one GPU's share of configs[3] (--docs 125000) with kernel stats; configs[3]'s INORD mix; the mixed alphabet;
#      configs[0]; the over-limit INORD study; the host-memory path
#   c: BASELINE configs[4]'s device half with its traffic counters
# Outputs land in gpurun_out/r4/; the summaries are copied into profiles/ afterwards (this script does not touch it).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r4
mkdir -p $O
B="python3 bench.py --steps 5 --warmup 2 --cpu-docs 0"
pmc() {   # $1 = output dir, rest = counters; the program itself follows `--` (no env / shell hop under the profiler)
  d=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $d -o run --output-format csv -- $RUN > $d.log 2>&1
}
if [ "$1" = "a" ]; then
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/r4_final_bench.json
RUN=$B
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r4_final_kernel_stats.csv \;
pmc $O/pmc_fetch FETCH_SIZE
pmc $O/pmc_write WRITE_SIZE
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r4_pmc_traffic.json --docs 1000000 > $O/pmc_summary.log 2>&1
pmc $O/pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 tools/sq_summary.py $O/pmc_tcc > $O/r4_tcc_counters.json
pmc $O/sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
python3 tools/sq_summary.py $O/sq1 --docs 1000000 > $O/r4_sq_counters_a.json
pmc $O/sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM
python3 tools/sq_summary.py $O/sq2 --docs 1000000 > $O/r4_sq_counters_b.json
pmc $O/tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
python3 tools/sq_summary.py $O/tcp --docs 1000000 > $O/r4_vmem_counters.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/sq1 $O/sq2 $O/tcp
python3 tools/probe_scan.py --docs 1000000 --unordered --modes 64,0,1 --reps 3 2>&1 | grep "scan debug\|GFT_SCAN" | tail -6 > $O/r4_scan_phase_clocks.txt || true
# SALU / VALU / LDS by phase: the timing-study knock-outs (1 filter only, 12 no shorts + no buckets, 4 no buckets, 8 no shorts)
for m in 0 1 12 4 8; do
  bash tools/sq_debug_modes.sh scan5 $m || true
  cp gpurun_out/sqd_scan5_$m.json $O/r4_sq_scan5_mode_$m.json 2>/dev/null || true
done
echo part a collected
elif [ "$1" = "b" ]; then
python3 bench.py --steps 40 --warmup 5 --docs 125000 --cpu-docs 20000 > $O/bench_share.log 2>&1
tail -1 $O/bench_share.log > $O/r4_share_bench.json
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --docs 125000 --cpu-docs 0 > $O/stats_share.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r4_share_kernel_stats.csv \;
rm -rf $O/stats
python3 bench.py --steps 10 --warmup 2 --inord 0.5 --cpu-docs 20000 > $O/bench_inord.log 2>&1
tail -1 $O/bench_inord.log > $O/r4_inord_bench.json
python3 bench.py --steps 10 --warmup 2 --alphabet mixed > $O/bench_mixed.log 2>&1
tail -1 $O/bench_mixed.log > $O/r4_mixed_bench.json
python3 tools/bench_c1.py > $O/r4_c1_bench.json 2> $O/c1.log
GFT_SOLVE_DEBUG=8 python3 tools/bench_c1.py 2>&1 | grep "solve debug" | tail -18 > $O/r4_c1_solver_phase_clocks.txt || true
timeout -k 10 300 python3 tools/bench_overlimit.py --json $O/r4_overlimit.json > $O/overlimit.log 2>&1
python3 tools/bench_latency.py --batch-docs 250000 --reps 50 > $O/r4_host_path.json 2> $O/lat.log
echo part b collected
else
C5="--terms 100000 --exprs 1000 --inord 0.5 --docs 200000"
python3 bench.py --steps 5 --warmup 2 $C5 --cpu-docs 2000 > $O/bench_c5.log 2>&1
tail -1 $O/bench_c5.log > $O/r4_c5_bench.json
RUN="python3 bench.py --steps 3 --warmup 1 --cpu-docs 0 $C5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c5 -o run --output-format csv -- $RUN > $O/stats_c5.log 2>&1
find $O/stats_c5 -name "*kernel_stats.csv" -exec cp {} $O/r4_c5_kernel_stats.csv \;
pmc $O/pmc_fetch FETCH_SIZE
pmc $O/pmc_write WRITE_SIZE
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r4_c5_pmc_traffic.json --docs 200000 > $O/pmc_summary_c5.log 2>&1
rm -rf $O/stats_c5 $O/pmc_fetch $O/pmc_write
echo part c collected
fi
