#!/bin/bash
# configs[4]'s solver under its phase clocks and knock-outs (GFT_SOLVE_DEBUG), with and without INORD expressions
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/solve_c5
mkdir -p $O
C5="--terms 100000 --exprs 1000 --docs 200000 --cpu-docs 0 --steps 3 --warmup 1"
for inord in 0.5 0; do
  for dbg in 0 8 1 2; do
    echo "== inord $inord GFT_SOLVE_DEBUG=$dbg" >> $O/log.txt
    GFT_SOLVE_DEBUG=$dbg timeout -k 10 120 python3 bench.py $C5 --inord $inord > $O/out.txt 2> $O/err.txt
    grep "solve debug" $O/err.txt | tail -19 >> $O/log.txt
    python3 -c "
import json,sys
d=json.loads(open('$O/out.txt').read().strip().splitlines()[-1])
print(d['kernels_ms_per_step'], d.get('parity'))" >> $O/log.txt 2>&1
  done
done
