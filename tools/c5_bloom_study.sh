#!/bin/bash
# configs[4]'s scan with Bloom levels of several sizes in front of the global fingerprint table (GFT_SCAN5_BLOOM_KB)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/c5_bloom
mkdir -p $O
C5="--terms 100000 --exprs 1000 --inord 0.5 --docs 200000 --steps 5 --warmup 2"
for kb in 0 16 32 64; do
  echo "== GFT_SCAN5_BLOOM_KB=$kb" >> $O/log.txt
  GFT_SCAN_DEBUG=0 GFT_SCAN5_BLOOM_KB=$kb timeout -k 10 200 python3 bench.py $C5 --cpu-docs $([ $kb = 32 ] && echo 2000 || echo 0) > $O/out.txt 2> $O/err.txt || exit 1
  grep "build debug\] scan5" $O/err.txt | tail -1 >> $O/log.txt
  python3 -c "
import json
d=json.loads(open('$O/out.txt').read().strip().splitlines()[-1])
print(d['value'], d['kernels_ms_per_step'], d.get('parity'))" >> $O/log.txt 2>&1
done
