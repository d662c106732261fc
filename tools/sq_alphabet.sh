#!/bin/bash
# SQ counters of the production scan kernel per document on the two alphabets of the workload generator (lower / mixed),
# filter only (GFT_SCAN_DEBUG=1 needs the timing-study instantiation: lower only) and whole kernel -> gpurun_out/sq_alphabet.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/sqa
rm -rf $O && mkdir -p $O
: > gpurun_out/sq_alphabet.txt
for a in lower mixed; do
  B="python3 tools/probe_scan.py --docs 500000 --unordered --alphabet $a --modes 0 --reps 2"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $O/$a -o run --output-format csv -- $B > $O/$a.log 2>&1 || exit 1
  echo "== $a" >> gpurun_out/sq_alphabet.txt
  python3 tools/sq_summary.py $O/$a --docs 500000 | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({k:{x:round(y,1) for x,y in v.items()} for k,v in d.items() if k.startswith('k_scan') and '<' in k},indent=1))" >> gpurun_out/sq_alphabet.txt
done
rm -rf $O
