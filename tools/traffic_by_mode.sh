#!/bin/bash
# HBM-side reads of the production scan kernel under its timing-study knock-outs: which phase fetches the bytes beyond the text?
# TCC_EA0_RDREQ (128-byte line fills) / TCC_MISS / TCC_HIT per document (the counter set of tools/collect_profiles_r3.sh; FETCH_SIZE in
# the same pass hangs rocprofv3 on this image) for GFT_SCAN_DEBUG = 0 (production), 4 (no bucket table), 8 (no short-term
# records), 12 (both), 1 (filter only) -> gpurun_out/traffic_modes.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/tm
rm -rf $O && mkdir -p $O
export GFT_SCAN_KERNEL=scan5
: > gpurun_out/traffic_modes.txt
for mode in ${MODES:-0 4 12}; do
  export GFT_SCAN_DEBUG=$mode
  timeout -k 10 70 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/p$mode -o run --output-format csv -- python3 tools/probe_scan.py --docs 500000 --unordered --modes $mode --reps 2 > $O/p$mode.log 2>&1 || { echo "mode $mode failed" >> gpurun_out/traffic_modes.txt; break; }
  python3 tools/sq_summary.py $O/p$mode --docs 500000 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('k_scan5'): print('mode $mode', k, {a: round(b, 2) for a, b in v.items()})" >> gpurun_out/traffic_modes.txt 2>&1
done
rm -rf $O
cat gpurun_out/traffic_modes.txt
