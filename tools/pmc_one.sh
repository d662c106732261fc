#!/bin/bash
# one rocprofv3 --pmc pass over the scan probe: $1 = kernel (scan2 / scan5), $2.. = counters -> gpurun_out/pmc1_$1_<first counter>.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
K=$1; shift
O=gpurun_out/pmc1_tmp
rm -rf $O && mkdir -p $O
export GFT_SCAN_KERNEL=$K
timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/p -o run --output-format csv -- python3 tools/probe_scan.py --docs 500000 --unordered --modes 0 --reps 2 > $O/p.log 2>&1
echo "rc=$?" >> $O/p.log
python3 tools/sq_summary.py $O/p --docs 500000 | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({k:v for k,v in d.items() if k.startswith('k_scan') and '<' in k},indent=1))" > gpurun_out/pmc1_${K}_$1.json 2>&1
tail -2 $O/p.log >> gpurun_out/pmc1_${K}_$1.json
rm -rf $O
