#!/bin/bash
# FETCH_SIZE of the scan kernel for the current environment (GFT_SCAN_KERNEL, GFT_SCAN4_ROUND ...), bytes per launch (x2 corrected)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/tq_$$
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-docs 0 > $O/log 2>&1
python3 - <<PY
import csv,glob,re
from collections import defaultdict
rows=defaultdict(list)
for p in glob.glob("$O/f/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        m=re.search(r"\b(k_scan\d\w*)",r["Kernel_Name"])
        if m and r["Counter_Name"]=="FETCH_SIZE": rows[m.group(1)].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
for k,v in rows.items():
    g=max(x[0] for x in v); vals=[x[1] for x in v if x[0]==g]
    print(k,"read GB per launch (x2):",round(2*sum(vals)/len(vals)*1024/1e9,2))
PY
rm -rf $O
