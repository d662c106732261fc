#!/bin/bash
# same-box A/B of library builds on the headline configuration: $@ = names (libgft_<name>.so; "cur" = libgft.so); three rounds
# interleaved so that clock drift hits every build alike -> gpurun_out/ab/<name>_<round>.json, one summary line per run
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/ab
ARGS=${AB_ARGS:---steps 20 --warmup 3 --cpu-docs 0}
for r in 1 2 3; do
  for n in "$@"; do
    lib=gofindthem_amd/libgft_$n.so; [ $n = cur ] && lib=gofindthem_amd/libgft.so
    GFT_LIBRARY=$PWD/$lib timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/ab/${n}_$r.json 2> gpurun_out/ab/${n}_$r.err || { echo "$n round $r FAILED"; tail -3 gpurun_out/ab/${n}_$r.err; }
    python3 - $n $r <<'PY'
import json,sys
n,r=sys.argv[1:3]
try:
    d=json.load(open("gpurun_out/ab/%s_%s.json"%(n,r)))
    k=d["kernels_ms_per_step"]
    print("%-10s round %s: step %.3f ms  scan %.3f (min %.3f)  solve %.3f  %s" % (n, r, d["ms_per_step"], k["scan"], d["roofline"]["scan_ms_min"], k["solve"], d["parity"][:40]))
except Exception as e:
    print(n, r, "no result:", e)
PY
  done
done
