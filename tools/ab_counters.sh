#!/bin/bash
# same-box comparison of library builds beyond wall time: $@ = names (libgft_<name>.so; "cur" = libgft.so).  Per build: the scan
# kernel's phase clocks (GFT_SCAN_DEBUG=64), then two rocprofv3 --pmc passes over the scan probe (SQ instruction / wait
# counters; the vector-memory path) -> gpurun_out/abc/<name>_{phases.txt,sq.json,tcp.json}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/abc
mkdir -p $O
for n in "$@"; do
  lib=gofindthem_amd/libgft_$n.so; [ $n = cur ] && lib=gofindthem_amd/libgft.so
  export GFT_LIBRARY=$PWD/$lib
  timeout -k 10 200 python3 tools/probe_scan.py --docs 1000000 --unordered --modes 64,0 --reps 3 2>&1 | grep "scan debug\|GFT_SCAN\|ms" | tail -8 > $O/${n}_phases.txt
  P="python3 tools/probe_scan.py --docs 500000 --unordered --modes 0 --reps 2"
  rm -rf $O/tmp; mkdir -p $O/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/tmp/sq -o run --output-format csv -- $P > $O/tmp/sq.log 2>&1
  python3 tools/sq_summary.py $O/tmp/sq --docs 500000 > $O/${n}_sq.json 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $O/tmp/tcp -o run --output-format csv -- $P > $O/tmp/tcp.log 2>&1
  python3 tools/sq_summary.py $O/tmp/tcp --docs 500000 > $O/${n}_tcp.json 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_IFETCH -d $O/tmp/sq2 -o run --output-format csv -- $P > $O/tmp/sq2.log 2>&1
  python3 tools/sq_summary.py $O/tmp/sq2 --docs 500000 > $O/${n}_sq2.json 2>&1
  rm -rf $O/tmp
  echo "== $n"; cat $O/${n}_phases.txt
  python3 - $O $n <<'PY'
import json,sys
o,n=sys.argv[1:3]
for f in ("sq","sq2","tcp"):
    try:
        d=json.load(open("%s/%s_%s.json"%(o,n,f)))
        for k,v in d.get("kernels",d).items():
            if k.startswith("k_scan5"):
                print(f, {a:round(b,1) for a,b in v.items() if isinstance(b,(int,float))})
    except Exception as e:
        print(f,"no result",e)
PY
done
