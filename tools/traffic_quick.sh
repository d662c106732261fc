cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tq; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 5 --warmup 2 --cpu-docs 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o runc --output-format csv -- $B > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o runc --output-format csv -- $B > $O/w.log 2>&1
python3 tools/pmc_summary.py $O/f $O/w $O/traffic.json --docs 1000000 > $O/s.log 2>&1
python3 - <<'PY'
import json
p=json.load(open("gpurun_out/tq/traffic.json"))
for k,v in p["kernels"].items():
    if k.startswith("k_scan2"): print(k, v)
PY
python3 bench.py --steps 20 --warmup 3 --cpu-docs 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms_per_step'])"
rm -rf $O/f $O/w
