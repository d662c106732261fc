#!/bin/bash
# A/B studies of the scan kernel: gofindthem_amd/libgft_$1.so = the current objects of libgft.so with gft_scan5.hip compiled
# again under the defines $2.. (run `python -m gofindthem_amd.build` first).  Same-box comparisons then run both libraries in
# one gpurun call: GFT_LIBRARY=gofindthem_amd/libgft_$1.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
CS=gofindthem_amd/csrc
python3 -m gofindthem_amd.build > /dev/null      # (the other objects: current)
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall "$@" -c $CS/gft_scan5.hip -o build/variants/gft_scan5_$NAME.o
objs=$(ls $CS/*.o | grep -v gft_scan5.hip.o)
hipcc -shared -o gofindthem_amd/libgft_$NAME.so $objs build/variants/gft_scan5_$NAME.o
echo gofindthem_amd/libgft_$NAME.so
