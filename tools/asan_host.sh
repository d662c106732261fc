#!/bin/bash
# Address/UB-sanitizer run of libgft's HOST code (table compilers, DSL compiler, JSON reader, finder and group mirrors, the
# C ABI) on the CPU: builds build/asan/libgft.so with -fsanitize=address,undefined (device code objects are compiled as
# usual: GPU ASan is not available on this pool) and runs the `-m "not gpu"` suite against it.
#   tools/asan_host.sh            # from the repo root; exit code = pytest's
set -e
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p $OUT
CS=gofindthem_amd/csrc
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
objs=""
for f in gft_kernels.hip gft_solve.hip gft_scan3.hip gft_scan5.hip; do
  o=$OUT/$f.o
  [ $o -nt $CS/$f ] || hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -c $CS/$f -o $o
  objs="$objs $o"
done
for f in gft_api.cpp ac_tables.cpp scan2_tables.cpp scan3_tables.cpp dsl_compile.cpp finder_host.cpp json_mini.cpp group_host.cpp host_solve.cpp; do
  o=$OUT/$f.o
  hipcc -x hip --offload-arch=gfx950 -O1 -std=c++17 -fPIC $SAN -c $CS/$f -o $o
  objs="$objs $o"
done
hipcc -shared $SAN -shared-libsan -o $OUT/libgft.so $objs
RT=$(dirname $(hipcc -print-file-name=libclang_rt.asan-x86_64.so))
GFT_LIBRARY=$PWD/$OUT/libgft.so LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
