"""In-tree build of the HIP/C++ pieces for gfx950 (no cmake needed: a handful of hipcc/g++ invocations).

    python -m gofindthem_amd.build          # builds libgft.so (product) and libgfworkload.so (bench utility)

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only container; the built
.so files are git-ignored but travel to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgft.so")
ARCH = "gfx950"

# the product library carries the two production scan kernels (scan5, scan3) and the DFA kernel (an independent algorithm,
# the cross-check); GFT_EXTRA_KERNELS=1 adds the earlier suffix-window kernels scan2 / scan4 (tools/ studies, the opt-in
# cross-check job of tests/test_gpu_parity.py)
EXTRA = os.environ.get("GFT_EXTRA_KERNELS", "0") == "1"
HIP_SOURCES = ["gft_kernels.hip", "gft_solve.hip", "gft_scan3.hip", "gft_scan5.hip"] + (["gft_scan2.hip", "gft_scan4.hip"] if EXTRA else [])
STAMP = os.path.join(CSRC, ".build_flags")
CXX_SOURCES = ["gft_api.cpp", "ac_tables.cpp", "scan2_tables.cpp", "scan3_tables.cpp", "dsl_compile.cpp", "finder_host.cpp", "json_mini.cpp", "group_host.cpp", "host_solve.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES + CXX_SOURCES if os.path.exists(os.path.join(CSRC, f))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "gft.h"))
    flags = "extra=%d" % int(EXTRA)
    if not os.path.exists(STAMP) or open(STAMP).read() != flags:
        force = True                 # (another set of kernels than the objects on disk were built for)
    if not force and not _newer(LIB, srcs + hdrs):
        return LIB
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s) + ".o")
        if force or _newer(o, [s] + hdrs):
            cmd = ["hipcc", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", s, "-o", o]
            if EXTRA:
                cmd.insert(1, "-DGFT_EXTRA_KERNELS")
            if s.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]   # host code that includes hip_runtime.h; no kernels inside
            jobs.append(cmd)
        objs.append(o)
    if jobs:
        # (a handful of translation units, the kernels take a minute each: compile them side by side)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    cmd = ["hipcc", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(flags)
    return LIB


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    from . import workload
    wl = workload.build(force)
    return lib, wl


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
