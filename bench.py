#!/usr/bin/env python3
"""bench.py -- ProcessText hot path on MI355X: documents/s and input GB/s, 10 k-term dictionary + 1 k expressions
over 1 M synthetic ~4 KB documents per GPU (BASELINE.json configs[2]; SURVEY.md 8(d) workload).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the rank's batch, inputs already resident in HBM:
Finder.ProcessDevice = work-unit setup + Aho-Corasick scan kernel + CSR gather + solver kernel -> hit bitmap,
plus (N > 1) one RCCL gather of every rank's bitmap to rank 0.  Documents are independent, so ranks own
disjoint document ranges and nothing else is exchanged: --scaling strong (default, the configuration the metric is quoted
on: 1 M documents in total, sharded across the N GPUs) or weak (1 M documents per GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (scan kernel, live HIP-event
timing) and `cpu_baseline` (the CPU oracle timed on this host, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=1_000_000, help="documents: in total (--scaling strong) or per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): the configuration the metric is quoted on -- --docs documents in total, sharded "
                         "across the GPUs (BASELINE.json configs[3]: 1 M documents across 8); weak: --docs documents per GPU")
    ap.add_argument("--terms", type=int, default=10_000)
    ap.add_argument("--exprs", type=int, default=1_000)
    ap.add_argument("--inord", type=float, default=0.0, help="fraction of INORD(...) expressions (config 4: 0.5)")
    ap.add_argument("--cpu-docs", type=int, default=-1, help="documents in the CPU baseline sample (-1 auto, 0 off)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--parity-docs", type=int, default=2048)
    ap.add_argument("--cu-margin", type=int, default=-1,
                    help="compute units left free of the engine's persistent kernels (gft_set_cu_margin); default: 0 on one GPU, "
                         "2 x N (at most 16) on N > 1, where RCCL's send / receive kernels of batch i's gather run beside batch i + 1")
    ap.add_argument("--alphabet", choices=["lower", "mixed"], default="lower",
                    help="lower: SURVEY.md 8(d) corpus over a-z (the configuration the metric is quoted on); mixed: the same "
                         "words with capitals, digits, punctuation and two-byte UTF-8 letters (> 48 byte classes, a real "
                         "word list's shape)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    from gofindthem_amd.engine import Engine  # noqa: F401  (fails loudly if libgft.so is missing)
    from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine
    from gofindthem_amd.workload import Workload, make_expressions
    from gofindthem_amd import _lib
    import ctypes as C

    # ---- workload ---------------------------------------------------------------------------------------
    t_setup = time.time()
    wl = Workload(args.terms, alphabet=args.alphabet)
    terms = wl.terms()
    exprs = make_expressions(terms, args.exprs, inord_fraction=args.inord, cover=True)
    finder = Finder(GpuEngine.__new__(GpuEngine), EmptyRgxEngine(), caseSensitive=False, device=local_rank)
    # (NewFinder(GpuEngine, EmptyRgxEngine, caseSensitive=false) as in BMDslSearch, benchmark_test.go:417; the
    # finder owns the gft_engine behind its GPU substring engine)
    finder.AddExpressions(exprs)
    # (the parser lower-cases the literals of a case-insensitive finder, dsl/parser.go:79-81: on the mixed alphabet a few
    # capitalised words fall onto their lower-case twins)
    n_keywords = len(finder.GetKeywords())
    assert n_keywords == args.terms or (args.alphabet == "mixed" and n_keywords > 0.97 * args.terms), "the expressions must reference the whole dictionary"
    finder.ForceBuild()
    L = _lib.load()
    eh = finder.engine_handle()
    stream = torch.cuda.current_stream().cuda_stream
    assert L.gft_set_stream(eh, stream) == 0
    # N > 1: the scan and solver kernels hold every CU they run on (one workgroup with the CU's whole LDS), so RCCL's kernels
    # would wait for a launch to end and then keep the next launch's workgroups waiting; a few CUs are left to them
    cu_margin = args.cu_margin if args.cu_margin >= 0 else (min(16, 2 * world) if world > 1 else 0)
    assert L.gft_set_cu_margin(eh, cu_margin) == 0
    from gofindthem_amd.sharding import BitmapGather, all_ranks_ok, max_over_ranks, shard_range, split_docs
    total_docs = args.docs if args.scaling == "strong" else args.docs * world
    if args.scaling == "strong":
        shards = split_docs(total_docs, world)       # contiguous document ranges (the synthetic documents are alike: equal
        first, my_docs = shards[rank]                # counts are equal bytes; a real corpus is cut by bytes, gft_split_docs)
        rows = max(n for _, n in shards)             # (gather buffers are equal-sized: the last shard may be one row short)
    else:
        first, my_docs = shard_range(rank, world, args.docs)
        rows = my_docs
    args.docs = my_docs                              # from here on: this rank's documents
    text, doc_off = wl.docs_device(first, args.docs, device=dev)
    words = (args.exprs + 31) // 32
    # two result buffers: the gather of batch i (the path's only exchange step, RCCL over xGMI) is in flight on the
    # process group's stream while batch i + 1 is scanned and solved; everything is complete before the closing fence
    bitmaps = [torch.zeros((rows, words), dtype=torch.int32, device=dev) for _ in range(2)]
    bitmap = bitmaps[0]
    gather = BitmapGather(bitmaps)         # rank 0 receives every rank's bitmap
    text_bytes = int(text.numel())
    n_steps_done = [0]

    # Two batches in flight (gft_process_device_begin / _end): batch i + 1 is enqueued before the host reads batch i's verdict
    # (the one 56-byte read-back of a step) and starts its gather, so neither the read-back nor the launches leave the device
    # idle -- a tenth of a step at 125 000 documents per GPU.  Every step's work is complete behind the closing fence.
    begun = []                             # slots of the batches begun and not ended yet

    def end_oldest():
        slot = begun.pop(0)
        finder.ProcessDeviceEnd()
        gather.start(slot)

    def step():
        slot = n_steps_done[0] % len(bitmaps)
        n_steps_done[0] += 1
        if slot in begun:                  # (one result buffer: the batch that owns it ends first)
            end_oldest()
        gather.wait(slot)                  # the previous exchange out of this buffer has landed
        finder.ProcessDeviceBegin(text.data_ptr(), doc_off.data_ptr(), args.docs, bitmaps[slot].data_ptr())
        begun.append(slot)
        while len(begun) > 1:
            end_oldest()

    def fence():
        while begun:
            end_oldest()
        gather.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # match volume H (drives the algorithmic bytes of the scan kernel)
    m = _lib.GftMatches()
    rc = L.gft_scan_device(eh, text.data_ptr(), doc_off.data_ptr(), args.docs, _lib.GFT_FOLD_ASCII, C.byref(m))
    assert rc == 0, L.gft_last_error(eh)
    n_matches = int(m.n_matches)
    # hits per document (the CSR offsets of that scan live in engine-owned device memory)
    hits_max = None
    try:
        moff = torch.empty(args.docs + 1, dtype=torch.int64, device=dev)
        hip = C.CDLL("libamdhip64.so")
        if hip.hipMemcpy(C.c_void_p(moff.data_ptr()), C.c_void_p(m.match_off), C.c_size_t(8 * (args.docs + 1)), C.c_int(3)) == 0:
            hits_max = int((moff[1:] - moff[:-1]).max().item())
    except OSError:
        pass
    from gofindthem_amd.workload import read_ceiling_gbps
    ceiling_gbps = read_ceiling_gbps(text)
    setup_s = time.time() - t_setup

    # ---- timed region -------------------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    fence()                                # (the warm-up batches are complete -- a rerun of the last one included -- before anything is counted)
    # (mode 2: HIP events around the scan kernel's launches only -- the kernel the roofline prices.  Every bracket is two
    # event nodes on the stream, a few microseconds between two kernels each; the solver's and the auxiliary kernels'
    # times are read from the same step run again behind the timed region, below)
    L.gft_profile_enable(eh, 2)
    L.gft_profile_reset(eh)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    L.gft_profile_enable(eh, 0)
    elapsed = max_over_ranks(elapsed, dev)

    def prof(name):
        ms, n = C.c_double(), C.c_uint64()
        L.gft_profile_read(eh, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, int(n.value)
    scan_ms, scan_n = prof("scan")
    # the spread of the dominant kernel from launch to launch: the same step again, every launch read on its own (behind the
    # timed region -- reading the events synchronises)
    scan_each = []
    solve_ms = aux_ms = 0.0
    again_steps = min(args.steps, 20)
    L.gft_profile_enable(eh, 1)
    for _ in range(again_steps):
        L.gft_profile_reset(eh)
        step()
        fence()
        ms1, n1 = prof("scan")
        if n1:
            scan_each.append(ms1 / n1)
        solve_ms += prof("solve")[0]
        aux_ms += prof("aux")[0]
    L.gft_profile_enable(eh, 0)

    # ---- same-run consistency that needs no oracle: the corpus generator's host and device forms agree on a sample,
    # and (N > 1) rank 0's slice of the gathered result is what it computed itself --------------------------------
    bitmap = bitmaps[(n_steps_done[0] - 1) % len(bitmaps)]     # what the last step wrote
    S = min(args.parity_docs, args.docs)
    h_text, h_off = wl.docs_host(first, S)
    d_off = doc_off[:S + 1].cpu().numpy().astype(np.uint64)
    gen_ok = bool(np.array_equal(d_off, h_off)) and bool(
        np.array_equal(text[:int(h_off[-1])].cpu().numpy(), h_text))
    parity_ok = all_ranks_ok(gen_ok, dev)
    parity = "not checked in this run (the CPU oracle runs in the cpu_baseline leg, N = 1 only); tests/ -m gpu hold the parity proofs"
    if world > 1:
        # the exchange itself: every rank's checksum of what it computed against rank 0's checksum of what it received
        # (one tiny all_gather, outside the timed region)
        from gofindthem_amd.sharding import verify_gather
        last = (n_steps_done[0] - 1) % len(bitmaps)
        arrived, n_shards = verify_gather(gather, last, args.docs)
        arrived = all_ranks_ok(arrived, dev)
        parity_ok = parity_ok and arrived
        bitmap = bitmaps[last]
        parity = ("gathered bitmap verified against the per-rank checksums of all %d shards (the CPU oracle runs in the cpu_baseline "
                  "leg, N = 1 only)" % n_shards) if arrived else "MISMATCH between a rank's bitmap and what rank 0 received"

    # ---- CPU baseline (rank 0, N == 1): the oracle = our restatement of the reference path, on this host ---------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_docs != 0:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, args.cpu_threads)      # a 1-GPU box's CPU share is 16 cores
        n_cpu = args.cpu_docs if args.cpu_docs > 0 else min(args.docs, 24000 * cores)   # ~10-15 s of CPU work
        n_cpu = min(n_cpu, args.docs)
        from oracle.pyoracle import Oracle     # the checker: only this leg touches oracle/
        orc = Oracle(sorted(k.encode("utf-8") for k in finder.GetKeywords()))   # BuildEngine receives the finder's keyword set
        orc.set_expressions(exprs, case_sensitive=False)
        c_off = doc_off[:n_cpu + 1].cpu().numpy().astype(np.uint64)
        c_text = text[:int(c_off[-1])].cpu().numpy()
        n1 = min(n_cpu, 6000)                   # the reference's own execution model: one goroutine (benchmark_test.go:422-425)
        t1 = time.perf_counter()
        orc.process(c_text[:int(c_off[n1])], c_off[:n1 + 1], fold=True, n_threads=1)
        t1 = time.perf_counter() - t1
        tc = time.perf_counter()
        ref = orc.process(c_text, c_off, fold=True, n_threads=cores)
        tc = time.perf_counter() - tc
        same = bool(np.array_equal(ref, bitmap[:n_cpu].cpu().numpy().view(np.uint32)))
        parity_ok = parity_ok and same
        parity = ("bit-exact vs CPU oracle on the first %d documents" % n_cpu) if same else "MISMATCH vs CPU oracle"
        cpu = {"value": n_cpu / tc, "unit": "docs/s", "cores": cores, "kind": "port",
               "single_thread_docs_per_s": n1 / t1,
               "input_GBps": float(c_off[-1]) / tc / 1e9,
               "sample": "first %d documents of the same corpus (%.1f MB), oracle/ac_oracle.cpp ProcessText "
                         "restatement, %d std::thread workers, %.1f s; bitmap equal to the GPU's: %s"
                         % (n_cpu, float(c_off[-1]) / 1e6, cores, tc, same)}

    if rank == 0:
        import shutil
        # the unmodified reference is timed beside the path only where a Go toolchain and a module cache exist
        go_ref = ("Go reference not runnable (no `go` binary on this box, and the module github.com/pedroegsilva/ahocorasick is not "
                  "vendored); CPU baseline = C++ restatement (oracle/ac_oracle.cpp)") if shutil.which("go") is None else \
                 "a `go` binary exists here, but /root/reference does not travel to the GPU box: CPU baseline = C++ restatement (oracle/ac_oracle.cpp)"
        # which BASELINE.json configuration the arguments describe
        if args.alphabet != "lower":
            cfg_name = "BASELINE.json configs[2] shape on the MIXED alphabet (not a BASELINE configuration)"
        elif args.terms == 10_000 and args.exprs == 1_000 and args.inord == 0:
            cfg_name = "BASELINE.json configs[2]" if world == 1 else "BASELINE.json configs[2] sharded over %d GPUs" % world
        elif args.terms == 10_000 and args.exprs == 1_000:
            cfg_name = "BASELINE.json configs[3]" + (" (one GPU's part of it)" if world == 1 else "")
        elif args.terms == 100_000:
            cfg_name = "BASELINE.json configs[4], device half (no regex leaves)"
        elif args.terms == 1_000:
            cfg_name = "BASELINE.json configs[1] dictionary through the ProcessText path"
        else:
            cfg_name = "custom (not a BASELINE configuration)"
        scan_kernel = (L.gft_scan_kernel(eh) or b"").decode()
        kernel_names = {"scan5": "k_scan5 (suffix-window scan, one probe per two bytes)", "scan2": "k_scan2 (suffix-window scan)", "scan4": "k_scan4 (streaming suffix-window scan)", "scan3": "k_scan3 (stride-2 suffix-window scan)", "dfa": "k_scan_units (two-tier DFA)"}
        docs_total = total_docs * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        # algorithmic bytes of the dominant (scan) kernel per launch: text once + one offset entry per document +
        # 8 bytes per match written
        # (u32 term id + u32 position per match; without INORD expressions the path is presence-only and
        # writes the term ids alone, so only those 4 bytes count)
        per_match = 8 if args.inord > 0 else 4
        alg_bytes = text_bytes + 8 * args.docs + per_match * n_matches
        scan_avg_ms = scan_ms / max(scan_n, 1)
        achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_n else 0.0
        # HBM traffic of the scan kernel from rocprofv3 PMC passes (profiles/r3_pmc_traffic.json; it cannot be
        # collected from inside this process) -- only attached when this run uses the profiled configuration
        traffic = None
        for name in ("r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json"):   # (the newest set that has this kernel)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pmc = json.load(fh)
                if args.alphabet == "lower" and pmc["config"] == {"docs": args.docs, "terms": args.terms, "exprs": args.exprs, "inord": args.inord}:
                    traffic = next(v["traffic_bytes"] for k, v in pmc["kernels"].items() if k.startswith("k_" + scan_kernel))
                    break
            except (OSError, KeyError, ValueError, StopIteration):
                pass
        out = {
            "metric": "ProcessText throughput: documents/s (and input GB/s), %d-term dictionary + %d expressions, "
                      "%d docs of ~4 KB %s" % (args.terms, args.exprs, total_docs if args.scaling == "strong" else args.docs,
                                               "sharded over %d GPU(s)" % world if args.scaling == "strong" else "per GPU"),
            "value": docs_total / elapsed,
            "unit": "docs/s",
            "input_GBps": text_bytes * (total_docs / max(args.docs, 1)) * args.steps / elapsed / 1e9,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d terms + %d %s expressions, %d docs x ~4 KB on this GPU "
                                   "(SURVEY.md 8(d) generator%s), case-insensitive finder, inputs resident in HBM"
                                   % (cfg_name, args.terms, args.exprs, "AND/OR/NOT" if args.inord == 0 else
                                      "AND/OR/NOT + %.0f%% INORD" % (args.inord * 100), args.docs,
                                      "" if args.alphabet == "lower" else ", MIXED alphabet: capitals, digits, punctuation, UTF-8 letters"),
                       "alphabet": args.alphabet, "keywords": n_keywords,
                       "docs_total": total_docs, "docs_per_gpu": args.docs, "text_bytes_per_gpu": text_bytes, "matches_per_gpu": n_matches,
                       "matches_per_doc": n_matches / args.docs, "matches_per_doc_max": hits_max,
                       "parallelism": "docs sharded x%d" % world, "cu_margin": cu_margin},
            "roofline": {"bound": "hbm", "kernel": kernel_names.get(scan_kernel, scan_kernel), "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "measured_read_ceiling": ceiling_gbps, "frac_of_measured_ceiling": achieved / ceiling_gbps,
                         "guide_achievable_read": 6300.0, "frac_of_guide_achievable": achieved / 6300.0,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n,
                         "scan_ms_min": float(np.min(scan_each)) if scan_each else None,
                         "scan_ms_median": float(np.median(scan_each)) if scan_each else None,
                         "scan_ms_stddev": float(np.std(scan_each)) if scan_each else None,
                         "scan_ms_spread_launches": len(scan_each)},
            "kernels_ms_per_step": {"scan": scan_ms / args.steps, "solve": solve_ms / max(again_steps, 1),
                                    "aux(units+prefix sums+gather)": aux_ms / max(again_steps, 1),
                                    "how": "scan: HIP events inside the timed region; solve, aux: the same step run %d more "
                                           "times behind it with every launch bracketed" % again_steps},
            "cpu_baseline": cpu,
            "go_reference": go_ref,
            "parity": parity if parity_ok else "PARITY FAILED",
            "setup_s": setup_s,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
