"""N > 1: document sharding + the bitmap gather (gofindthem_amd/sharding.py), world_size 2.

CPU leg (gloo): the exchange logic runs for real; per-rank bitmaps come from the CPU oracle (test infrastructure --
the product has no CPU path).  GPU leg: two processes share the one card of the GPU box, each runs the HIP path on
its own shard, rank 0 checks the gathered result against the oracle over the whole range.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_DOCS, N_TERMS, N_EXPRS = 96, 300, 70


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, use_gpu, out_path, backend="gloo"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # (the process group comes first: nothing has touched the GPU yet when the launcher's rendezvous runs)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = "cuda:0" if backend == "nccl" else "cpu"
    force = world == 1            # a group of one rank takes the collective path on request
    from gofindthem_amd.sharding import BitmapGather, all_ranks_ok, max_over_ranks, shard_range
    from gofindthem_amd.workload import Workload, make_expressions
    from oracle.pyoracle import Oracle

    wl = Workload(N_TERMS)
    terms = wl.terms()
    exprs = make_expressions(terms, N_EXPRS, inord_fraction=0.4, cover=True)
    per = N_DOCS // world
    first, n = shard_range(rank, world, per)
    text, off = wl.docs_host(first, n)
    if use_gpu:
        from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine
        f = Finder(GpuEngine.__new__(GpuEngine), EmptyRgxEngine(), False, device=0)
        f.AddExpressions(exprs)
        local = torch.from_numpy(f.ProcessTexts(blob=text, doc_off=off).view(np.int32)).to(dev)
    else:
        o = Oracle(terms)
        o.set_expressions(exprs, False)
        local = torch.from_numpy(o.process(text, off, fold=True).view(np.int32)).to(dev)
    g = BitmapGather(local, force_collective=force)
    assert g.collective
    g()
    # pipelined form (bench.py, N > 1): two result buffers, gathers in flight while the next batch is computed
    other = local ^ 0x55
    gp = BitmapGather([local.clone(), other], force_collective=force)
    gp.start(0)
    gp.start(1)
    gp.wait(0)
    gp.drain()
    pipelined_ok = True
    if rank == 0:
        pipelined_ok = bool(torch.equal(gp.full(0), g.full())) and bool(torch.equal(gp.full(1), g.full() ^ 0x55))
    # bench.py's N > 1 parity: per-rank checksums against what rank 0 received -- and it notices a damaged shard
    from gofindthem_amd.sharding import verify_gather
    arrived, n_shards = verify_gather(gp, 1, n)
    pipelined_ok = pipelined_ok and arrived and n_shards == world
    if rank == 0:
        gp.slot_parts[1][world - 1][n // 2, 0] ^= 1 << 7
    damaged, _ = verify_gather(gp, 1, n)
    if rank == 0:
        pipelined_ok = pipelined_ok and not damaged
    ok = all_ranks_ok(pipelined_ok, dev)
    t = max_over_ranks(float(rank), dev)
    if rank == 0:
        full = g.full().cpu().numpy().view(np.uint32)
        o = Oracle(terms)
        o.set_expressions(exprs, False)
        wtext, woff = wl.docs_host(0, per * world)
        want = o.process(wtext, woff, fold=True)
        np.save(out_path, np.array([int(np.array_equal(full, want)), int(ok), int(t == world - 1), full.shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


def _run(use_gpu, tmp_path, world=2, backend="gloo"):
    out = str(tmp_path / "res.npy")
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, use_gpu, out, backend), nprocs=world, join=True, start_method="spawn")
    res = np.load(out)
    assert res.tolist() == [1, 1, 1, N_DOCS // world * world]


def test_gather_two_ranks_gloo_cpu(tmp_path):
    _run(False, tmp_path)


def test_gather_one_rank_takes_the_collective_path_on_request_gloo_cpu(tmp_path):
    """BitmapGather(force_collective=True) / verify_gather in a group of ONE rank: the exchange goes through the process
    group instead of being skipped (the form the one-rank nccl test below runs on the GPU box)"""
    _run(False, tmp_path, world=1)


@pytest.mark.gpu
def test_gather_two_ranks_gpu_compute(tmp_path):
    _run(True, tmp_path)


@pytest.mark.gpu
def test_gather_one_rank_nccl_is_rccl(tmp_path):
    """ONE rank with backend "nccl" (= RCCL on ROCm; the process group is created before anything
    touches the GPU) runs the HIP path on its shard and sends its bitmap through start / wait / drain / verify_gather --
    dist.gather and all_gather of device tensors over RCCL -- and rank 0 checks the result against the oracle."""
    _run(True, tmp_path, world=1, backend="nccl")


def test_split_docs():
    from gofindthem_amd.sharding import shard_range, split_docs
    assert split_docs(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert split_docs(0, 2) == [(0, 0), (0, 0)]
    assert shard_range(3, 8, 1000) == (3000, 1000)


def test_split_docs_by_bytes():
    """contiguous ranges, every document exactly once, byte-balanced: one huge document does not drag its neighbours along"""
    import numpy as np
    from gofindthem_amd.sharding import split_docs_by_bytes
    lens = [10] * 50 + [5000] + [10] * 49
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 4, 8):
        parts = split_docs_by_bytes(off, world)
        assert len(parts) == world
        assert parts[0][0] == 0 and sum(n for _, n in parts) == 100
        for (a, n), (b, _) in zip(parts, parts[1:]):
            assert a + n == b
    two = split_docs_by_bytes(off, 2)
    assert two[0] == (0, 50) or two[0] == (0, 51)                      # the 5 000-byte document is the border
    eq = split_docs_by_bytes(np.arange(0, 4100 * 1001, 4100, dtype=np.uint64), 4)
    assert [n for _, n in eq] == [250, 250, 250, 250]
    assert split_docs_by_bytes(np.array([7], dtype=np.uint64), 3) == [(0, 0)] * 3
    assert split_docs_by_bytes(np.array([0, 0, 0], dtype=np.uint64), 2)[0][0] == 0
