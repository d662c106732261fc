"""Document sharding across ranks (one process per GPU) and the single exchange step of the path.

ProcessText keeps no state between documents (finder/finder.go:139-179), so the corpus is cut into contiguous
document ranges, one per rank; the compiled dictionary and expression programs are replicated.  The only exchange
is the gather of every rank's expression-hit bitmap to rank 0 -- one collective per batch (RCCL over xGMI when the
process group's backend is "nccl"; the same code runs over gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, docs_per_rank):
    """weak scaling: every rank owns `docs_per_rank` consecutive documents -> [first, first + n)"""
    return rank * docs_per_rank, docs_per_rank


def split_docs(n_docs, world):
    """strong scaling helper: n_docs cut into `world` contiguous ranges of near-equal size -> list of (first, n)"""
    base, rem = divmod(n_docs, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((first, n))
        first += n
    return out


def split_docs_by_bytes(doc_off, world):
    """strong scaling over a real corpus: contiguous document ranges with near-equal TEXT BYTES rather
    than near-equal document counts, so that a few very long documents do not make one rank the straggler.
    doc_off = the batch's n_docs + 1 ascending offsets -> list of (first, n) per rank, covering every document once."""
    import numpy as np
    off = np.asarray(doc_off, dtype=np.uint64)
    n_docs = len(off) - 1
    if n_docs <= 0:
        return [(0, 0)] * world
    rel = (off - off[0]).astype(np.float64)
    total = rel[-1]
    # rank r starts at the first document whose start offset reaches r / world of the bytes
    cuts = [int(np.searchsorted(rel[:-1], total * r / world, side="left")) for r in range(world)] + [n_docs]
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


class BitmapGather:
    """rank 0 receives every rank's [docs, words] int32 bitmap; buffers are allocated once and reused per step.

    One local bitmap: `gather()` is the plain collective.  Several local bitmaps (slots): `start(slot)` launches the
    gather of that slot asynchronously and `wait(slot)` is called before the slot is written again, so the exchange of
    batch i travels over xGMI while batch i + 1 is scanned and solved (the collective runs on the process group's own
    stream); `drain()` completes everything that is still in flight."""

    def __init__(self, local_bitmap, rows_per_rank=None, force_collective=False):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # a group of ONE rank normally skips the exchange; force_collective sends it through the process group all the same
        # (dist.gather / all_gather of one rank over RCCL: how the collective path is exercised on a one-GPU box)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.locals = list(local_bitmap) if isinstance(local_bitmap, (list, tuple)) else [local_bitmap]
        self.local = self.locals[0]
        self.slot_parts = [None] * len(self.locals)
        self.pending = [None] * len(self.locals)
        if self.collective and self.rank == 0:
            rows = rows_per_rank or [self.local.shape[0]] * self.world
            self.slot_parts = [[torch.empty((r, self.local.shape[1]), dtype=self.local.dtype, device=self.local.device)
                                for r in rows] for _ in self.locals]
        self.parts = self.slot_parts[0]
        if self.collective and rows_per_rank is not None and len(set(rows_per_rank)) > 1:
            raise ValueError("dist.gather needs equal shard sizes; pad the last shard or use split sizes that divide")

    def __call__(self):
        if self.collective:
            dist.gather(self.local, self.parts, dst=0)
        return self.parts if self.collective else [self.local]

    def start(self, slot):
        if self.collective:
            self.pending[slot] = dist.gather(self.locals[slot], self.slot_parts[slot], dst=0, async_op=True)

    def wait(self, slot):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def drain(self):
        for s in range(len(self.locals)):
            self.wait(s)

    def full(self, slot=0):
        """rank 0: the bitmap of all documents in document order"""
        parts = self.slot_parts[slot] if self.collective else [self.locals[slot]]
        return torch.cat(parts, 0) if self.rank == 0 else None


def bitmap_checksum(t):
    """64-bit position-weighted checksum of an int32 bitmap (on the tensor's device; wraps like uint64 arithmetic):
    a word that changed, moved or went missing changes it"""
    w = t.reshape(-1).to(torch.int64) & 0xFFFFFFFF
    k = torch.arange(1, w.numel() + 1, dtype=torch.int64, device=t.device) * 0x9E3779B1 + 0x7F4A7C15
    return int(((w + 1) * k).sum().item()) & 0xFFFFFFFFFFFFFFFF


def verify_gather(gather, slot, rows):
    """Did every shard's bitmap arrive on rank 0 intact?  Every rank folds the first `rows` rows of its LOCAL bitmap of
    `slot` into a checksum on its device; one tiny all_gather brings the checksums (and row counts) together; rank 0
    computes the same checksums over what the gather delivered and compares.  -> (ok on rank 0 / True elsewhere, shards
    checked).  Outside any timed region: it synchronises."""
    if not gather.collective:
        return True, 1
    local = gather.locals[slot][:rows]
    dev = local.device
    c = bitmap_checksum(local)
    mine = torch.tensor([c - (1 << 64) if c >= (1 << 63) else c, rows], dtype=torch.int64, device=dev)   # (as a signed word)
    allv = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(gather.world)]
    dist.all_gather(allv, mine)
    ok = True
    if gather.rank == 0:
        for r, v in enumerate(allv):
            want, n = int(v[0].item()) & 0xFFFFFFFFFFFFFFFF, int(v[1].item())
            ok = ok and bitmap_checksum(gather.slot_parts[slot][r][:n]) == want
    return ok, gather.world


def all_ranks_ok(flag, device):
    """logical AND of a per-rank boolean"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
