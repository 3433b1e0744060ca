"""Multi-GPU plumbing of the read-sharded path: one process per GPU, reads dealt out in
consecutive batches, the packed device index replicated by ONE broadcast, SAM merged on the host in batch
order.  Nothing here touches the GPU directly, so the same code runs under gloo on CPU in the tests and
under nccl (= RCCL over xGMI) on a node.

Mirrors what the reference does with pthreads inside one batch (alnse.c:1419-1439): the work unit is the
batch of N_SEQS reads (aln.h:27) instead of the read, and the ordered `puts` loop becomes merge_ordered().
"""
import numpy as np

N_SEQS = 100000          # aln.h:27


def batch_bounds(n_reads, batch=N_SEQS):
    """[(first, last+1)] of the consecutive batches the reference's reader would form."""
    return [(b, min(b + batch, n_reads)) for b in range(0, n_reads, batch)]


def my_batches(n_reads, rank, world, batch=N_SEQS):
    """Batches of this rank: round-robin by batch sequence number."""
    return [(i, lo, hi) for i, (lo, hi) in enumerate(batch_bounds(n_reads, batch)) if i % world == rank]


def broadcast_bytes(buf, src=0, group=None):
    """Broadcast a uint8 tensor (the packed index image) from `src` to every rank; returns the tensor.
    Non-source ranks pass None and get a freshly allocated tensor on the device of `like`."""
    import torch
    import torch.distributed as dist
    dev = buf.device if buf is not None else None
    n = torch.tensor([buf.numel() if buf is not None else 0], dtype=torch.int64)
    backend = dist.get_backend(group)
    if backend == "nccl":
        n = n.cuda()
    dist.broadcast(n, src, group=group)
    if buf is None:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=n.device if backend == "nccl" else "cpu")
    dist.broadcast(buf, src, group=group)
    return buf


def gather_ordered(local_parts, n_batches, dst=0, group=None):
    """local_parts: {batch_no: bytes}.  Returns on `dst` the list of all parts in batch order (the host-side
    SAM merge), None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    gathered = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(local_parts, gathered, dst=dst, group=group)
    if gathered is None:
        return None
    merged = {}
    for d in gathered:
        merged.update(d)
    missing = [i for i in range(n_batches) if i not in merged]
    if missing:
        raise RuntimeError("batches missing from the merge: %s" % missing[:8])
    return [merged[i] for i in range(n_batches)]


def align_sharded(align_fn, format_fn, seqs, offs, rank, world, batch=N_SEQS):
    """Runs align_fn(seqs_slice, offs_slice) -> results on this rank's batches and format_fn(lo, hi, results)
    -> bytes on each; returns {batch_no: bytes}."""
    n = len(offs) - 1
    out = {}
    for i, lo, hi in my_batches(n, rank, world, batch):
        o = (offs[lo:hi + 1] - offs[lo]).astype(np.uint32)
        s = seqs[int(offs[lo]):int(offs[hi])]
        out[i] = format_fn(lo, hi, align_fn(s, o))
    return out
