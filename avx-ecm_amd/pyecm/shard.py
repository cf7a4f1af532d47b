"""Multi-GPU sharding of the curve batch.

Curves are independent (same N, same op tape, different sigma), so the batch is split on the host:
rank g of G owns the contiguous global curve indices [g*B/G, (g+1)*B/G), i.e. sigma0 + that range
(the reference's sigma_k = sigma0 + k at threads=1: main.c:761, ecm.c:1187).  There is no data-path
collective.  The ONE collective is the "factor found" reduction that replaces the reference's
sequential scan + `if (found) break` (ecm.c:1323-1370, 1485-1532): a max-reduce over a small integer
record, RCCL (backend "nccl") on GPUs, gloo in the CPU tests.  Save-file lines are gathered to rank 0
in global curve order so save_b1.txt is identical for any G.
"""


def shard_bounds(total, rank, world):
    """[lo, hi) of global curve indices owned by `rank`; contiguous, sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi


def shard_sigmas(sigma0, total, rank, world):
    lo, hi = shard_bounds(total, rank, world)
    return [sigma0 + k for k in range(lo, hi)]


NONE = -1


def encode_found(global_curve_or_none, total):
    """Record whose MAX over ranks selects the LOWEST global curve index that found a factor
    (the one the reference's scan would report first): value = total - index, 0 = nothing found."""
    return 0 if global_curve_or_none is None else total - int(global_curve_or_none)


def decode_found(value, total):
    return None if value == 0 else total - int(value)


def allreduce_found(dist, local_found_global_index, total, device="cpu"):
    """One all-reduce (MAX) of the found record.  Returns the lowest global curve index with a
    factor over all ranks, or None."""
    import torch
    t = torch.tensor([encode_found(local_found_global_index, total)], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():
        # also for a world of one: a single rank launched through torch.distributed.run runs the very collective the
        # N-rank job runs (RCCL on a device tensor), so the path is exercised on whatever hardware there is
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return decode_found(int(t.item()), total)


def gather_lines(dist, lines):
    """Gather each rank's save lines to rank 0, concatenated in rank (= global curve) order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(lines)
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(list(lines), out, dst=0)
    if dist.get_rank() != 0:
        return None
    return [l for part in out for l in part]
