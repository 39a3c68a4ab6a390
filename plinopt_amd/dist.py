"""Seed-space sharding across ranks (one process per GPU) and the single MIN
all-reduce that replaces the reference's `#pragma omp critical` best-so-far
(include/plinopt_optimize.inl:1214-1237).  Candidates are independent, so there
is no data-path collective: only the 8-byte (cost, seed) word is reduced
(RCCL over xGMI with backend "nccl"; gloo in the CPU tests)."""
from . import capi

INF = (1 << 63) - 1


def shard_range(seed0, nseeds, rank, world):
    """Contiguous block of the seed range owned by `rank` (first ranks take the remainder)."""
    q, r = divmod(nseeds, world)
    start = seed0 + rank * q + min(rank, r)
    return start, q + (1 if rank < r else 0)


def pack_key(adds, muls, seed_off, cost_mode=capi.COST_SUM_THEN_ADD):
    """63-bit word, integer order == (cmpOpCount, seed) order: cost in the high 31 bits."""
    if cost_mode == capi.COST_ADD_THEN_MUL:
        hi = (adds << 15) | muls
        assert adds < (1 << 16) and muls < (1 << 15)
    elif cost_mode == capi.COST_SUM:
        hi = (adds + muls) << 15
    else:
        hi = ((adds + muls) << 15) | adds
        assert adds < (1 << 15)
    assert hi < (1 << 31) and 0 <= seed_off < (1 << 32)
    return (hi << 32) | seed_off


def unpack_seed_off(word):
    return word & 0xFFFFFFFF


def allreduce_best(local, seed0, cost_mode=capi.COST_SUM_THEN_ADD, group=None, device=None):
    """local = (adds, muls, seed) or None.  Returns the global winner's seed and packed word.
    One all_reduce(MIN) of a single int64."""
    import torch
    import torch.distributed as dist
    word = INF
    if local is not None:
        adds, muls, seed = local
        word = pack_key(adds, muls, seed - seed0, cost_mode)
    t = torch.tensor([word], dtype=torch.int64, device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    w = int(t.item())
    if w == INF:
        return None, w
    return seed0 + unpack_seed_off(w), w
