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
    """63-bit word whose integer order is the (cmpOpCount, seed) order:
    [cost field 1 : 20 bits][cost field 2 : 20 bits][seed offset : 23 bits].
    Returns None when a field does not fit (the caller then reduces in two stages)."""
    if cost_mode == capi.COST_ADD_THEN_MUL:
        f1, f2 = adds, muls
    elif cost_mode == capi.COST_SUM:
        f1, f2 = adds + muls, 0
    else:
        f1, f2 = adds + muls, adds
    if f1 >= (1 << 20) or f2 >= (1 << 20) or not (0 <= seed_off < (1 << 23)):
        return None
    return (f1 << 43) | (f2 << 23) | seed_off


def unpack_seed_off(word):
    return word & ((1 << 23) - 1)


def cost_word(adds, muls, cost_mode=capi.COST_SUM_THEN_ADD):
    """cost only (no seed), for the two-stage reduction"""
    if cost_mode == capi.COST_ADD_THEN_MUL:
        return (adds << 31) | muls
    if cost_mode == capi.COST_SUM:
        return (adds + muls) << 31
    return ((adds + muls) << 31) | adds


def allreduce_best(local, seed0, cost_mode=capi.COST_SUM_THEN_ADD, group=None, device=None, fields=False):
    """local = (adds, muls, seed) or None.  Returns (winning seed, reduced word); with fields=True a third element (f1, f2): the two
    cost fields of the order (sum, adds / adds, muls / sum, 0 by cost mode) decoded for whichever of the two word layouts was reduced --
    callers that compare or print costs use these, never bits of `word`.
    ONE all_reduce(MIN) of a single int64 when (cost, seed offset) packs into 63 bits on every rank
    (always the case for the workloads of bench.py); otherwise two 8-byte MIN all-reduces
    (cost first, then the smallest seed among the ranks holding that cost)."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev = device if device is not None else "cpu"
    word = INF
    fits = 1
    if local is not None:
        adds, muls, seed = local
        w = pack_key(adds, muls, seed - seed0, cost_mode)
        if w is None:
            fits = 0
        else:
            word = w
    t = torch.tensor([word, fits], dtype=torch.int64, device=dev)      # MIN of fits == 1 iff EVERY rank's candidate packs into one word
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    if int(t[1].item()) == 1:
        w = int(t[0].item())
        if w == INF:
            return (None, w, None) if fields else (None, w)
        r = (seed0 + unpack_seed_off(w), w)
        return r + ((w >> 43, (w >> 23) & 0xFFFFF),) if fields else r
    # two-stage fallback: wide costs or far-apart seeds
    c = INF if local is None else cost_word(local[0], local[1], cost_mode)
    tc = torch.tensor([c], dtype=torch.int64, device=dev)
    if multi:
        dist.all_reduce(tc, op=dist.ReduceOp.MIN, group=group)
    best_c = int(tc.item())
    if best_c == INF:
        return (None, INF, None) if fields else (None, INF)
    sd = local[2] - seed0 if (local is not None and c == best_c) else INF
    ts = torch.tensor([sd], dtype=torch.int64, device=dev)
    if multi:
        dist.all_reduce(ts, op=dist.ReduceOp.MIN, group=group)
    r = (seed0 + int(ts.item()), best_c)
    return r + ((best_c >> 31, best_c & ((1 << 31) - 1)),) if fields else r


def allreduce_tril_best(local, seed0, group=None, device=None, fields=False):
    """Same single MIN all-reduce for the in-place trilinear search (`SearchTriLinearAlgorithm`'s critical sections,
    include/plinopt_inplace.inl:891-921): local = ((ADD, SCA, MUL), seed, variant) or None; order (ADD, SCA, seed, variant).
    Returns (seed, variant, word) or (None, None, INF)."""
    pseudo = None
    if local is not None:
        (add, sca, _mul), seed, variant = local
        pseudo = (add, sca, seed0 + (((seed - seed0) << 1) | variant))
    s, w, fl = allreduce_best(pseudo, seed0, capi.COST_ADD_THEN_MUL, group=group, device=device, fields=True)
    if s is None:
        return (None, None, INF, None) if fields else (None, None, INF)
    off = s - seed0
    r = (seed0 + (off >> 1), off & 1, w)
    return r + (fl,) if fields else r                     # fl = (ADD, SCA) of the winner


def allreduce_cob_best(local, n, group=None, device=None):
    """Change-of-basis enumeration sharded by (i,j,k) prefix (plo_cob_search_range): local = (zeros_v, zeros_w, index, found).
    `localSparsifier` keeps the FIRST candidate of best (zeros_v, zeros_w) in loop order (include/plinopt_sparsify.inl:179-194,
    :299-314), i.e. the largest score with the smallest index: one 8-byte MAX all-reduce of (score << 32 | ~index).
    Returns (zeros_v, zeros_w, index, found) of the whole enumeration."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev = device if device is not None else "cpu"
    word = 0
    if local is not None and local[3]:
        zv, zw, idx, _ = local
        word = ((zv * (n + 1) + zw + 1) << 32) | ((~idx) & 0xFFFFFFFF)
    t = torch.tensor([word], dtype=torch.int64, device=dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    w = int(t.item())
    if w == 0:
        return None
    sc = (w >> 32) - 1
    return sc // (n + 1), sc % (n + 1), (~w) & 0xFFFFFFFF, 1
