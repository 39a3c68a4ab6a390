"""plinopt_amd -- MI355X (gfx950) drop-in for the hot path of jgdumas/plinopt:
the randomized multi-start search for minimum-cost straight-line programs.

Only what the path needs lives here: `csrc/` (HIP kernels + the C-ABI of
include/plinopt_hip.h, built into libplinopt_hip.so) and a thin host mirror of
the reference's restart-loop interface (search.py, dist.py)."""
from . import capi  # noqa: F401
from .search import CSEPlan, CSEChain, TrilPlan, TRIL_BASE_SEED, cmp_op_count_key, cob_search, cob_search_batch, chain_batch, kernel_search, cse_search_multi, kernel_search_multi, tril_search_multi  # noqa: F401
from .dist import shard_range, pack_key, allreduce_best, allreduce_tril_best  # noqa: F401
