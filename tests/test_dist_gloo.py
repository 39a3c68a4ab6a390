"""N>1 path on CPU: two gloo ranks shard the seed range, each reduces its shard to a packed
(cost, seed) word and ONE MIN all-reduce yields the same winner as the unsharded search.
The per-shard search is done by the CPU oracle here (stand-in for the device; test-only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from plinopt_amd.dist import shard_range, allreduce_best
from plo_testlib import DATA, OracleMatrix
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), 131071)
seed0, n = 777, 3001
for mode in (0, 1, 2):
    s, cnt = shard_range(seed0, n, rank, world)
    local = M.search(s, cnt, cost_mode=mode) if cnt else None
    seed, word = allreduce_best(local, seed0, mode)
    exp = M.search(seed0, n, cost_mode=mode)
    assert seed == exp[2], (mode, seed, exp)
# costs / seed offsets too wide for one word: two-stage reduction gives the same winner
big0 = 10 ** 12
s, cnt = shard_range(big0, 40, rank, world)
local = M.search(s, cnt)
seed, word = allreduce_best((local[0] + (1 << 20), local[1], local[2]), 0, 0)
exp = M.search(big0, 40)
assert seed == exp[2], (seed, exp)
# exactly ONE rank does not fit the one-word form (its seed offset is beyond 2^23): the other rank's word must not win by
# default -- rank 0 holds (100|0, seed 1), rank 1 the better (50|0, seed 2^23+5)
local = (100, 0, 1) if rank == 0 else (50, 0, (1 << 23) + 5)
seed, word = allreduce_best(local, 0, 0)
assert seed == (1 << 23) + 5, seed
local = (50, 0, 1) if rank == 0 else (100, 0, (1 << 23) + 5)
seed, word = allreduce_best(local, 0, 0)
assert seed == 1, seed
# a rank with an empty shard must not disturb the reduction
s, cnt = shard_range(5, 1, rank, world)
local = M.search(s, cnt) if cnt else None
seed, word = allreduce_best(local, 5, 0)
assert seed == 5
# the trilplacer restart loop shards the same way: order (ADD, SCA, seed, variant)
from plinopt_amd.dist import allreduce_tril_best
from plo_testlib import OracleTril
T = OracleTril.from_sms(*(os.path.join(DATA, "4x4x4_49_156" + x) for x in ("_L.sms", "_R.sms", "_P.sms")))
s, cnt = shard_range(40, 151, rank, world)
seed, variant, word = allreduce_tril_best(T.search(s, cnt) if cnt else None, 40)
exp = T.search(40, 151)
assert (seed, variant) == (exp[1], exp[2]), (seed, variant, exp)
# the change-of-basis enumeration shards by (i,j,k) prefix: best score wins, the smaller index among equal scores
from plinopt_amd.dist import allreduce_cob_best
got = allreduce_cob_best((30, 2, 5000 + rank, 1), 16)
assert got == (30, 2, 5000, 1), got
got = allreduce_cob_best((30, 2 + rank, 7, 1), 16)
assert got == (30, 1 + world, 7, 1), got
got = allreduce_cob_best((29, 15, 1, 1) if rank == 0 else None, 16)
assert got == (29, 15, 1, 1), got
assert allreduce_cob_best((0, 0, 0, 0), 16) is None
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run_world(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1" if world > 2 else "2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == world


def test_two_rank_seed_shard_min_allreduce(tmp_path):
    _run_world(tmp_path, 2, 29533)


def test_eight_rank_seed_shard_min_allreduce(tmp_path):
    """the world size of the 8-GPU configs of BASELINE.json (configs[3], configs[4]): eight gloo ranks, the same single MIN
    (MAX for the change-of-basis enumeration) all-reduce, empty shards included (a 1-seed range over 8 ranks)"""
    _run_world(tmp_path, 8, 29541)


def test_shard_range_partitions_exactly():
    from plinopt_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            got = [shard_range(100, n, r, world) for r in range(world)]
            assert sum(c for _, c in got) == n
            pos = 100
            for s, c in got:
                assert s == pos
                pos += c


def test_pack_key_order():
    from plinopt_amd.dist import pack_key
    assert pack_key(10, 2, 5) < pack_key(11, 1, 0) < pack_key(9, 4, 0)
    assert pack_key(10, 2, 5) < pack_key(10, 2, 6)
    assert pack_key(3, 100, 0, 1) < pack_key(4, 0, 0, 1)
    assert pack_key(0, 0, 0) >= 0 and pack_key(34409, 4546, 2 ** 23 - 1) < 2 ** 63     # config-5 sized costs fit
    assert pack_key(2 ** 20, 0, 0) is None and pack_key(1, 1, 2 ** 23) is None          # -> two-stage reduction
