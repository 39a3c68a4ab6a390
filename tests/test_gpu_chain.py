"""Chained candidates (LU method): Optimizer() on a first matrix, then on a second one with the same random
stream running on, costs added -- HIP kernel vs CPU oracle."""
import os

import pytest

from plo_testlib import DATA, OracleMatrix, oracle_chain

pytestmark = pytest.mark.gpu
P = 131071
PAIRS = [("2x2x2_7_Winograd_L.sms", "2x2x2_7_Winograd_P.sms"), ("4x4x4_49_156_L.sms", "3x3x3_23_58_P.sms"),
         ("2x2x2_7_DPS-accurate_L.sms", "4x4x4_49_156_R.sms"), ("3o3o6_Toom4_P.sms", "2o2o4_5_Toom3_P.sms"),
         ("cyclic.sms", "4x4x4_48_rational_P.sms")]


@pytest.mark.parametrize("na,nb", PAIRS)
def test_chain_costs_and_argmin(hip, na, nb):
    from plinopt_amd import CSEChain
    A = OracleMatrix.from_sms(os.path.join(DATA, na), P)
    B = OracleMatrix.from_sms(os.path.join(DATA, nb), P)
    ch = CSEChain((A.m, A.n, A.rowptr, A.col, A.val), (B.m, B.n, B.rowptr, B.col, B.val), P)
    n = 300
    ga, gm = ch.cost_many(seed0=11, n=n)
    exp = [oracle_chain(A, B, 11 + k) for k in range(n)]
    assert list(zip(ga, gm)) == exp
    best = min(range(n), key=lambda k: (exp[k][0] + exp[k][1], exp[k][0], k))
    assert ch.search(11, n) == (exp[best][0], exp[best][1], 11 + best)


def test_many_pairs_in_one_launch(hip):
    """plo_cse_chain_batch: candidate c runs on pair c // per_pair with seed seed0 + c (the kernel method's shape: a new
    pair of matrices every few restarts).  Same costs as the oracle chain per candidate, same argmin."""
    from plinopt_amd import chain_batch
    mats = {}
    for na, nb in PAIRS:
        for x in (na, nb):
            mats[x] = OracleMatrix.from_sms(os.path.join(DATA, x), P)
    order = PAIRS * 3 + [PAIRS[1], PAIRS[0]]                      # 17 pairs of different sizes, unit and general paths mixed
    pairs = [((mats[a].m, mats[a].n, mats[a].rowptr, mats[a].col, mats[a].val), (mats[b].m, mats[b].n, mats[b].rowptr, mats[b].col, mats[b].val)) for a, b in order]
    per, seed0 = 23, 1000
    ga, gm, best, st = chain_batch(pairs, P, seed0, per)
    exp = [oracle_chain(mats[order[c // per][0]], mats[order[c // per][1]], seed0 + c) for c in range(len(order) * per)]
    assert list(zip(ga, gm)) == exp
    b = min(range(len(exp)), key=lambda k: (exp[k][0] + exp[k][1], exp[k][0], k))
    assert best == (exp[b][0], exp[b][1], seed0 + b)
    assert st["launches"] == 1
