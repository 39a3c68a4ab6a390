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
