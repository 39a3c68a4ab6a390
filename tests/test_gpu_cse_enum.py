"""-E on the GPU: the schedules of RecSub's tree (reference include/plinopt_optimize.inl:889-1013) addressed by index,
bit-exact (adds, muls, radix product) against the oracle, and exhaustive search on inputs whose tree is small."""
import os

import pytest

from plo_testlib import DATA, OracleMatrix

pytestmark = pytest.mark.gpu
P = 131071


def _plan(M):
    from plinopt_amd import CSEPlan
    return CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p)


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "2x2x2_7_Winograd_P.sms", "cyclic.sms", "3x3x3_23_58_L.sms", "4x4x4_49_156_L.sms",
                                  "2x2x2_7_DPS-accurate_L.sms", "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms"])
def test_schedules_match_the_oracle(hip, name):
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    plan = _plan(M)
    for first, n in ((0, 700), (10 ** 12 + 12345, 300)):
        assert plan.enum_cost_many(first, n) == tuple(M.enum_cost_many(first, n, nthreads=8))


def test_exhaustive_search_on_winograd(hip):
    """The whole tree of Winograd's L has 3 leaves... of its P a few more: the enumeration is complete once count >= the
    largest radix product, and the best schedule reaches the known optimum (4 additions for L)."""
    from plinopt_amd import capi
    for name, best_adds in (("2x2x2_7_Winograd_L.sms", 4), ("2x2x2_7_Winograd_R.sms", 4), ("2x2x2_7_Winograd_P.sms", 7)):
        M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
        plan = _plan(M)
        n = 1
        while True:
            (a, mu, idx), maxprod = plan.enum_search(0, n)
            if maxprod <= n:
                break
            n = maxprod
        assert n < 10 ** 6
        oa, om, op = M.enum_cost_many(0, n, nthreads=8)
        assert max(op) == maxprod and (a, mu, idx) == min((x, y, k) for k, (x, y) in enumerate(zip(oa, om)))
        assert a == best_adds and mu == 0
        assert plan.enum_search(0, n, capi.COST_SUM_THEN_ADD)[0][:2] == (a, mu)


def test_enumeration_is_refused_for_hbm_plans(hip):
    from plinopt_amd import CSEPlan, capi
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p, hbm=True)
    with pytest.raises(capi.PloError) as e:
        plan.enum_search(0, 10)
    assert e.value.code == capi.PLO_E_UNSUPPORTED


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms", "2x2x2_7_DPS-accurate_P.sms", "2x2x2_7_DPS-accurate_R.sms"])
def test_recsub_accounting_on_gpu_equals_literal_recsub(hip, name):
    """PLO_COST_RECSUB: the enumeration chooses by (additions, multiplications before ProgramGen) = RecSub's own counts
    (reference include/plinopt_optimize.inl:950-959); the minimum over the whole tree equals the literal recursive
    restatement of the oracle."""
    import ctypes
    from plinopt_amd import CSEPlan, capi
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    adds, muls_r, _, _ = M.recsub()
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, P)
    L = capi.lib()
    n, best = 1, None
    while True:
        b, st, mp = capi.Best(), capi.Stats(), ctypes.c_uint64()
        capi.check(L.plo_cse_enum_search_plan(plan._h, 0, n, capi.COST_RECSUB, ctypes.byref(b), ctypes.byref(mp), ctypes.byref(st)))
        best = (b.adds, b.muls)
        if mp.value <= n:
            break
        n = mp.value
    assert best == (adds, muls_r), (best, adds, muls_r, n)
