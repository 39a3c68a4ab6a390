"""Parity of the HIP wave kernel (through the C-ABI) with the CPU oracle:
bit-exact (adds, muls) per seed, identical argmin under (cmpOpCount, seed)."""
import os

import pytest

from plo_testlib import DATA, OracleMatrix

pytestmark = pytest.mark.gpu

P = 131071

UNIT_FILES = [
    "2x2x2_7_Winograd_L.sms", "2x2x2_7_Winograd_P.sms", "2x2x2_7_Strassen_R.sms", "cyclic.sms",
    "3x3x3_23_58_L.sms", "3x3x3_23_58_P.sms", "3x3x3_23_Grey-221_P.sms",
    "4x4x4_49_156_L.sms", "4x4x4_49_156_R.sms", "4x4x4_49_156_P.sms",
    "3x4x7_63_rational_L.sms", "3x3x6_40_L.sms", "6x3x3_40_DPS-accurate_R.sms", "4x4T_8+26_RXTX_P.sms",
    "4o4o4_F32_Standard_L.sms", "3x4x7_63_rational-ALT_L.sms",
]


def _plan(M):
    from plinopt_amd import CSEPlan
    return CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p)


@pytest.mark.parametrize("name", UNIT_FILES)
def test_cost_many_matches_oracle(hip, name):
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    nseeds = 3000 if len(M.col) < 100 else 1200
    plan = _plan(M)
    ga, gm = plan.cost_many(seed0=0, n=nseeds)
    oa, om = M.cost_many(seed0=0, nseeds=nseeds, nthreads=8)
    assert ga == oa
    assert gm == om
    # explicit, scattered 64-bit seeds
    seeds = [(1 << 40) + 7919 * k * k for k in range(257)] + [2 ** 64 - 1, 2 ** 63, 0]
    ga, gm = plan.cost_many(seeds=seeds)
    oa, om = M.cost_many(seeds=seeds)
    assert (ga, gm) == (oa, om)


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "3x3x3_23_58_P.sms"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_search_argmin_matches_oracle(hip, name, mode):
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    plan = _plan(M)
    n = 5000
    got = plan.search(12345, n, cost_mode=mode)
    exp = M.search(12345, n, cost_mode=mode, nthreads=8)
    assert got == exp


def test_other_modulus(hip):
    for p in (7, 65537, 2147483629):
        M = OracleMatrix.from_sms(os.path.join(DATA, "3x3x3_23_58_L.sms"), p)
        plan = _plan(M)
        assert plan.cost_many(seed0=5, n=500) == tuple(M.cost_many(seed0=5, nseeds=500, nthreads=8))


@pytest.mark.parametrize("name", ["4x4x4_49_156_P.sms", "4x4x4_49_156_L.sms", "3x3x6_40_P.sms", "4x4x4_48_rational_P.sms"])
def test_largest_31_bit_prime_on_wide_matrices(hip, name):
    """Round 3: 51-bit pair keys (13 value bits per slot instead of 20): a 31-bit ratio leaves 10 bits per column, so the largest
    prime below 2^31 works on every LDS-sized matrix of the data set -- 4x4x4_49_156_P (49 columns, up to 141 with the created
    ones) was PLO_E_CAPACITY with the 44-bit key of round 2 (reference field: Modular<Integer>, src/optimizer.cpp:125-139)."""
    p = 2147483629
    M = OracleMatrix.from_sms(os.path.join(DATA, name), p)
    plan = _plan(M)
    assert not plan.is_hbm
    assert plan.cost_many(seed0=9, n=400) == tuple(M.cost_many(seed0=9, nseeds=400, nthreads=8))
    assert plan.search(9, 400) == M.search(9, 400, nthreads=8)


def _wide(p, vals, seed=1):
    import random
    rng = random.Random(seed)
    rows = [sorted(rng.sample(range(1100), 6)) for _ in range(40)]
    rp, c, v = [0], [], []
    for r in rows:
        for j in r:
            c.append(j); v.append(rng.choice(vals))
        rp.append(len(c))
    return 40, 1100, rp, c, v


def test_capacity_error_is_loud(hip):
    """A pair key beyond 51 bits (more than 1024 columns with a 31-bit modulus) is refused by the LDS kernel; the HBM family takes it with
    ratio identifiers in its 48-bit key when the matrix has at most 32 distinct coefficients (next test) and refuses it otherwise:
    the plan fails loudly, nothing is rerouted silently."""
    import random
    from plinopt_amd import CSEPlan, capi
    p = 2147483629
    rng = random.Random(2)
    vals = [1, p - 1] + [rng.randint(2, p - 2) for _ in range(40)]
    m, n, rp, c, v = _wide(p, vals)
    assert len(set(v)) > 32
    with pytest.raises(capi.PloError) as e:
        CSEPlan(m, n, rp, c, v, p)
    assert e.value.code == capi.PLO_E_CAPACITY and "ratio identifiers" in str(e.value)


@pytest.mark.parametrize("p", [2147483629, 2147483647, 1073741827])
def test_31_bit_prime_with_1100_columns_on_the_hbm_family(hip, p):
    """Round 4: a modulus too wide for a residue in the HBM family's 48-bit pair key (31 bits leave 8 per column) gets the ratio's
    IDENTIFIER there (plo::cse_big_kernel<2, ., true>; at most 32 distinct coefficients, i.e. every +-1 matrix and the few-valued
    kind) -- PLO_E_CAPACITY in round 3 (reference field: Modular<Integer>, src/optimizer.cpp:125-139).  Against the literal oracle."""
    from plinopt_amd import CSEPlan
    m, n, rp, c, v = _wide(p, [1, p - 1, 2, p - 3, 5])
    M = OracleMatrix(m, n, rp, c, v, p)
    plan = CSEPlan(m, n, rp, c, v, p)
    assert plan.is_hbm
    assert plan.cost_many(seed0=3, n=12) == tuple(M.cost_many(seed0=3, nseeds=12, nthreads=8))
    assert plan.search(3, 12) == M.search(3, 12, nthreads=8)
    plan.close()


@pytest.mark.parametrize("name,nseeds", [("2x2x2_7_Winograd_L.sms", 100000), ("cyclic.sms", 30000), ("4x4x4_49_156_L.sms", 12000),
                                         ("4x4x4_49_156_R.sms", 12000), ("4x4x4_49_156_P.sms", 10000)])
def test_ten_thousand_seeds_per_baseline_config(hip, name, nseeds):
    """SURVEY 8c parity definition (A): all cost components equal for >= 10^4 seeds on every BASELINE config
    that the oracle can walk; (B): the same argmin over the whole range."""
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    plan = _plan(M)
    ga, gm = plan.cost_many(seed0=10 ** 9, n=nseeds)
    oa, om = M.cost_many(seed0=10 ** 9, nseeds=nseeds, nthreads=16)
    assert ga == oa and gm == om
    assert plan.search(10 ** 9, nseeds) == M.search(10 ** 9, nseeds, nthreads=16)
