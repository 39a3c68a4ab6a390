"""The HBM-resident kernel family (plo_cse_big.hip, one workgroup per candidate) against the CPU
oracle on inputs the oracle finishes in seconds: the same per-seed (adds, muls) and the same argmin as
the LDS-resident wave kernel, forced with PLO_PLAN_HBM."""
import glob
import os

import pytest

import synth
from plo_testlib import DATA, OracleMatrix

pytestmark = pytest.mark.gpu
P = 131071
SOME = ["2x2x2_7_Winograd_L.sms", "cyclic.sms", "4x4x4_49_156_L.sms", "4x4x4_49_156_P.sms", "3x3x3_23_58_R.sms",
        "2x2x2_7_DPS-accurate_L.sms", "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms", "3x3x6_40_DPS-accurate_P.sms",
        "2o2o4_5_Toom3_P.sms", "4x4x4_48_accurate-CoB_R.sms", "3x4x7_63_rational_R.sms", "4o4o8_Toom5_P.sms"]


def _plan(M):
    from plinopt_amd import CSEPlan
    pl = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p, hbm=True)
    assert pl.is_hbm
    return pl


@pytest.mark.parametrize("name", SOME)
def test_hbm_variant_matches_oracle(hip, name):
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    plan = _plan(M)
    n = 200
    assert plan.cost_many(seed0=77, n=n) == tuple(M.cost_many(seed0=77, nseeds=n, nthreads=8))


def test_hbm_variant_search_argmin(hip):
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    plan = _plan(M)
    for mode in (0, 1, 2):
        assert plan.search(5, 1500, cost_mode=mode) == M.search(5, 1500, cost_mode=mode, nthreads=8)


def test_hbm_variant_synthetic(hip):
    from plinopt_amd import CSEPlan
    for s in range(60):
        m, n, rows = synth.small_valued(s, P)
        rp, c, v = synth.to_csr(rows, P)
        M = OracleMatrix(m, n, rp, c, v, P)
        plan = CSEPlan(m, n, rp, c, v, P, hbm=True)
        assert plan.cost_many(seed0=s, n=16) == tuple(M.cost_many(seed0=s, nseeds=16)), (s, rows)
        plan.close()


def test_rows_longer_than_the_wave_kernel_limit_select_hbm_automatically(hip):
    """A 100-entry row does not fit the wave kernel (64 lanes per row): plo_cse_plan_create picks the
    HBM family by itself; results still equal the oracle."""
    import random
    from plinopt_amd import CSEPlan
    rng = random.Random(3)
    rows = [{j: rng.choice([1, P - 1, 2, P - 2]) for j in range(100) if rng.random() < 0.8} for _ in range(12)]
    rp, c, v = synth.to_csr(rows, P)
    M = OracleMatrix(12, 100, rp, c, v, P)
    plan = CSEPlan(12, 100, rp, c, v, P)
    assert plan.is_hbm
    assert plan.cost_many(seed0=0, n=24) == tuple(M.cost_many(seed0=0, nseeds=24, nthreads=8))


@pytest.mark.parametrize("shape,unit_frac,seed", [((220, 48), 1.0, 1), ((180, 64), 0.7, 2), ((64, 96), 0.5, 3)])
def test_hbm_variant_mid_size_random(hip, shape, unit_frac, seed):
    """Matrices too large for the wave kernel's limits in at least one dimension (rows > 64 entries or op-counts), a few
    thousand non-zeros, coefficients from {+-1, +-2, +-1/2}: the oracle still finishes in seconds."""
    from plinopt_amd import CSEPlan
    m, n = shape
    mm, nn, rows = synth.sweep(900 + seed, P, m, n, density=0.3, unit_frac=unit_frac)
    rp, c, v = synth.to_csr(rows, P)
    M = OracleMatrix(mm, nn, rp, c, v, P)
    plan = CSEPlan(mm, nn, rp, c, v, P, hbm=True)
    ncand = 6
    assert plan.cost_many(seed0=40, n=ncand) == tuple(M.cost_many(seed0=40, nseeds=ncand, nthreads=8))


def test_general_matrix_with_many_rows_goes_to_hbm_family(hip):
    """> 64 rows with rational coefficients: the wave kernel's general ProgramGen keeps one row per lane, so plan creation
    selects the HBM family by itself (never a run-time 'unsupported')."""
    from plinopt_amd import CSEPlan
    mm, nn, rows = synth.sweep(4242, P, 100, 24, density=0.3, unit_frac=0.6)
    rp, c, v = synth.to_csr(rows, P)
    M = OracleMatrix(mm, nn, rp, c, v, P)
    plan = CSEPlan(mm, nn, rp, c, v, P)
    assert plan.is_hbm
    assert plan.cost_many(seed0=3, n=12) == tuple(M.cost_many(seed0=3, nseeds=12, nthreads=8))
    assert plan.search(3, 12) == M.search(3, 12, nthreads=8)


@pytest.mark.parametrize("name", ["4x4x4_49_156_P.sms", "4x4x4_48_rational_P.sms", "3x4x7_63_rational_R.sms"])
def test_tiny_aggregation_table_takes_the_spill_path(hip, name, monkeypatch):
    """A 64-entry LDS aggregation table overflows at almost every step: entries then retire in HBM directly and their
    pair with the new column waits in the spill list for the flush's second pass.  Same costs as the oracle."""
    monkeypatch.setenv("PLO_BIG_AGGBITS", "6")
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    plan = _plan(M)
    assert plan.cost_many(seed0=11, n=150) == tuple(M.cost_many(seed0=11, nseeds=150, nthreads=8))


def _soak_case(s):
    """random matrix number s of tests/soak_hbm.py"""
    import random
    rng = random.Random(7000 + s)
    m, n = rng.randint(150, 400), rng.randint(150, 320)
    dens = rng.choice([0.2, 0.35, 0.5])
    vals = [1, P - 1] + [rng.randint(2, P - 2) for _ in range(rng.randint(0, 3))]
    rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < dens} for _ in range(m)]
    return m, n, [r if r else {0: 1} for r in rows]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case,refits", [(5, 1), (164, 2), (1, 0)])
def test_candidates_that_outgrow_the_planned_structures(hip, tmp_path, case, refits):
    """The structures of the deferred updates (and the eager pair table behind them) are sized from the INPUT's triples.  On random dense
    matrices with few distinct values the live triples of a candidate outgrow that (found by tests/soak_hbm.py: a fatal "pair table"
    error, and with the eager table an insertion that probed 2^22 slots per key -- the launch looked hung).  The device reports it, the plan
    is rebuilt (eager table, then four times its slots) and the launch repeated: costs equal the build's scalable host engine
    (`bin/optimizer --replay`, which prints the oracle's text wherever the oracle can walk), seed by seed."""
    import re
    import subprocess
    from plo_testlib import ROOT
    from plinopt_amd import CSEPlan
    m, n, rows = _soak_case(case)
    rp, c, v = synth.to_csr(rows, P)
    plan = CSEPlan(m, n, rp, c, v, P, hbm=True)
    got = plan.cost_many(seed0=case * 10, n=3)
    assert plan.hbm_counters()["eager_refits"] == refits
    plan.close()
    path = tmp_path / "m.sms"
    with open(path, "w") as f:
        f.write("%d %d M\n" % (m, n))
        for i, r in enumerate(rows):
            for j in sorted(r):
                f.write("%d %d %d\n" % (i + 1, j + 1, r[j] if r[j] <= P // 2 else r[j] - P))
        f.write("0 0 0\n")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    for k in range(3):
        r = subprocess.run([os.path.join(ROOT, "bin", "optimizer"), "-q", str(P), "--gpu", "0", "--replay", "--seed", str(case * 10 + k), str(path)], capture_output=True, text=True, timeout=120)
        a, mu = int(re.search(r"# (\d+)\tadditions", r.stderr).group(1)), int(re.search(r"# (\d+)\tmultiplications", r.stderr).group(1))
        assert (got[0][k], got[1][k]) == (a, mu), (case, k)

