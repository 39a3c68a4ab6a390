"""Synthetic sweeps: ProgramGen corner paths (FactorOutColumns/Rows, Triangle and its
never-reset `found` flag), ragged/empty rows, single-row and single-column inputs."""
import pytest

import synth
from plo_testlib import OracleMatrix

pytestmark = pytest.mark.gpu
P = 131071


def _both(m, n, rows, p, nseeds, seed0=0):
    from plinopt_amd import CSEPlan
    rp, c, v = synth.to_csr(rows, p)
    M = OracleMatrix(m, n, rp, c, v, p)
    plan = CSEPlan(m, n, rp, c, v, p)
    got = plan.cost_many(seed0=seed0, n=nseeds)
    exp = tuple(M.cost_many(seed0=seed0, nseeds=nseeds, nthreads=4))
    plan.close()
    return got, exp


def test_triangle_micro(hip):
    # <ab|b ; a|.> : rows [2 .] and [6 3]; one triangle saves one multiplication (2 instead of 3)
    got, exp = _both(2, 2, [{0: 2}, {0: 6, 1: 3}], P, 4)
    assert got == exp
    assert got[1][0] == 2 and got[0][0] == 1


@pytest.mark.parametrize("block", range(8))
def test_small_valued_random(hip, block):
    for s in range(block * 40, block * 40 + 40):
        m, n, rows = synth.small_valued(s, P)
        got, exp = _both(m, n, rows, P, 24, seed0=s)
        assert got == exp, (s, rows)


@pytest.mark.parametrize("shape", [(16, 8), (32, 16), (64, 32), (16, 32), (64, 8)])
def test_survey_sweep_shapes(hip, shape):
    m, n = shape
    for s in range(6):
        mm, nn, rows = synth.sweep(12345 + s, P, m, n)
        got, exp = _both(mm, nn, rows, P, 64)
        assert got == exp, (shape, s)


def test_edge_shapes(hip):
    cases = [
        (1, 1, [{0: 5}]),                                   # single entry
        (1, 4, [{0: 1, 1: P - 1, 2: 2, 3: 2}]),             # single row, repeated coefficient
        (4, 1, [{0: 3}, {0: 3}, {0: P - 3}, {0: 1}]),       # single column, repeated coefficient
        (3, 3, [{}, {0: 1, 2: 1}, {}]),                     # empty rows
        (2, 3, [{0: 1, 1: 1, 2: 1}, {0: 1, 1: 1, 2: 1}]),   # identical rows
        (3, 2, [{}, {}, {}]),                               # all-zero matrix
    ]
    for m, n, rows in cases:
        got, exp = _both(m, n, rows, P, 8)
        assert got == exp, rows


def test_wide_rows_and_many_rows(hip):
    # 64-entry rows (the row-length limit of the wave kernel) and > 64 rows (two mask words)
    import random
    rng = random.Random(7)
    rows = [{j: rng.choice([1, P - 1]) for j in range(64) if rng.random() < 0.9} for _ in range(6)]
    got, exp = _both(6, 64, rows, P, 16)
    assert got == exp
    rows = [{j: rng.choice([1, P - 1]) for j in range(6) if rng.random() < 0.5} for _ in range(150)]
    got, exp = _both(150, 6, rows, P, 16)
    assert got == exp
