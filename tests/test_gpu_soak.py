"""A bounded slice of every soak (tests/soak_*.py: one-off scripts of rounds 2-3, tens of thousands of random cases, 0 mismatches) inside
`-m gpu`, so that the driver re-runs them every round: fixed seeds, fixed counts, each kernel against the CPU ORACLE (oracle/*.c), not
against the product's host engines.

  wave kernel          200 random matrices of up to 200 x 64 over five moduli, 8 seeds each              (soak_wave.py)
  HBM family            40 matrices the literal oracle can walk, forced through plo_cse_big.hip, 3 seeds;
                        two of the dense few-valued kind whose live triples outgrow the planned structures (soak_hbm.py);
                        36 more over moduli of 25 to 31 bits (ratio identifiers in the pair keys)
  kernel method        200 matrices x 12 restarts: decomposition, both images, both programs             (soak_misc.py)
  in-place trilinear   500 triples (+-1 and rational, plain and `-e`) x 12 seeds x both variants          (soak_misc.py)
  change of basis      300 groups of 1-4 enumerations, single and batched launches                         (soak_misc.py --cob)
  schedule enumeration 500 matrices x 96 schedules of the exhaustive tree (-E)                              (soak_misc.py --enum)

A refusal the header documents (PLO_E_CAPACITY / PLO_E_UNSUPPORTED) is counted and bounded, anything else fails."""
import os
import random
from concurrent.futures import ThreadPoolExecutor
from fractions import Fraction

import pytest

import synth
from plo_testlib import OracleMatrix, OracleTril

pytestmark = pytest.mark.gpu
P = 131071


def _refusal(e):
    from plinopt_amd import capi
    return getattr(e, "code", 0) in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED)


def test_wave_kernel_on_random_matrices(hip):
    from plinopt_amd import CSEPlan
    ran = refused = 0
    s = 0
    while ran < 200:
        rng = random.Random(9000 + s)
        p = rng.choice([7, 101, 131071, 2147483629, 2147483647])
        unit = rng.random() < 0.4
        m = rng.randint(2, 200 if unit else 64)
        n = rng.randint(2, 64)
        dens = rng.choice([0.1, 0.3, 0.6, 0.9])
        vals = [1, p - 1] if unit else [1, p - 1] + [rng.randint(2, p - 2) % p or 1 for _ in range(rng.choice([1, 3, 30]))]
        rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < dens} for _ in range(m)]
        rows = [r if r else {0: 1} for r in rows]
        s += 1
        if sum(len(r) for r in rows) > 500:                     # (the literal oracle rescans the pair map at every step)
            continue
        rp, c, v = synth.to_csr(rows, p)
        try:
            plan = CSEPlan(m, n, rp, c, v, p)
            got = plan.cost_many(seed0=s, n=8)
            plan.close()
        except Exception as e:
            if _refusal(e):
                refused += 1
                continue
            raise
        assert got == tuple(OracleMatrix(m, n, rp, c, v, p).cost_many(seed0=s, nseeds=8, nthreads=8)), (s - 1, m, n, p)
        ran += 1
    assert refused <= 20


def _hbm_cases():
    out = []
    for s in range(80):
        if len(out) == 38:
            break
        rng = random.Random(7100 + s)
        m, n = rng.randint(20, 90), rng.randint(16, 64)
        dens = rng.choice([0.15, 0.25, 0.4])
        nv = rng.choice([0, 1, 3, 40, 600])                    # kernel modes 2 (ratio identifiers), 1 (value table in LDS) and 0 (in global memory)
        vals = [1, P - 1] + [rng.randint(2, P - 2) for _ in range(nv)]
        rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < dens} for _ in range(m)]
        rows = [r if r else {0: 1} for r in rows]
        if sum(len(r) for r in rows) <= 1500:
            out.append((s, m, n, rows))
    for s in (0, 1):                                            # dense, two or three distinct values: the live triples of frequency >= 2 GROW
        rng = random.Random(7300 + s)
        m, n = 40, 32
        vals = [1, P - 1] + [rng.randint(2, P - 2) for _ in range(1 + s)]
        out.append((1000 + s, m, n, [{j: rng.choice(vals) for j in range(n) if rng.random() < 0.8} for _ in range(m)]))
    return out


def test_hbm_family_on_matrices_the_literal_oracle_walks(hip):
    from plinopt_amd import CSEPlan
    cases = _hbm_cases()
    assert len(cases) == 40

    def oracle(case):
        s, m, n, rows = case
        rp, c, v = synth.to_csr(rows, P)
        return tuple(OracleMatrix(m, n, rp, c, v, P).cost_many(seed0=s * 10, nseeds=3, nthreads=1))
    with ThreadPoolExecutor(max_workers=8) as ex:
        exp = list(ex.map(oracle, cases))
    refits = 0
    for (s, m, n, rows), e in zip(cases, exp):
        rp, c, v = synth.to_csr(rows, P)
        if os.environ.get("PLO_TEST_TRACE"):
            print("case", s, m, n, len(c), len(set(v)), flush=True)
        plan = CSEPlan(m, n, rp, c, v, P, hbm=True)
        assert plan.is_hbm
        got = plan.cost_many(seed0=s * 10, n=3)
        refits += plan.hbm_counters()["eager_refits"]
        plan.close()
        assert got == e, (s, m, n)
    print("HBM soak slice: %d matrices, %d plan rebuilds" % (len(cases), refits))


def test_hbm_family_with_wide_moduli(hip):
    """Moduli of 25 to 31 bits on the HBM family: residues in the pair keys while they leave room for the columns, ratio identifiers
    beyond (31 bits: always) -- 36 matrices of at most 32 distinct coefficients, 3 seeds each, against the literal oracle."""
    from plinopt_amd import CSEPlan
    cases = []
    for s in range(36):
        rng = random.Random(7500 + s)
        p = rng.choice([2147483629, 2147483647, 1073741827, 16777259])
        m, n = rng.randint(20, 80), rng.randint(16, 300)
        per_row = rng.randint(3, 12)
        nv = rng.choice([0, 1, 3, 12, 30])
        vals = [1, p - 1] + [rng.randint(2, p - 2) for _ in range(nv)]
        rows = [{j: rng.choice(vals) for j in rng.sample(range(n), min(n, per_row))} for _ in range(m)]
        cases.append((s, p, m, n, rows))

    def oracle(case):
        s, p, m, n, rows = case
        rp, c, v = synth.to_csr(rows, p)
        return tuple(OracleMatrix(m, n, rp, c, v, p).cost_many(seed0=s * 7, nseeds=3, nthreads=1))
    with ThreadPoolExecutor(max_workers=8) as ex:
        exp = list(ex.map(oracle, cases))
    for (s, p, m, n, rows), e in zip(cases, exp):
        rp, c, v = synth.to_csr(rows, p)
        plan = CSEPlan(m, n, rp, c, v, p, hbm=True)
        assert plan.is_hbm
        got = plan.cost_many(seed0=s * 7, n=3)
        plan.close()
        assert got == e, (s, p, m, n)


def test_kernel_method_on_random_matrices(hip):
    from plinopt_amd import kernel_search
    ran = refused = 0
    s = 0
    while ran < 200:
        rng = random.Random(11000 + s)
        s += 1
        n = rng.randint(2, 24)
        m = rng.randint(n + 1, min(128, n + 40))
        vals = [1, P - 1] if rng.random() < 0.7 else [1, P - 1, 2, P - 2, 3]
        rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < rng.choice([0.2, 0.5])} for _ in range(m)]
        rows = [r if r else {rng.randrange(n): 1} for r in rows]
        rp, c, v = synth.to_csr(rows, P)
        M = OracleMatrix(m, n, rp, c, v, P)
        if M.kernel_restart(1) is None:
            continue
        try:
            adds, muls, info, best, st = kernel_search((m, n, rp, c, v), P, s, 12)
        except Exception as e:
            if _refusal(e):
                refused += 1
                continue
            raise
        ran += 1
        assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == [M.kernel_restart(s + k) for k in range(12)], (s - 1, m, n)
    assert refused <= 30


def test_trilinear_kernel_on_random_triples(hip):
    from plinopt_amd import TrilPlan
    vals_q = [Fraction(1), Fraction(-1), Fraction(1, 2), Fraction(-2), Fraction(3), Fraction(-2, 3)]
    ran = refused = 0
    s = 0
    while ran < 500:
        rng = random.Random(12000 + s)
        s += 1
        m = rng.randint(2, 40)
        na, nb, nc = rng.randint(2, 12), rng.randint(2, 12), rng.randint(2, 12)
        unit = rng.random() < 0.5
        expanded = rng.random() < 0.3

        def mat(rows, cols):
            e = {}
            for i in range(rows):
                js = [j for j in range(cols) if rng.random() < 0.4] or [rng.randrange(cols)]
                for j in js:
                    e[(i, j)] = rng.choice(vals_q[:2] if unit else vals_q)
            return e
        A, B = (m, na, mat(m, na)), (m, nb, mat(m, nb))
        T = mat(m, nc)                                          # T = C^T: m x nc, every row non-empty
        O = OracleTril(A, B, (nc, m, {(j, i): v for (i, j), v in T.items()}))
        try:
            if unit:
                G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=expanded)
            else:
                G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num], [int(x) for x in den]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=expanded)
            got = G.cost_many(seed0=s, n=12)
        except Exception as e:
            if _refusal(e):
                refused += 1
                continue
            raise
        ran += 1
        assert got == O.cost_many(seed0=s, nseeds=12, expanded=expanded), (s - 1, m, na, nb, nc, unit, expanded)
    assert refused == 0


def test_change_of_basis_enumerations_on_random_blocks(hip):
    from plinopt_amd import cob_search, cob_search_batch
    from plo_testlib import oracle_cob_search
    for s in range(300):
        rng = random.Random(13000 + s)
        n, m = rng.randint(1, 13), rng.randint(1, 70)
        row = rng.randint(0, n - 1)
        off = (row // 4) * 4
        probs = []
        for _ in range(rng.randint(1, 4)):
            p = rng.choice([7, 101, 131071, 2147483629, 2147483647])
            TM = [rng.choice([0, 0, 1, p - 1, 2, 3, 5]) % p for _ in range(n * m)]
            Cand = [0] * (n * n)
            for i in range(row):
                for j in range(n):
                    Cand[i * n + j] = rng.choice([0, 0, 1, p - 1, 2]) % p
            C = rng.randint(1, 7)
            coeffs = [0, 1, p - 1, 2 % p, (p - 2) % p, 3 % p, pow(2, -1, p), (p - pow(2, -1, p)) % p, 5 % p][:C]
            w0 = rng.choice([-1, 0, m // 2, m])
            probs.append((TM, Cand, coeffs, p, w0, 0 if w0 >= 0 else -1))
        exp = [oracle_cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, w1) for (TM, Cand, coeffs, p, w0, w1) in probs]
        got, _ = cob_search_batch(n, m, row, off, probs)
        one = [cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, w1)[0] for (TM, Cand, coeffs, p, w0, w1) in probs]
        assert got == exp and one == exp, (s, n, m, row)


def test_schedule_enumeration_on_random_matrices(hip):
    from plinopt_amd import CSEPlan
    ran = 0
    s = 0
    while ran < 500:
        rng = random.Random(14000 + s)
        s += 1
        p = rng.choice([7, 131071, 2147483629])
        m, n = rng.randint(2, 14), rng.randint(2, 12)
        vals = [1, p - 1] if rng.random() < 0.5 else [1, p - 1, 2 % p or 1, 3 % p or 1]
        rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < 0.5} for _ in range(m)]
        rows = [r if r else {0: 1} for r in rows]
        rp, c, v = synth.to_csr(rows, p)
        try:
            plan = CSEPlan(m, n, rp, c, v, p)
            first = rng.choice([0, 0, 17, 1000])
            got = plan.enum_cost_many(first, 96)
            plan.close()
        except Exception as e:
            if _refusal(e):
                continue
            raise
        ran += 1
        assert got == tuple(OracleMatrix(m, n, rp, c, v, p).enum_cost_many(first, 96, nthreads=8)), (s - 1, m, n, p)
