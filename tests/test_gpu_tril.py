"""GPU parity of the trilplacer search (plo_tril_* of include/plinopt_hip.h) against oracle/plo_tril_oracle.c:
(ADD, SCA, MUL) of both variants for every seed, bit-exact, and the same argmin."""
import glob
import os

import pytest

from plo_testlib import DATA, TRIL_BASE_SEED, OracleTril, read_sms

pytestmark = pytest.mark.gpu


def unit_triples():
    out = []
    for l in sorted(glob.glob(os.path.join(DATA, "*_L.sms"))):
        r, p = l[:-6] + "_R.sms", l[:-6] + "_P.sms"
        if not (os.path.exists(r) and os.path.exists(p)):
            continue
        try:
            mats = [read_sms(f) for f in (l, r, p)]
        except ValueError:
            continue
        if all(abs(v) == 1 for _, _, e in mats for v in e.values()) and mats[2][0] <= 64 and mats[0][1] <= 64 and mats[1][1] <= 64:
            rows_ok = all(len({i for (i, j) in e}) == m for m, n, e in mats[:2]) and len({j for (i, j) in mats[2][2]}) == mats[2][1]
            if rows_ok:
                out.append(os.path.basename(l)[:-6])
    return out


UNIT = unit_triples()


def plans(name):
    from plinopt_amd import TrilPlan
    O = OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))
    mats = [(n, rp, col, [int(x) for x in num]) for n, (rp, col, num, den) in zip(O.dims, O.csr)]
    return O, TrilPlan(O.m, mats)


def test_unit_fixtures_present():
    assert "4x4x4_49_156" in UNIT and "2x2x2_7_Winograd" in UNIT and len(UNIT) >= 5


@pytest.mark.parametrize("name", UNIT)
def test_cost_many_bit_exact(hip, name):
    O, G = plans(name)
    n = 48 if O.m > 30 else 96
    seeds = [TRIL_BASE_SEED, 0, 1, 2**40 + 7] + list(range(1000, 1000 + n))
    assert G.cost_many(seeds=seeds) == O.cost_many(seeds=seeds)


def test_search_same_argmin(hip):
    O, G = plans("4x4x4_49_156")
    assert G.search(5000, 300) == O.search(5000, 300)
    O, G = plans("2x2x2_7_Winograd")
    assert G.search(0, 2000) == O.search(0, 2000)


def all_triples():
    out = []
    for l in sorted(glob.glob(os.path.join(DATA, "*_L.sms"))):
        r, p = l[:-6] + "_R.sms", l[:-6] + "_P.sms"
        if not (os.path.exists(r) and os.path.exists(p)):
            continue
        try:
            [read_sms(f) for f in (l, r, p)]
        except ValueError:
            continue
        out.append(os.path.basename(l)[:-6])
    return out


ALL = all_triples()
RATIONAL = [t for t in ALL if t not in UNIT]


def test_every_fixture_triple_is_listed():
    assert len(ALL) == 47 and len(RATIONAL) == 25


@pytest.mark.parametrize("name", RATIONAL)
def test_rational_fixtures_cost_many_bit_exact(hip, name):
    """Round 3: the reference's matrices are rationals (Atom::_val is a Givaro::Rational, plinopt_inplace.inl:19); the device programs
    carry the coefficients modulo a 31-bit prime (plo_tril_plan_create_q).  All six counts of every seed equal the oracle's, which
    works over Q -- on each of the 25 {L,R,P} triples with coefficients other than +-1 (with the 22 unit ones: all 47)."""
    from plinopt_amd import TrilPlan
    O = OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))
    G = TrilPlan(O.m, [(n, rp, col, [int(x) for x in num], [int(x) for x in den]) for n, (rp, col, num, den) in zip(O.dims, O.csr)])
    n = 32 if O.m > 30 else 64
    seeds = [TRIL_BASE_SEED, 0, 1, 2**40 + 7] + list(range(3000, 3000 + n))
    assert G.cost_many(seeds=seeds) == O.cost_many(seeds=seeds)


def test_rational_search_same_argmin(hip):
    from plinopt_amd import TrilPlan
    for name, n in (("4x4x4_48_rational", 300), ("2x2x2_7_DPS-accurate", 1500)):
        O = OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))
        G = TrilPlan(O.m, [(nn, rp, col, [int(x) for x in num], [int(x) for x in den]) for nn, (rp, col, num, den) in zip(O.dims, O.csr)])
        assert G.search(100, n) == O.search(100, n)


def test_integer_entries_through_the_integer_interface(hip):
    """plo_tril_plan_create_x with entries 2 and 3: refused in rounds 1-2, now the rational instantiation (denominators 1)"""
    from fractions import Fraction
    from plinopt_amd import TrilPlan
    A = [[1, 2], [1, 1]]; B = [[1, 1], [-1, 1]]; T = [[1, 1], [3, 1]]
    dic = lambda M: {(i, j): Fraction(v) for i, r in enumerate(M) for j, v in enumerate(r) if v}
    O = OracleTril((2, 2, dic(A)), (2, 2, dic(B)), (2, 2, {(j, i): v for (i, j), v in dic(T).items()}))
    csr = lambda M: (2, [0, 2, 4], [0, 1, 0, 1], [v for r in M for v in r])
    G = TrilPlan(2, [csr(A), csr(B), csr(T)])
    seeds = [TRIL_BASE_SEED, 0, 1, 2, 3, 4, 5]
    assert G.cost_many(seeds=seeds) == O.cost_many(seeds=seeds)


@pytest.mark.parametrize("name", RATIONAL)
def test_expanded_rational_fixtures_cost_many_bit_exact(hip, name):
    """Round 4: `trilplacer -e` with rational coefficients on the device (refused until round 3): TransposedDoubleAlgorithm
    (plinopt_inplace.inl:507-598) with its scaling atoms `*y`, `*a` and the atoms of z = -y c y and c (:532-535, :561-565) on
    DoubleExpand(T) (:676-716), coefficients as residues modulo the 31-bit prime.  All six counts of every seed equal the oracle's,
    which works over Q, on each of the 25 rational triples (with the 22 unit ones below: all 47 with expanded=True)."""
    from plinopt_amd import TrilPlan
    O = OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))
    G = TrilPlan(O.m, [(n, rp, col, [int(x) for x in num], [int(x) for x in den]) for n, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=True)
    n = 24 if O.m > 30 else 48
    seeds = [TRIL_BASE_SEED, 0, 1, 2**40 + 7] + list(range(4000, 4000 + n))
    assert G.cost_many(seeds=seeds) == O.cost_many(seeds=seeds, expanded=True)


def test_expanded_rational_search_and_cli(hip):
    import re
    import subprocess
    from plinopt_amd import TrilPlan
    from plo_testlib import ROOT
    for name, n in (("4x4x4_48_rational", 200), ("2x2x2_7_DPS-accurate", 800)):
        files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
        O = OracleTril.from_sms(*files)
        G = TrilPlan(O.m, [(nn, rp, col, [int(x) for x in num], [int(x) for x in den]) for nn, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=True)
        assert G.search(100, n) == O.search(100, n, expanded=True)
        g = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-e", "-O", str(n), "--seed", "100"] + files, capture_output=True, text=True, timeout=600)
        h = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-e", "-O", str(n), "--seed", "100", "--gpu", "0"] + files, capture_output=True, text=True, timeout=600)
        assert g.returncode == 0 and h.returncode == 0, g.stderr + h.stderr
        assert "restarts on GPU" in g.stderr and "restarts on host" in h.stderr and g.stdout == h.stdout
        assert re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", g.stderr) == re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", h.stderr)


def test_ten_thousand_seeds_on_config4(hip):
    """BASELINE configs[3] (4x4x4_49_156 L, R, P): 10^4 restarts, all six counts per restart and the argmin."""
    O, G = plans("4x4x4_49_156")
    n = 10000
    assert G.cost_many(seed0=2 * 10 ** 9, n=n) == O.cost_many(seed0=2 * 10 ** 9, nseeds=n)
    assert G.search(2 * 10 ** 9, n) == O.search(2 * 10 ** 9, n)


# ----------------------------------------------------------------------------- trilplacer -e
@pytest.mark.parametrize("name", UNIT)
def test_expanded_cost_many_bit_exact(hip, name):
    """`trilplacer -e`: the c program is TransposedDoubleAlgorithm on DoubleExpand(T) (plinopt_inplace.inl:507-598, :676-716;
    plo_tril_plan_create_x with expanded = 1): (ADD, SCA, MUL) of both variants per seed against the oracle, whose
    expanded programs pass the double-size in-place check (tests/test_tril_oracle.py)."""
    from plinopt_amd import TrilPlan
    O = OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))
    G = TrilPlan(O.m, [(n, rp, col, [int(x) for x in num]) for n, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=True)
    n = 32 if O.m > 30 else 64
    seeds = [TRIL_BASE_SEED, 0, 1, 2**40 + 7] + list(range(2000, 2000 + n))
    assert G.cost_many(seeds=seeds) == O.cost_many(seeds=seeds, expanded=True)


def test_expanded_search_same_argmin_and_cli(hip):
    import re
    import subprocess
    from plinopt_amd import TrilPlan
    from plo_testlib import ROOT
    name = "4x4x4_49_156"
    files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
    O = OracleTril.from_sms(*files)
    G = TrilPlan(O.m, [(n, rp, col, [int(x) for x in num]) for n, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=True)
    assert G.search(700, 400) == O.search(700, 400, expanded=True)
    # bin/trilplacer -e: GPU search, host replay; same winner and text as the host loop
    g = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-e", "-O", "400", "--seed", "700"] + files, capture_output=True, text=True, timeout=300)
    h = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-e", "-O", "400", "--seed", "700", "--gpu", "0"] + files, capture_output=True, text=True, timeout=300)
    assert g.returncode == 0 and h.returncode == 0, g.stderr + h.stderr
    assert "restarts on GPU" in g.stderr and g.stdout == h.stdout
    assert re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", g.stderr) == re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", h.stderr)


@pytest.mark.parametrize("name,loops", [("4x4x4_48_rational", 600), ("2x2x2_7_DPS-accurate", 3000), ("2x2x2_7_DPS-intermediate-12.0695", 2000), ("3x3x6_40_DPS-accurate", 400)])
def test_trilplacer_cli_runs_rational_inputs_on_the_gpu(hip, name, loops):
    """bin/trilplacer on matrices with rational coefficients: the restart loop on the GPU (rounds 1-2: host loop), the winner replayed
    over Q (128-bit rationals) and checked against the device's counts; same program and counts as the host loop."""
    import re
    import subprocess
    from plo_testlib import ROOT
    files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
    g = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-O", str(loops), "--seed", "41"] + files, capture_output=True, text=True, timeout=600)
    h = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-O", str(loops), "--seed", "41", "--gpu", "0"] + files, capture_output=True, text=True, timeout=600)
    assert g.returncode == 0 and h.returncode == 0, g.stderr + h.stderr
    assert "restarts on GPU" in g.stderr and "restarts on host" in h.stderr
    assert g.stdout == h.stdout
    assert re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", g.stderr) == re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", h.stderr)

