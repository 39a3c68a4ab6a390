"""Pins of the CPU oracle (oracle/plo_oracle.c) against the reference's own fixtures.

The reference pins no op-count or program text (its tests assert semantic
correctness only: Makefile:85-87, bin/FDT.sh:58), so the oracle is pinned by
  (1) every emitted SLP evaluates to the input matrix (the `slpcheck` criterion),
  (2) reported (adds, muls) == op-count of the emitted text (`lineOperations`),
  (3) the hand-checked first tie set of Winograd L,
  (4) best cost <= the stored-SLP bounds of data/*.slp, which themselves verify.
"""
import glob
import json
import os

import pytest

import synth
from plo_testlib import (DATA, GOLDEN, FieldP, FieldQ, OracleMatrix, count_ops, eval_slp, read_sms)

P = 131071
ALL = sorted(os.path.basename(f) for f in glob.glob(os.path.join(DATA, "*.sms")) if "-X_" not in f)


@pytest.mark.parametrize("name", ALL)
def test_emitted_slp_computes_the_matrix(name):
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    D = M.dense()
    F = FieldP(P)
    for seed in (0, 1, 99):
        a, mu, txt = M.optimizer(seed)
        assert eval_slp(txt, F) == D
        assert count_ops(txt) == (a, mu)
        assert M.cost_many(seeds=[seed]) == ([a], [mu])       # count-only path == text path


def test_winograd_first_tie_set_hand_checked():
    # 6 distinct triples; two of frequency 3, in map order (0,2,-1),(2,3,1)
    M = OracleMatrix.from_sms(os.path.join(DATA, "2x2x2_7_Winograd_L.sms"), P)
    ties, maxfrq = M.first_ties()
    assert maxfrq == 3
    assert ties == [(0, 2, P - 1), (2, 3, 1)]


def test_other_tie_sets_from_survey_table():
    # SURVEY.md 8a size table: max frequency (number of ties)
    for name, frq, nties in [("cyclic.sms", 4, 8), ("4x4x4_49_156_L.sms", 12, 12), ("4x4x4_49_156_P.sms", 12, 2)]:
        M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
        ties, maxfrq = M.first_ties()
        assert (maxfrq, len(ties)) == (frq, nties), name
        assert ties == sorted(ties)


@pytest.mark.parametrize("name,bound", [
    ("2x2x2_7_Winograd_L", 4), ("2x2x2_7_Winograd_R", 4), ("2x2x2_7_Winograd_P", 7),
])
def test_reaches_known_optimum_on_winograd(name, bound):
    M = OracleMatrix.from_sms(os.path.join(DATA, name + ".sms"), P)
    a, mu, seed = M.search(0, 200)
    assert (a, mu) == (bound, 0)


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L", "3x3x3_23_58_L", "4x4x4_49_156_L", "4x4x4_49_156_P"])
def test_stored_slp_is_a_valid_bound(name):
    """data/<name>.slp computes data/<name>.sms (data/Makefile:37-38 `%.chk`); its op-count is an
    achievable bound that the greedy search approaches (quality sanity, not parity)."""
    m, n, ent = read_sms(os.path.join(DATA, name + ".sms"))
    txt = open(os.path.join(DATA, name + ".slp")).read()
    assert eval_slp(txt, FieldQ) == ent
    stored_adds, stored_muls = count_ops(txt)
    M = OracleMatrix.from_sms(os.path.join(DATA, name + ".sms"), P)
    a, mu, _ = M.search(0, 300, nthreads=4)
    na, nm = M.naive_ops()
    assert a + mu < na + nm
    assert a + mu <= stored_adds + stored_muls + max(4, (stored_adds + stored_muls) // 8)


def test_triangle_and_factoring_paths_emit_valid_programs():
    F = FieldP(P)
    tri = 0
    for s in range(600):
        m, n, rows = synth.small_valued(s, P)
        rp, c, v = synth.to_csr(rows, P)
        M = OracleMatrix(m, n, rp, c, v, P)
        a, mu, txt = M.optimizer(s)
        assert eval_slp(txt, F) == M.dense(), (s, rows)
        assert count_ops(txt) == (a, mu)
    m, n, rows = 2, 2, [{0: 2}, {0: 6, 1: 3}]
    rp, c, v = synth.to_csr(rows, P)
    a, mu, txt = OracleMatrix(m, n, rp, c, v, P).optimizer(0)
    assert (a, mu) == (1, 2)            # one triangle: 2 multiplications instead of 3


def test_golden_costs_fixture():
    """Frozen (adds, muls) vectors of the oracle itself (tests/golden/oracle_costs.json, made by
    tests/golden/make_oracle_costs.py): guards the checker against silent drift."""
    G = json.load(open(os.path.join(GOLDEN, "oracle_costs.json")))
    for name, rec in G["matrices"].items():
        M = OracleMatrix.from_sms(os.path.join(DATA, name), G["p"])
        a, mu = M.cost_many(seed0=rec["seed0"], nseeds=len(rec["adds"]))
        assert a == rec["adds"] and mu == rec["muls"], name


def test_enumerated_schedules_emit_valid_programs():
    """-E: every schedule of RecSub's tree addressed by index prints a program that computes the matrix, with the reported
    op-count; on Winograd's L the tree has three leaves and the best needs the known 4 additions."""
    M = OracleMatrix.from_sms(os.path.join(DATA, "2x2x2_7_Winograd_L.sms"), 131071)
    a, mu, pr = M.enum_cost_many(0, 64)
    assert max(pr) == 3 and min(a) == 4
    for name, idxs in (("2x2x2_7_Winograd_L.sms", range(3)), ("cyclic.sms", [0, 1, 5, 999, 123456789]), ("4x4x4_48_rational_P.sms", [0, 7, 10 ** 9])):
        M = OracleMatrix.from_sms(os.path.join(DATA, name), 131071)
        D = M.dense()
        for idx in idxs:
            adds, muls, prod, text = M.enum_optimizer(idx)
            assert eval_slp(text, FieldP(131071)) == D
            assert count_ops(text) == (adds, muls) and prod >= 1


# ----------------------------------------------------------------------------- stored factorizations of the reference
def _spmul(A, B):
    by_row = {}
    for (i, k), v in B.items():
        by_row.setdefault(i, []).append((k, v))
    C = {}
    for (i, k), v in A.items():
        for j, w in by_row.get(k, []):
            C[(i, j)] = C.get((i, j), 0) + v * w
    return {k: v for k, v in C.items() if v != 0}


def _alt_pairs():
    import glob
    out = []
    for alt in sorted(glob.glob(os.path.join(DATA, "*-ALT_*.sms"))):
        cob, orig = alt.replace("-ALT_", "-CoB_"), alt.replace("-ALT_", "_")
        if os.path.exists(cob) and os.path.exists(orig):
            out.append(os.path.basename(alt))
    return out


@pytest.mark.parametrize("alt", _alt_pairs())
def test_stored_alternative_bases_factor_their_matrix(alt):
    """The reference holds factorizations X = ALT . CoB of some algorithms (bin/factorizer outputs, data/*-ALT_*.sms with
    data/*-CoB_*.sms; for the P side X = CoB . ALT): what `consistency` (plinopt_sparsify.inl:872-907) asserts.  They pin the
    fixtures and give the sparsity a factorization of these inputs can reach (used as a bound by the factorizer tests).
    The 2x2x2_7_DPS-accurate-ALT files are the sparse basis of Winograd's algorithm, not a factorization of the
    DPS-accurate matrices: only their shape is checked."""
    mo, no, X = read_sms(os.path.join(DATA, alt.replace("-ALT_", "_")))
    ma, na, A = read_sms(os.path.join(DATA, alt))
    mc, nc, C = read_sms(os.path.join(DATA, alt.replace("-ALT_", "-CoB_")))
    if alt.startswith("2x2x2_7_DPS-accurate"):
        assert (ma, na) == (mo, no) and mc == nc
        return
    if alt.endswith("_P.sms"):
        assert nc == ma and _spmul(C, A) == X
    else:
        assert na == mc and _spmul(A, C) == X
    assert len(A) < len(X)                     # the alternative basis is sparser than the algorithm's own matrix
