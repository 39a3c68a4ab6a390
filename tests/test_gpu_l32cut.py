"""The HBM-resident kernel family (plo_cse_big.hip) against the LITERAL oracle on a cut of the metric's own input.

tests/golden/l32cut_costs.json holds (adds, muls) of seeds 1..16 on rows [0,128) of 32x32x32_15096_L mod 131071
(rows of 48 and 288 entries, 425,784 pair instances), walked by oracle/plo_oracle.c in the build container
(tests/golden/make_l32cut_costs.py, ~3 minutes per seed) -- not by product code.  The kernel must reproduce them
bit-exactly, also when its rare paths are forced: a 64-entry LDS aggregation table (spill list and direct HBM
retirements), a 2-entry tie list (bisection on the key value) and single-entry keys in the LDS table (count + 1/x
packed beside x, or the inversion in the flush).  The counters returned by plo_cse_plan_hbm_counters prove that the
paths were taken (reference include/plinopt_optimize.inl:237-312)."""
import json
import os

import pytest

from plo_testlib import GOLDEN, l32_cut, l32_rows

pytestmark = pytest.mark.gpu
P = 131071


@pytest.fixture(scope="module")
def cut():
    G = json.load(open(os.path.join(GOLDEN, "l32cut_costs.json")))
    _, _, rows = l32_rows(P)
    m, n, rp, c, v = l32_cut(G["row_lo"], G["row_hi"], P, rows)
    assert len(c) == G["nnz"]
    return G, (m, n, rp, c, v), rows


def _run(csr, G, n=16):
    from plinopt_amd import CSEPlan
    m, nn, rp, c, v = csr
    plan = CSEPlan(m, nn, rp, c, v, P)
    assert plan.is_hbm                                        # rows of 288 entries: beyond the wave kernel
    a, mu = plan.cost_many(seed0=G["seed0"], n=n)
    cnt = plan.hbm_counters()
    plan.close()
    return a, mu, cnt


def test_cut_costs_equal_the_literal_oracle(hip, cut):
    G, csr, _ = cut
    a, mu, cnt = _run(csr, G)
    assert a == G["adds"] and mu == G["muls"]
    assert cnt["candidates"] == 16 and cnt["level_rebuilds"] > 16 and cnt["full_scans"] >= 16 and cnt["steps"] > 16 * 600
    print("HBM counters (natural run):", cnt)


def test_cut_with_forced_spills_and_bisection(hip, cut, monkeypatch):
    """64-entry aggregation table: most sweeps overflow it (spill list + direct retirements + slot-list overflow);
    2-entry tie list: every tie pick with more than 2 ties in the chosen column is resolved by bisection."""
    G, csr, _ = cut
    monkeypatch.setenv("PLO_BIG_AGGBITS", "6")
    monkeypatch.setenv("PLO_BIG_SELCAP", "2")
    a, mu, cnt = _run(csr, G)
    assert a == G["adds"] and mu == G["muls"]
    assert cnt["spilled_pairs"] > 10000 and cnt["bisections"] > 100, cnt
    print("HBM counters (forced paths):", cnt)


def test_cut_with_single_ratio_entries(hip, cut, monkeypatch):
    """The LDS aggregation entry without the inverse ratio beside the ratio (what a modulus above 2^21 gets): the flush
    inverts (table of inverses, or Fermat when the modulus is too large for a table)."""
    G, csr, _ = cut
    monkeypatch.setenv("PLO_BIG_NODUAL", "1")
    a, mu, _ = _run(csr, G, n=8)
    assert a == G["adds"][:8] and mu == G["muls"][:8]


def test_cut_with_ratio_identifiers_in_the_pair_keys(hip, cut, monkeypatch):
    """plo::cse_big_kernel<2, ., true> (what a modulus too wide for a residue in the 48-bit pair key gets, e.g. a 31-bit prime with more
    than 256 columns): the ratio field of a key holds the ratio's rank among the sorted ratios, so keys keep their order and the tie
    pick (plinopt_optimize.inl:244-253) walks the same sequence.  Forced here on the modulus of the fixture: same costs, deferred
    and eager."""
    G, csr, _ = cut
    monkeypatch.setenv("PLO_BIG_IDKEYS", "1")
    a, mu, cnt = _run(csr, G, n=8)
    assert a == G["adds"][:8] and mu == G["muls"][:8]
    monkeypatch.setenv("PLO_BIG_EAGER", "1")
    a, mu, cnt = _run(csr, G, n=4)
    assert a == G["adds"][:4] and mu == G["muls"][:4]


def test_cut_b_rows_of_four_chunks(hip, cut):
    """Second cut (tests/golden/l32cutB_costs.json, literal oracle): 24 short rows and two rows of 256 entries -- a row
    is swept in four 64-lane chunks and rewritten in place across chunk boundaries."""
    from plinopt_amd import CSEPlan
    from plo_testlib import l32_cut_b
    _, _, rows = cut
    GB = json.load(open(os.path.join(GOLDEN, "l32cutB_costs.json")))
    m, n, rp, c, v = l32_cut_b(P, rows)
    assert len(c) == GB["nnz"]
    plan = CSEPlan(m, n, rp, c, v, P)
    assert plan.is_hbm
    assert plan.cost_many(seed0=GB["seed0"], n=len(GB["adds"])) == (GB["adds"], GB["muls"])
