"""CPU side of the cut goldens: the literal-oracle costs of tests/golden/l32cut*_costs.json (made in the build
container by tests/golden/make_l32cut_costs.py) against (1) the oracle itself on the small cut B (guards the fixture),
(2) the scalable host engine plo_fast.hpp through `bin/optimizer --replay --engine fast` on both cuts -- the engine
whose costs are the full-size golden values of config 5 (tests/golden/config5_costs.json) -- and (3) the reference's
own criterion: the replayed program computes the cut matrix (bin/SLPchecker)."""
import json
import os
import re
import subprocess

import pytest

from plo_testlib import GOLDEN, ROOT, OracleMatrix, l32_cut, l32_cut_b, l32_rows, write_sms

P = 131071
OPT = os.path.join(ROOT, "bin", "optimizer")
CHK = os.path.join(ROOT, "bin", "SLPchecker")


@pytest.fixture(scope="module")
def rows():
    return l32_rows(P)[2]


def _fast_cost(sms, seed, check=False):
    r = subprocess.run([OPT, "-q", str(P), "--replay", "--engine", "fast", "--seed", str(seed), sms], capture_output=True, text=True, check=True)
    a = int(re.search(r"# (\d+)\tadditions", r.stderr).group(1)); mu = int(re.search(r"# (\d+)\tmultiplications", r.stderr).group(1))
    if check:
        chk = subprocess.run([CHK, "-q", str(P), "-M", sms], input=r.stdout, capture_output=True, text=True)
        assert chk.returncode == 0 and "SUCCESS" in chk.stderr and ("%d,%d" % (a, mu)) in chk.stderr, chk.stderr
    return a, mu


def test_cut_b_fixture_is_what_the_oracle_computes(rows):
    G = json.load(open(os.path.join(GOLDEN, "l32cutB_costs.json")))
    m, n, rp, c, v = l32_cut_b(P, rows)
    assert len(c) == G["nnz"]
    M = OracleMatrix(m, n, rp, c, v, P)
    a, mu = M.cost_many(seed0=G["seed0"], nseeds=2)           # ~6 s per seed
    assert a == G["adds"][:2] and mu == G["muls"][:2]


@pytest.mark.parametrize("which", ["A", "B"])
def test_host_engine_equals_literal_oracle_on_the_cuts(rows, which, tmp_path):
    G = json.load(open(os.path.join(GOLDEN, "l32cut_costs.json" if which == "A" else "l32cutB_costs.json")))
    m, n, rp, c, v = l32_cut(G["row_lo"], G["row_hi"], P, rows) if which == "A" else l32_cut_b(P, rows)
    sms = str(tmp_path / "cut.sms")
    write_sms(sms, m, n, rp, c, v)
    for k in range(len(G["adds"])):
        assert _fast_cost(sms, G["seed0"] + k, check=(k == 0)) == (G["adds"][k], G["muls"][k]), (which, k)


def test_host_engine_equals_literal_oracle_on_the_long_rows(tmp_path):
    """tests/golden/longrow_costs.json (literal oracle on rows of 461-627 entries, tests/golden/make_longrow_costs.py): the scalable host
    engine -- the source of config 5's full-size goldens -- gives the same costs, and its program computes the matrix."""
    from plo_testlib import longrow_matrix
    G = json.load(open(os.path.join(GOLDEN, "longrow_costs.json")))
    m, n, rp, c, v = longrow_matrix(P)
    assert len(c) == G["nnz"] and [rp[i + 1] - rp[i] for i in range(m)] == G["row_lengths"]
    sms = str(tmp_path / "longrow.sms")
    write_sms(sms, m, n, rp, c, v)
    for k in range(len(G["adds"])):
        assert _fast_cost(sms, G["seed0"] + k, check=(k == 0)) == (G["adds"][k], G["muls"][k]), k
