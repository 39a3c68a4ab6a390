"""Pins the trilplacer oracle (oracle/plo_tril_oracle.c): every program it emits is run by the independent in-place
interpreter of plo_testlib on random rational inputs and must (1) add the bilinear map sum_l (A_l.a)(B_l.b) T_l to c,
(2) restore a and b, (3) contain exactly the reported (ADD, SCA, MUL) operations.  This is the criterion of the
reference's own Maple check (-DINPLACE_CHECKER, include/plinopt_inplace.inl:935-1010)."""
import glob
import os
import random
from fractions import Fraction

import pytest

from plo_testlib import DATA, TRIL_BASE_SEED, LowHigh, OracleTril, read_sms, run_inplace_program


def triples():
    out = []
    for l in sorted(glob.glob(os.path.join(DATA, "*_L.sms"))):
        r, p = l[:-6] + "_R.sms", l[:-6] + "_P.sms"
        if os.path.exists(r) and os.path.exists(p):
            try:                                       # a few fixtures hold a symbolic placeholder (sqrt(3) as X): not rational
                for f in (l, r, p):
                    read_sms(f)
            except ValueError:
                continue
            out.append(os.path.basename(l)[:-6])
    return out


ALL = triples()
SMALL = [t for t in ALL if os.path.getsize(os.path.join(DATA, t + "_L.sms")) < 4000]


def load(name):
    return OracleTril.from_sms(*(os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")))


def check_program(T, text, ops, rnd):
    na, nb, nc = T.dims
    a0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(na)]
    b0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(nb)]
    c0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(nc)]
    a, b, c = list(a0), list(b0), list(c0)
    counted = run_inplace_program(text, a, b, c)
    ea, eb, et = T.ent
    want = list(c0)
    for l in range(T.m):
        la = sum((v * a0[j] for (i, j), v in ea.items() if i == l), Fraction(0))
        lb = sum((v * b0[j] for (i, j), v in eb.items() if i == l), Fraction(0))
        for (i, k), v in et.items():
            if i == l:
                want[k] += v * la * lb
    assert a == a0, "a not restored"
    assert b == b0, "b not restored"
    assert c == want, "c != c0 + bilinear map"
    assert counted == ops, (counted, ops)


def test_fixture_triples_present():
    assert "4x4x4_49_156" in ALL and "2x2x2_7_Winograd" in ALL and len(ALL) >= 10


@pytest.mark.parametrize("name", SMALL)
def test_programs_compute_the_bilinear_map_in_place(name):
    T = load(name)
    rnd = random.Random(hash(name) & 0xFFFF)
    for seed in [TRIL_BASE_SEED, 0, 1, 2, 12345]:
        for variant in (0, 1):
            ops, text = T.program(seed, variant)
            check_program(T, text, ops, rnd)


def test_cost_many_matches_program_counts_and_search():
    T = load("4x4x4_49_156")
    costs = T.cost_many(seed0=100, nseeds=12)
    for k, (o, u) in enumerate(costs):
        assert T.program(100 + k, 0)[0] == o and T.program(100 + k, 1)[0] == u
        assert o[2] == u[2] == 49
    flat = [(o[0], o[1], 100 + k, 0) for k, (o, u) in enumerate(costs)] + [(u[0], u[1], 100 + k, 1) for k, (o, u) in enumerate(costs)]
    best = min(flat)
    ops, seed, var = T.search(100, 12)
    assert (ops[0], ops[1], seed, var) == best


def test_base_seed_is_the_unpermuted_oriented_program():
    T = load("2x2x2_7_Winograd")
    (o, u), = T.cost_many(seeds=[TRIL_BASE_SEED])
    assert o == u and o[2] == 7
    # Winograd's 7 products: naive in-place cost is 2*(nnz-rows) additions per matrix; the search must not be worse
    assert o[0] <= 2 * ((14 - 7) + (14 - 7) + (14 - 7))


def check_expanded_program(T, text, ops, rnd):
    """`trilplacer -e`: c has one more entry, an AXPY adds the double-size product a_i*b_j as (a_i*b_j)*low to c_k and
    (a_i*b_j)*hig to c_{k'}; with DoubleExpand (plinopt_inplace.inl:676-716) row 2l of TT is row l of T (the low halves)
    and row 2l+1 the same entries one column to the right (the high halves): the reference's Maple check :1047-1061."""
    na, nb, nc = T.dims
    a0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(na)]
    b0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(nb)]
    c0 = [Fraction(rnd.randint(-9, 9), rnd.randint(1, 5)) for _ in range(nc + 1)]
    a, b, c = list(a0), list(b0), [LowHigh(x) for x in c0]
    counted = run_inplace_program(text, a, b, c)
    ea, eb, et = T.ent
    want = [LowHigh(x) for x in c0]
    for l in range(T.m):
        la = sum((v * a0[j] for (i, j), v in ea.items() if i == l), Fraction(0))
        lb = sum((v * b0[j] for (i, j), v in eb.items() if i == l), Fraction(0))
        for (i, k), v in et.items():
            if i == l:
                want[k] = want[k] + LowHigh(0, v * la * lb, 0)
                want[k + 1] = want[k + 1] + LowHigh(0, 0, v * la * lb)
    assert a == a0 and b == b0, "inputs not restored"
    assert c == want, "c != c0 + expanded bilinear map"
    assert counted == ops, (counted, ops)


@pytest.mark.parametrize("name", SMALL)
def test_expanded_programs_compute_the_double_size_map_in_place(name):
    """trilplacer -e (TransposedDoubleAlgorithm, plinopt_inplace.inl:507-598): README example
    `trilplacer data/1o1o2_3_Karatsuba_{L,R,P}.sms -e`."""
    T = load(name)
    rnd = random.Random((hash(name) >> 3) & 0xFFFF)
    for seed in [TRIL_BASE_SEED, 0, 1, 7]:
        for variant in (0, 1):
            ops, text = T.program(seed, variant, expanded=True)
            check_expanded_program(T, text, ops, rnd)
            assert ops[2] == T.m            # m double-size AXPYs


def test_expanded_cost_many_matches_programs_and_search():
    T = load("4x4x4_49_156")
    costs = T.cost_many(seed0=100, nseeds=8, expanded=True)
    for k, (o, u) in enumerate(costs):
        assert T.program(100 + k, 0, expanded=True)[0] == o and T.program(100 + k, 1, expanded=True)[0] == u
    flat = [(o[0], o[1], 100 + k, 0) for k, (o, u) in enumerate(costs)] + [(u[0], u[1], 100 + k, 1) for k, (o, u) in enumerate(costs)]
    ops, seed, var = T.search(100, 8, expanded=True)
    assert (ops[0], ops[1], seed, var) == min(flat)
