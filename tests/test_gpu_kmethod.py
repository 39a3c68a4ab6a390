"""`bin/optimizer -K` with everything on the device (plo::kmethod_kernel through `plo_kernel_search`): per restart, the
nullspace decomposition (this build's rule for nullspacedecomp, reference include/plinopt_optimize.inl:689-884), the two LDS
images and the two chained Optimizer calls (:1322-1333) -- against the oracle's independent restatement
(oracle/plo_oracle.c `plo_oracle_kernel_restart`) restart by restart."""
import os

import pytest

from plo_testlib import DATA, OracleMatrix

pytestmark = pytest.mark.gpu
P = 131071
NAMES = ["4x4x4_49_156_L.sms", "4x4x4_49_156_R.sms", "4x4x4_49_156_P.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "4x4x4_48_rational_P.sms",
         "2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms", "3x3x3_23_58_L.sms", "3x3x3_23_58_P.sms", "2o2o4_5_Toom3_P.sms"]


def usable(M):
    return M.m <= 128 and M.n <= 64 and M.m - M.n <= 64 and M.kernel_restart(1) is not None


def with_identity(M):
    """[M ; I]: what `-F` hands to the kernel method (reference include/plinopt_optimize.inl:640-648, the identity added to the goals)"""
    rp, c, v = list(M.rowptr), list(M.col), list(M.val)
    for j in range(M.n):
        c.append(j); v.append(1); rp.append(len(c))
    return OracleMatrix(M.m + M.n, M.n, rp, c, v, M.p)


@pytest.mark.parametrize("name", NAMES)
def test_every_restart_equals_the_oracle(hip, name):
    from plinopt_amd import kernel_search
    M = OracleMatrix.from_sms(os.path.join(DATA, name), P)
    if not usable(M):
        pytest.skip("more than 128 rows / 64 columns / 64 dependent rows, or zero dimensional kernel")
    n, seed0 = 200, 5
    adds, muls, info, best, st = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n)
    exp = [M.kernel_restart(seed0 + k) for k in range(n)]
    assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == exp
    k = min(range(n), key=lambda k: (exp[k][0] + exp[k][1], exp[k][0], k))
    assert best == (exp[k][0], exp[k][1], seed0 + k)
    assert st["launches"] >= 2 and st["candidates"] == n          # the sizing launch + the search


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "4x4x4_49_156_R.sms", "4x4x4_49_156_P.sms", "3x3x6_40_L.sms", "3x3x3_23_58_P.sms", "4x4x4_48_rational_L.sms"])
def test_more_than_64_rows(hip, name):
    """`-F` on 4x4x4: [M ; I] has 65 rows (49 dependent ones on the L side, 16 on the P side) -- a lane of the wave eliminates rows
    lane and lane + 64, the column masks of Free have two words.  Every restart equals the oracle's."""
    from plinopt_amd import capi, kernel_search
    M = with_identity(OracleMatrix.from_sms(os.path.join(DATA, name), P))
    n, seed0 = 120, 9
    exp = [M.kernel_restart(seed0 + k) for k in range(n)]
    unit = all(x in (1, P - 1) for x in M.val)
    if M.m > 64 and not unit:                                 # non +-1 coefficients with more than 64 rows: ProgramGen of the wave kernel keeps one row per lane
        with pytest.raises(capi.PloError) as e:
            kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n)
        assert e.value.code in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED)
        return
    adds, muls, info, best, st = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n)
    assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == exp
    k = min(range(n), key=lambda k: (exp[k][0] + exp[k][1], exp[k][0], k))
    assert best == (exp[k][0], exp[k][1], seed0 + k)


def test_blocks_of_restarts_share_a_decomposition(hip):
    """per_block = 16 (`--kernel-block 16`): the decomposition of a block comes from its first seed; the host path does the same"""
    from plinopt_amd import kernel_search
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    n, seed0, per = 96, 40, 16
    adds, muls, info, best, _ = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n, per_block=per)
    for b in range(n // per):
        r = M.kernel_restart(seed0 + b * per)
        assert info[b * per] == r[2:] and (adds[b * per], muls[b * per]) == r[:2]          # first restart of a block = the oracle's restart of that seed
        assert all(info[b * per + k] == r[2:] for k in range(per))


def test_unsupported_shapes_are_reported(hip):
    from plinopt_amd import capi, kernel_search
    M = OracleMatrix.from_sms(os.path.join(DATA, "2x2x2_7_DPS-accurate_P.sms"), P)          # 4 x 7, full row rank
    with pytest.raises(capi.PloError) as e:
        kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, 1, 10)
    assert e.value.code == capi.PLO_E_UNSUPPORTED


def test_a_full_pair_table_is_repeated_with_more_slots(hip, monkeypatch):
    """Dep's pair table is sized from a sample; when a restart fills it the device reports it and the call is repeated with twice
    the slots (here the first size is forced 64 times too small): same results as the oracle."""
    from plinopt_amd import kernel_search
    monkeypatch.setenv("PLO_KMETHOD_PAIRS_DIV", "64")
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    n, seed0 = 120, 900
    adds, muls, info, best, st = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n)
    assert st["launches"] >= 3                                     # sizing + at least one repeated search launch
    assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == [M.kernel_restart(seed0 + k) for k in range(n)]


def test_more_entries_than_sampled_is_repeated_with_the_hard_bound(hip, monkeypatch):
    """Dep's arrays are sized from the entry counts of a sample of the decompositions (rows packed per restart); a restart with more
    entries is reported by the device and the launch repeated with rows x rank (here the first size is forced 8 times too small)."""
    from plinopt_amd import kernel_search
    monkeypatch.setenv("PLO_KMETHOD_ENT_DIV", "8")
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    n, seed0 = 100, 3
    adds, muls, info, best, st = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, seed0, n)
    assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == [M.kernel_restart(seed0 + k) for k in range(n)]
    assert st["launches"] >= 3                                # sizing + undersized + repeated


@pytest.mark.parametrize("prime", [7, 101, 2147483647])
@pytest.mark.parametrize("name", ["4x4x4_48_rational_L.sms", "3x3x6_40_L.sms", "2x2x2_7_DPS-accurate_L.sms"])
def test_other_moduli(hip, name, prime):
    """small and large moduli (the reference's pipelines check modulo 7, bin/FDT.sh): inverses by Fermat, Barrett products with
    mu = floor(2^64 / p); matrices with rational coefficients (entries vanish or collide modulo small primes)"""
    from plinopt_amd import kernel_search
    M = OracleMatrix.from_sms(os.path.join(DATA, name), prime)
    if not usable(M):
        pytest.skip("shape or zero dimensional kernel modulo %d" % prime)
    n, seed0 = 60, 77
    from plinopt_amd import capi
    try:
        adds, muls, info, best, _ = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), prime, seed0, n)
    except capi.PloError as e:
        raise                                              # (round 2 skipped PLO_E_CAPACITY here: the 44-bit pair key; 51 bits now)
    exp = [M.kernel_restart(seed0 + k) for k in range(n)]
    assert [(a, mu) + i for a, mu, i in zip(adds, muls, info)] == exp
