"""Change-of-basis search (bin/sparsifier's hot loop): the GPU enumeration of |Coeffs|^4 candidate rows against
the literal CPU oracle (rank by Gaussian elimination per candidate), and the CLI end to end."""
import os
import random
import re
import subprocess

import pytest

from plo_testlib import DATA, ROOT, oracle_cob_search, read_sms, to_csr_mod

pytestmark = pytest.mark.gpu
P = 131071
SPS = os.path.join(ROOT, "bin", "sparsifier")


def _dense_T(name, p):
    """TM = M^T as a flat row-major n x m list"""
    m, n, ent = read_sms(os.path.join(DATA, name))
    rp, c, v = to_csr_mod(m, n, ent, p)
    TM = [0] * (n * m)
    for i in range(m):
        for k in range(rp[i], rp[i + 1]):
            TM[c[k] * m + i] = v[k]
    return n, m, TM


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "2x2x2_7_DPS-accurate_L.sms", "3x3x3_23_58_L.sms"])
def test_cob_enumeration_matches_oracle(hip, name):
    from plinopt_amd import cob_search
    n, m, TM = _dense_T(name, P)
    rng = random.Random(1)
    coeffs = [0, 1, P - 1, 2, P - 2, pow(2, -1, P), P - pow(2, -1, P)]
    Cand = [0] * (n * n)
    for row in range(min(n, 9)):                   # build the change of basis row by row, as localSparsifier does
        off = (row // 4) * 4
        for C in (3, 5, 7):
            for (w0, w1) in ((-1, -1), (m // 3, 1)):
                got, st = cob_search(n, m, TM, Cand, row, off, coeffs[:C], P, w0, w1)
                exp = oracle_cob_search(n, m, TM, Cand, row, off, coeffs[:C], P, w0, w1)
                assert got == exp, (name, row, C, w0, w1)
                assert st["candidates"] == C ** 4
        zv, zw, idx, found = oracle_cob_search(n, m, TM, Cand, row, off, coeffs[:5], P)
        assert found
        ids = []
        for _ in range(4):
            ids.append(idx % 5)
            idx //= 5
        ids.reverse()
        for t in range(4):
            if off + t < n:
                Cand[row * n + off + t] = coeffs[ids[t]]
        assert rng is not None


def test_cob_random_blocks(hip):
    from plinopt_amd import cob_search
    rng = random.Random(7)
    for trial in range(40):
        p = rng.choice([7, 101, 131071, 2147483629])
        n = rng.randint(1, 7)
        m = rng.randint(1, 30)
        TM = [rng.choice([0, 0, 1, p - 1, 2, 3]) % p for _ in range(n * m)]
        row = rng.randint(0, n - 1)
        off = (row // 4) * 4
        Cand = [0] * (n * n)
        for i in range(row):                        # previous rows: random (possibly dependent) rows
            for j in range(n):
                Cand[i * n + j] = rng.choice([0, 1, p - 1, 2]) % p
        C = rng.randint(1, 5)
        coeffs = [0, 1, p - 1, 2 % p, (p - 2) % p][:C]
        w0 = rng.choice([-1, 0, m // 2])
        got, _ = cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, 0 if w0 >= 0 else -1)
        exp = oracle_cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, 0 if w0 >= 0 else -1)
        assert got == exp, (trial, p, n, m, row, C, w0)


def test_cob_batch_of_enumerations_in_one_launch(hip):
    """plo_cob_search_batch: up to four enumerations of one shape (different moduli, matrices, thresholds; dependent chosen rows in
    some) in ONE launch -- each result equals the oracle's for that enumeration, and the launch counter says 1."""
    from plinopt_amd import cob_search_batch
    rng = random.Random(11)
    for trial in range(30):
        n, m = rng.randint(1, 7), rng.randint(1, 40)
        row = rng.randint(0, n - 1)
        off = (row // 4) * 4
        probs = []
        for _ in range(rng.randint(1, 4)):
            p = rng.choice([7, 101, 131071, 2147483629, 2147483647])
            TM = [rng.choice([0, 0, 1, p - 1, 2, 3]) % p for _ in range(n * m)]
            Cand = [0] * (n * n)
            for i in range(row):
                for j in range(n):
                    Cand[i * n + j] = rng.choice([0, 1, p - 1, 2]) % p
            C = rng.randint(1, 6)
            coeffs = [0, 1, p - 1, 2 % p, (p - 2) % p, 3 % p][:C]
            w0 = rng.choice([-1, 0, m // 2])
            probs.append((TM, Cand, coeffs, p, w0, 0 if w0 >= 0 else -1))
        got, st = cob_search_batch(n, m, row, off, probs)
        exp = [oracle_cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, w1) for (TM, Cand, coeffs, p, w0, w1) in probs]
        assert got == exp, (trial, n, m, row)
        assert st["launches"] <= 1 and st["candidates"] == sum(len(q[2]) ** 4 for q in probs)


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms", "3x3x3_23_58_P.sms"])
def test_sparsifier_cli_gpu_equals_host(hip, name):
    """bin/sparsifier -q p -c C on the GPU prints the same change of basis as the host enumeration and the
    factorization is consistent (the reference's own criterion, bin/FDT.sh:64)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    path = os.path.join(DATA, name)
    g = subprocess.run([SPS, "-q", str(P), "-c", "6", "-S", "--gpu-min-rows", "0", path], capture_output=True, text=True, timeout=600)
    h = subprocess.run([SPS, "-q", str(P), "-c", "6", "-S", "--gpu", "0", path], capture_output=True, text=True, timeout=600)
    assert g.returncode == 0 and h.returncode == 0, g.stderr + h.stderr
    assert "SUCCESS: consistent factorization" in g.stderr
    assert g.stdout == h.stdout
    assert re.search(r"with (\d+) non-zeroes", g.stderr).group(1) == re.search(r"with (\d+) non-zeroes", h.stderr).group(1)


@pytest.mark.parametrize("name,b,c", [("4x4x4_49_156_L.sms", 4, 4), ("2x2x2_7_Winograd_L.sms", 4, 6), ("2x2x2_7_DPS-accurate_L.sms", 4, 7), ("3x3x3_23_58_P.sms", 4, 6),
                                      ("4x4x4_48_rational_L.sms", 4, 5), ("3x3x6_40_R.sms", 3, 5)])
def test_sparsifier_cli_on_gpu_prints_the_oracles_basis(hip, name, b, c):
    """bin/sparsifier -q p on the GPU against the ORACLE's restatement of the whole tool (oracle/plo_sparsify_oracle.c: coefficient
    set, seed vector, enumeration through a literal testLinComb, fallback, FactorDiagonals, SparseFactor, sparseLU, blocks): the
    printed change of basis, the residue and the number of candidate rows are the oracle's."""
    from plo_testlib import dense_mod, oracle_sparsify, parse_sms_text
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    path = os.path.join(DATA, name)
    CoB, Res, cand = oracle_sparsify(dense_mod(path, P), P, b, c, True)
    g = subprocess.run([SPS, "-q", str(P), "-b", str(b), "-c", str(c), "-S", "--gpu-min-rows", "0", path], capture_output=True, text=True, timeout=600)
    assert g.returncode == 0 and "SUCCESS: consistent factorization" in g.stderr and re.search(r"# GPU: \d+ launches, enumeration kernels", g.stderr), g.stderr
    assert parse_sms_text(g.stdout) == CoB
    tail = g.stderr.split("residuum profile:")[1]
    assert parse_sms_text(tail[tail.index("\n") + 1:]) == Res
    assert ("# CoB enumeration: %d candidate rows" % cand) in g.stderr


@pytest.mark.parametrize("name,c", [("4x4x4_49_156_L.sms", "4"), ("2x2x2_7_Winograd_L.sms", "6"), ("2x2x2_7_DPS-accurate_L.sms", "6"), ("4x4x4_48_rational_L.sms", "5"),
                                    ("3x3x3_23_58_P.sms", "6")])
def test_sparsifier_over_the_rationals_on_gpu_equals_host_and_oracle(hip, name, c):
    """BASELINE configs[2] as written (`sparsifier -c 4 data/4x4x4_49_156_L.sms`, no -q): over Q, the field the reference runs it in
    (src/sparsifier.cpp:66-83).  The enumeration runs on the GPU modulo two 31-bit primes, every winner is re-evaluated over Q;
    same change of basis and residue as the host enumeration over Q, no (block, row) sent back to the host, consistent factorization."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    path = os.path.join(DATA, name)
    g = subprocess.run([SPS, "-c", c, "-S", "--gpu-min-rows", "0", path], capture_output=True, text=True, timeout=900)
    h = subprocess.run([SPS, "-c", c, "-S", "--gpu", "0", path], capture_output=True, text=True, timeout=900)
    assert g.returncode == 0 and h.returncode == 0, g.stderr + h.stderr
    assert "SUCCESS: consistent factorization" in g.stderr
    m = re.search(r"# GPU \(Q, two 31-bit primes \+ check over Q\): (\d+) enumerations in (\d+) launches \(both moduli of an enumeration in one\), kernels [0-9.e+-]+ ms, (\d+) on the host", g.stderr)
    assert m and int(m.group(1)) > 0 and int(m.group(3)) == 0, g.stderr
    assert int(m.group(2)) <= int(m.group(1))               # plo_cob_search_batch: ONE launch per enumeration (an enumeration after dependent rows needs none)
    assert g.stdout == h.stdout
    assert re.search(r"with (\d+) non-zeroes", g.stderr).group(1) == re.search(r"with (\d+) non-zeroes", h.stderr).group(1)
    # ... and both are what the ORACLE's Q instance gives (oracle/plo_sparsify_oracle.c: the restatement of plinopt_sparsify.inl instantiated
    # with checked rationals): change of basis, residue and the number of candidate rows
    from plo_testlib import dense_q, oracle_sparsify_q, parse_sms_text_q
    CoB, Res, cand = oracle_sparsify_q(dense_q(path), 4, int(c), True)
    assert parse_sms_text_q(g.stdout) == CoB
    tail = g.stderr.split("residuum profile:")[1]
    assert parse_sms_text_q(tail[tail.index("\n") + 1:]) == Res
    assert ("# CoB enumeration: %d candidate rows" % cand) in g.stderr


@pytest.mark.parametrize("b", [8, 16])
def test_fallback_rows_on_gpu_equal_the_oracle(hip, b):
    """`-c 1` (coefficient list {0}: every row after the seed is filled by the canonical fallback, plinopt_sparsify.inl:317-326, and the
    oracle's candidates carry the fallback's coordinate, :305 -- tests/test_sparsify_oracle.py): the GPU enumeration prints the oracle's basis."""
    import ctypes
    from plo_testlib import dense_mod, oracle, oracle_sparsify, parse_sms_text
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    L = oracle()
    L.plo_oracle_sparsify_carried.restype = ctypes.c_uint64
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    CoB, Res, cand = oracle_sparsify(dense_mod(path, P), P, b, 1, True)
    assert L.plo_oracle_sparsify_carried() > 0
    g = subprocess.run([SPS, "-q", str(P), "-b", str(b), "-c", "1", "-S", "--gpu-min-rows", "0", path], capture_output=True, text=True, timeout=600)
    assert g.returncode == 0 and "SUCCESS: consistent factorization" in g.stderr and re.search(r"# GPU: [1-9]\d* launches", g.stderr), g.stderr
    assert parse_sms_text(g.stdout) == CoB
    tail = g.stderr.split("residuum profile:")[1]
    assert parse_sms_text(tail[tail.index("\n") + 1:]) == Res
    assert ("# CoB enumeration: %d candidate rows" % cand) in g.stderr


def test_small_enumerations_stay_on_the_host_by_default(hip):
    """BASELINE configs[2] as written (`sparsifier -c 4`: 256 candidate rows per enumeration) pays nothing on the GPU -- a launch and two
    copies per (block, row) for microseconds of work, and the enumerations of a run are sequential by definition
    (plinopt_sparsify.inl:172-175, 282-314).  By default an enumeration of fewer than 20,000 rows is walked on the host (the tool says
    how many); `-c 12` (20,736 rows) is on the GPU; the result is the same either way."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    d = subprocess.run([SPS, "-c", "4", "-S", path], capture_output=True, text=True, timeout=600)
    f = subprocess.run([SPS, "-c", "4", "-S", "--gpu-min-rows", "0", path], capture_output=True, text=True, timeout=600)
    assert d.returncode == 0 and f.returncode == 0, d.stderr + f.stderr
    m = re.search(r"(\d+) enumerations in (\d+) launches .*; (\d+) enumerations of fewer than 20000 rows on the host", d.stderr)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) == 0 and int(m.group(3)) > 0, d.stderr
    assert d.stdout == f.stdout
    g = subprocess.run([SPS, "-q", str(P), "-c", "12", "-S", os.path.join(DATA, "2x2x2_7_Winograd_L.sms")], capture_output=True, text=True, timeout=600)
    # (SparseFactor starts with 3 coefficients and adds 4 per round, plinopt_sparsify.h:78-80: the rounds below 12 coefficients stay on the host)
    assert g.returncode == 0 and re.search(r"# GPU: [1-9]\d* launches", g.stderr) and re.search(r"; \d+ enumerations of fewer than 20000 rows on the host", g.stderr), g.stderr


@pytest.mark.parametrize("generic", [False, True])
def test_cob_enumeration_sharded_by_prefix_equals_whole(hip, generic, monkeypatch):
    """N-GPU split of ONE enumeration (plo_cob_search_range): shards of (i,j,k) prefixes, then the MAX reduction of
    plinopt_amd.dist.allreduce_cob_best (largest score, smallest index) -- here reduced on the host over 1, 2, 3 and 8
    shards, for the table kernel and for the generic one."""
    from plinopt_amd import cob_search
    from plinopt_amd.dist import shard_range
    if generic:
        monkeypatch.setenv("PLO_COB_GENERIC", "1")
    n, m, TM = _dense_T("4x4x4_49_156_L.sms", P)
    coeffs = [0, 1, P - 1, 2, P - 2, pow(2, -1, P), P - pow(2, -1, P)]
    Cand = [0] * (n * n)
    Cand[0] = 1
    C = len(coeffs)
    for row, w in ((0, (-1, -1)), (1, (-1, -1)), (1, (20, 1))):
        whole, _ = cob_search(n, m, TM, Cand, row, 0, coeffs, P, *w)
        assert whole == oracle_cob_search(n, m, TM, Cand, row, 0, coeffs, P, *w)
        for world in (1, 2, 3, 8):
            best, cand = None, 0
            for r in range(world):
                g0, gn = shard_range(0, C ** 3, r, world)
                res, st = cob_search(n, m, TM, Cand, row, 0, coeffs, P, *w, groups=(g0, gn))
                cand += st["candidates"]
                if res[3]:
                    key = (res[0] * (n + 1) + res[1], -res[2])
                    if best is None or key > best[0]:
                        best = (key, res)
            assert cand == C ** 4
            assert (best[1] if best else (w[0], w[1], 0, 0)) == whole, (row, w, world)
