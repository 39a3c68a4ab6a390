"""End to end through the reference CLI contract on the GPU: BASELINE configs[1]
`bin/optimizer -q 131071 -D -O 1000000 data/2x2x2_7_Winograd_L.sms`, then the reference's own
verification pipeline (`| SLPchecker -q -M`, bin/FDT.sh:58-60)."""
import os
import re
import subprocess

import pytest

from plo_testlib import DATA, ROOT, OracleMatrix

pytestmark = pytest.mark.gpu
OPT = os.path.join(ROOT, "bin", "optimizer")
CHK = os.path.join(ROOT, "bin", "SLPchecker")
P = 131071


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])


def run(cmd, stdin=None):
    r = subprocess.run(cmd, input=stdin, capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("name,loops,bound", [
    ("2x2x2_7_Winograd_L.sms", 1000000, 4),          # configs[1]: 10^6 restarts, known optimum 4 adds
    ("4x4x4_49_156_L.sms", 300000, None),
    ("2x2x2_7_DPS-accurate_L.sms", 50000, None),     # rational coefficients: multiplications counted
])
def test_optimizer_cli_on_gpu(name, loops, bound):
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "-D", "-O", str(loops), path])
    assert rc == 0, err
    m = re.search(r"# Found D: (\d+)\|(\d+) instead of (\d+)\|(\d+)\t\[seed (\d+)\]", err)
    assert m, err
    a, mu, seed = int(m.group(1)), int(m.group(2)), int(m.group(5))
    if bound is not None:
        assert (a, mu) == (bound, 0)
    M = OracleMatrix.from_sms(path, P)
    oa, om, otxt = M.optimizer(seed)
    assert (oa, om) == (a, mu)
    assert out == otxt                                  # identical emitted SLP for the winning seed
    sample = min(loops, 20000)                          # the winner is at least as good as any sampled seed
    sa, sm, sseed = M.search(0, sample, nthreads=8)
    assert (a + mu, a) <= (sa + sm, sa)
    if (a, mu) == (sa, sm):
        assert seed <= sseed
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


@pytest.mark.parametrize("name", ["2x2x2_7_DPS-accurate_L.sms", "4x4x4_49_156_L.sms", "4x4x4_48_rational_P.sms"])
def test_lu_method_on_gpu_equals_host_search(name):
    """-G on the GPU (chained candidates: Optimizer on U then on L, one stream) returns the same winner, the same
    program as the host loop (--gpu 0), and the program verifies."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "-G", "-O", "3000", path])
    assert rc == 0, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "-G", "-O", "3000", "--gpu", "0", path])
    assert rc0 == 0, err0
    g = re.search(r"# Found G: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\]", err)
    g0 = re.search(r"# Found G: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\]", err0)
    assert g and g0 and g.groups() == g0.groups(), (err, err0)
    assert out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def test_trilplacer_cli_on_gpu():
    """bin/trilplacer: restart loop on the GPU (plo_tril_search), winner replayed on the host; the program runs in
    place and the reported (ADD, SCA) are those of the oracle's argmin over the same seeds."""
    import random
    from plo_testlib import TRIL_BASE_SEED, OracleTril
    from test_tril_oracle import check_program
    name = "4x4x4_49_156"
    files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
    T = OracleTril.from_sms(*files)
    n = 400
    r = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-O", str(n), "--seed", "1000"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "restarts on GPU" in r.stderr
    ops, seed, var = T.search(1000, n)
    (base, _), = T.cost_many(seeds=[TRIL_BASE_SEED])
    want = ops if (ops[0], ops[1]) < (base[0], base[1]) else base
    got = tuple(int(x) for x in re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", r.stderr))
    assert got == want, (got, want, r.stderr)
    check_program(T, r.stdout, want, random.Random(5))


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms"])
def test_ab_method_on_gpu_equals_host_search(name):
    """-A: the factorization M = Alt.CoB is made on the host, the restart loop (Optimizer on CoB, then on Alt, one
    stream) runs on the GPU through the chained-candidate kernel; same winner and program as the host loop."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "--only", "A", "-O", "3000", path])
    assert rc == 0, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "A", "-O", "3000", "--gpu", "0", path])
    assert rc0 == 0, err0
    pat = r"# Found A: \(\d+x\d+x\d+ \d+/\d+\)\t(\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\]"
    g, g0 = re.search(pat, err), re.search(pat, err0)
    assert g and g0 and g.groups() == g0.groups(), (err, err0)
    assert "# GPU (A):" in err and out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "2x2x2_7_Winograd_L.sms"])
def test_kernel_method_on_gpu_equals_host_search(name):
    """-K: one nullspace decomposition per restart (reference include/plinopt_optimize.inl:1299-1340), made on the host; the
    two Optimizer calls of every restart on the GPU (batched chained-candidate kernel: all decompositions of a batch in one
    launch, matrices with emptied rows); same winner and program as the host loop.  --kernel-block 16 shares one
    decomposition between 16 restarts."""
    path = os.path.join(DATA, name)
    for extra, ndec in (([], 2000), (["--kernel-block", "16"], 125)):
        rc, out, err = run([OPT, "-q", str(P), "--only", "K", "-O", "2000", path] + extra)
        assert rc == 0, err
        rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "K", "-O", "2000", "--gpu", "0", path] + extra)
        assert rc0 == 0, err0
        pat = r"# Found K: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\+(\d+), (\d+) dependent rows\)"
        g, g0 = re.search(pat, err), re.search(pat, err0)
        assert g and g0 and g.groups() == g0.groups(), (err, err0)
        assert ("# GPU (K): 2000 candidates on %d decompositions" % ndec) in err and out == out0
        rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
        assert rc == 0 and "SUCCESS" in err2, err2


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "2x2x2_7_Winograd_P.sms", "2x2x2_7_DPS-accurate_L.sms"])
def test_exhaustive_method_on_gpu_equals_host(name):
    """-E: the schedules of RecSub's tree are candidates of the wave kernel (enumeration pick mode); same tree size,
    same best schedule and program as the host walk."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "--only", "E", path])
    assert rc == 0, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "E", "--gpu", "0", path])
    assert rc0 == 0, err0
    pat = r"# Found E: (\d+)\|(\d+) instead of \d+\|\d+\t\[schedule (\d+)\] \((\d+) schedules: the whole tree"
    g, g0 = re.search(pat, err), re.search(pat, err0)
    assert g and g0 and g.groups() == g0.groups() and "GPU" in err, (err, err0)
    assert out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def test_exhaustive_method_is_bounded_on_a_large_tree():
    """cyclic.sms has a tree of more than 8e8 schedules: -E stops at its budget (2^22 by default), says so, and still
    prints a verified program."""
    path = os.path.join(DATA, "cyclic.sms")
    rc, out, err = run([OPT, "-q", str(P), "--only", "E", path])
    assert rc == 0, err
    g = re.search(r"# Found E: (\d+)\|(\d+) instead of 26\|0\t\[schedule (\d+)\] \(4194304 schedules of a tree of at least (\d+), GPU", err)
    assert g and int(g.group(4)) > 4194304 and int(g.group(1)) <= 15, err
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


@pytest.mark.parametrize("name", ["2x2x2_7_DPS-accurate_L.sms"])
def test_all_row_orders_on_gpu_equals_host(name):
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "--only", "N", "-O", "4000", path])
    assert rc == 0, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "N", "-O", "4000", "--gpu", "0", path])
    assert rc0 == 0, err0
    pat = r"# Found N: (\d+)\|(\d+) instead of \d+\|\d+\t\[order (\d+), seed (\d+)\] \((\d+) row orders, (\d+) distinct decompositions, (\d+) restarts each"
    g, g0 = re.search(pat, err), re.search(pat, err0)
    assert g and g0 and g.groups() == g0.groups() and "GPU kernel" in err, (err, err0)
    assert out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms"])
def test_kernel_method_on_gpu_equals_the_oracle_restatement(name):
    """-K on the GPU (batched chained-candidate kernel, one decomposition per restart) against the ORACLE: the winner, its
    counts and its decomposition (rank, NotIndep, dependent rows) are those of oracle/plo_oracle.c
    `plo_oracle_kernel_restart` minimised over the same seeds."""
    from plo_testlib import OracleMatrix
    from test_host_tools import kernel_oracle_argmin
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 300
    (a, mu, rank, ni, nd), seed = kernel_oracle_argmin(M, 7, n)
    rc, out, err = run([OPT, "-q", str(P), "--only", "K", "-O", str(n), "--seed", "7", path])
    assert rc == 0 and "# GPU (K): %d candidates on %d decompositions" % (n, n) in err, err
    g = re.search(r"# Found K: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\+(\d+), (\d+) dependent rows\)", err)
    assert g and tuple(int(x) for x in g.groups()) == (a, mu, seed, rank, ni, nd), (g and g.groups(), (a, mu, seed, rank, ni, nd))


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "cyclic.sms"])
def test_lu_method_on_gpu_equals_the_oracle_restatement(name):
    """-G on the GPU (chained-candidate kernel on U and L) against the ORACLE: its own dense restatement of the LU rule, then its chained
    Optimizer, minimised over the same seeds"""
    from test_host_tools import lu_oracle_argmin
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 200
    a, mu, seed, rank = lu_oracle_argmin(M, 9, n)
    rc, out, err = run([OPT, "-q", str(P), "--only", "G", "-O", str(n), "--seed", "9", path])
    assert rc == 0 and "GPU" in err, err
    g = re.search(r"# Found G: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\)", err)
    assert g and tuple(int(x) for x in g.groups()) == (a, mu, seed, rank), (g and g.groups(), (a, mu, seed, rank))


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "3x3x3_23_58_L.sms"])
def test_ab_method_on_gpu_equals_the_oracle_restatement(name):
    """-A on the GPU (chained-candidate kernel on CoB and Alt), first inner dimension, against the ORACLE: its own restatement of the
    back-solver rule (1 + N/8 back-solves), then its chained Optimizer minimised over the same N seeds"""
    from test_host_tools import AB_PAT, ab_oracle_argmin
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 160
    exp = ab_oracle_argmin(M, 21, n)
    rc, out, err = run([OPT, "-q", str(P), "--only", "A", "-O", str(n), "--seed", "21", path])
    assert rc == 0 and "# GPU (A):" in err, err
    g = re.search(AB_PAT, err)
    got = tuple(int(x) for x in g.groups())
    assert (got[:5], got[5], got[6], got[7]) == exp, (got, exp)


@pytest.mark.parametrize("name", ["2x2x2_7_DPS-accurate_L.sms", "2x2x2_7_Winograd_L.sms"])
def test_all_row_orders_on_gpu_equal_the_oracle(name):
    """-N on the GPU (all distinct decompositions in one launch of the batched chain kernel) against the ORACLE's decompositions with
    prescribed orders: distinct decompositions, restarts each, winner (counts, order, seed)"""
    from test_host_tools import N_PAT, allkernels_oracle
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    exp = allkernels_oracle(M, 29, 4000)
    rc, out, err = run([OPT, "-q", str(P), "--only", "N", "-O", "4000", "--seed", "29", path])
    assert rc == 0 and "GPU kernel" in err, err
    g = re.search(N_PAT, err)
    assert g and tuple(int(x) for x in g.groups()) == exp, (g and g.groups(), exp)


def test_kernel_method_sharded_over_devices_equals_one_device():
    """-K --gpu 3: the restart range in three shards (here all on device 0: PLO_GPU_DEVICES), every shard = plo_kernel_search on its
    block -- one host thread per shard inside the library by default (plo_kernel_search_multi), one forked child per shard with
    --fork-shards; same winner, counts, decomposition and program as one device."""
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    env = dict(os.environ, PLO_GPU_DEVICES="0,0,0")
    r3 = subprocess.run([OPT, "-q", str(P), "--only", "K", "-O", "3001", "--seed", "11", "--gpu", "3", path], capture_output=True, text=True, timeout=600, env=env)
    rc1, out1, err1 = run([OPT, "-q", str(P), "--only", "K", "-O", "3001", "--seed", "11", path])
    assert r3.returncode == 0 and rc1 == 0, r3.stderr + err1
    pat = r"# Found K: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\+(\d+), (\d+) dependent rows\)"
    g3, g1 = re.search(pat, r3.stderr), re.search(pat, err1)
    assert g3 and g1 and g3.groups() == g1.groups(), (r3.stderr, err1)
    assert "# 3 shards (one GPU and one host thread each, one process, minimum on the host): -K" in r3.stderr and "# GPU (K): 3001 candidates on 3001 decompositions" in r3.stderr
    assert r3.stdout == out1
    rf = subprocess.run([OPT, "-q", str(P), "--only", "K", "-O", "3001", "--seed", "11", "--gpu", "3", "--fork-shards", path], capture_output=True, text=True, timeout=600, env=env)
    assert rf.returncode == 0 and rf.stdout == out1 and "# 3 shards (one forked process and one GPU each): -K" in rf.stderr, rf.stderr


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "2x2x2_7_DPS-accurate_L.sms", "4x4x4_48_rational_L.sms"])
def test_kernel_method_with_identity_goals_on_gpu(name):
    """-K -F: the restarts on [M ; I] run on the GPU like -K; same result as the host loop, program verifies."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(P), "--only", "K", "-F", "-O", "1000", path])
    assert rc == 0 and "# GPU (K): 1000 candidates" in err, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "K", "-F", "-O", "1000", "--gpu", "0", path])
    assert rc0 == 0 and out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def _dense_distinct(path, rows, cols, seed=5):
    import random
    rng = random.Random(seed)
    with open(path, "w") as f:
        f.write("%d %d M\n" % (rows, cols))
        for i in range(rows):
            for j in range(cols):
                if rng.random() < 0.9:
                    f.write("%d %d %d\n" % (i + 1, j + 1, rng.randint(2, 60000)))
        f.write("0 0 0\n")


def test_program_gen_with_more_than_64_rows_in_a_column(tmp_path):
    """100 rows whose 40 columns are almost full of pairwise different coefficients: after the CSE steps a column still holds a non +-1
    entry in more than 64 rows.  Until round 3 ProgramGen's Triangle on the device kept a column's rows one per lane and refused
    (`PLO_E_UNSUPPORTED`, found by tests/soak_hbm.py: 5 % of its random matrices); round 4 keeps them in four registers per lane
    (256 rows): the restarts stay on the GPU, counts per seed equal the ORACLE's (plinopt_optimize.inl:427-507 literally), the tool
    prints the program of `--gpu 0` and it verifies."""
    from plinopt_amd import CSEPlan
    from plo_testlib import OracleMatrix
    path = tmp_path / "dense.sms"
    _dense_distinct(path, 100, 40)
    M = OracleMatrix.from_sms(str(path), P)
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p)
    assert plan.is_hbm
    assert plan.cost_many(seed0=3, n=6) == tuple(M.cost_many(seed0=3, nseeds=6, nthreads=6))
    plan.close()
    rc, out, err = run([OPT, "-q", str(P), "--only", "D", "-O", "6", "--seed", "3", str(path)])
    assert rc == 0 and "refused" not in err and "# GPU: 6 candidates" in err, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "D", "-O", "6", "--seed", "3", "--gpu", "0", str(path)])
    assert rc0 == 0 and out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", str(path)], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def test_direct_method_refused_by_the_device_runs_on_the_host(tmp_path):
    """The same kind of matrix with 300 rows: more than 256 rows keep a non +-1 entry in one column, which ProgramGen's Triangle on the
    device does not take -- `PLO_E_UNSUPPORTED`, not an internal error.  `bin/optimizer -D` says so and runs the same restarts on the host:
    same program as `--gpu 0`, and it verifies."""
    path = tmp_path / "dense.sms"
    _dense_distinct(path, 300, 24)
    rc, out, err = run([OPT, "-q", str(P), "--only", "D", "-O", "3", "--seed", "3", str(path)])
    assert rc == 0 and "# -D on the GPU refused (" in err and "host search" in err, err
    rc0, out0, err0 = run([OPT, "-q", str(P), "--only", "D", "-O", "3", "--seed", "3", "--gpu", "0", str(path)])
    assert rc0 == 0 and out == out0
    rc, _, err2 = run([CHK, "-q", str(P), "-M", str(path)], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2

