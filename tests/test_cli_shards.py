"""`--gpu N` (N >= 2) of bin/optimizer and bin/trilplacer: the restarts in N contiguous seed shards -- by default one host thread
and one device per shard inside the library (plo_cse_search_multi, plo_kernel_search_multi, plo_tril_search_multi; the minimum
by RCCL MIN all-reduces), with --fork-shards one forked child per shard and the minimum under the tools' total order in the
parent (plo_host.hpp `forked_shards`).  The winner and the printed
program must be those of the unsharded search (reference: the iterations of `#pragma omp parallel for` are independent,
include/plinopt_optimize.inl:1204-1238, plinopt_inplace.inl:837-924).  CPU: every shard on the host engine
(PLO_SHARD_ENGINE=host); GPU (-m gpu): every shard on device 0 of the box (PLO_GPU_DEVICES=0,0,...)."""
import os
import subprocess

import pytest

from plo_testlib import DATA, ROOT

P = 131071
OPT = os.path.join(ROOT, "bin", "optimizer")
TRIL = os.path.join(ROOT, "bin", "trilplacer")


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    return r.returncode, r.stdout, r.stderr


def _found(err, tag):
    return [ln for ln in err.splitlines() if ln.startswith(tag)]


@pytest.fixture(scope="module", autouse=True)
def _tools():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])


@pytest.mark.parametrize("name,loops", [("4x4x4_49_156_L.sms", 301), ("2x2x2_7_Winograd_L.sms", 7), ("4x4x4_48_rational_P.sms", 64)])
@pytest.mark.parametrize("world", [2, 5])
def test_optimizer_shards_equal_one_process(name, loops, world):
    path = os.path.join(DATA, name)
    rc0, out0, err0 = _run([OPT, "-q", str(P), "-D", "-O", str(loops), "--seed", "11", "--gpu", "0", path])
    rcn, outn, errn = _run([OPT, "-q", str(P), "-D", "-O", str(loops), "--seed", "11", "--gpu", str(world), path], {"PLO_SHARD_ENGINE": "host"})
    assert rc0 == 0 and rcn == 0, err0 + errn
    assert out0 == outn and _found(err0, "# Found D") == _found(errn, "# Found D")
    assert ("# %d shards (host engine): %d candidates" % (world, loops)) in errn


@pytest.mark.parametrize("world", [2, 3])
def test_trilplacer_shards_equal_one_process(world):
    files = [os.path.join(DATA, "4x4x4_49_156" + x) for x in ("_L.sms", "_R.sms", "_P.sms")]
    rc0, out0, err0 = _run([TRIL] + files + ["-O", "150", "--seed", "9", "--gpu", "0"])
    rcn, outn, errn = _run([TRIL] + files + ["-O", "150", "--seed", "9", "--gpu", str(world)], {"PLO_SHARD_ENGINE": "host"})
    assert rc0 == 0 and rcn == 0, err0 + errn
    assert out0 == outn and _found(err0, "# Found") == _found(errn, "# Found")


def test_a_failing_shard_is_an_error():
    """a shard that cannot open its device (no GPU in the CPU test container) fails the run: exit status 2, no program"""
    path = os.path.join(DATA, "2x2x2_7_Winograd_L.sms")
    rc, out, err = _run([OPT, "-q", str(P), "-D", "-O", "8", "--gpu", "2", path], {"PLINOPT_HIP_LIB": "/nonexistent/libplinopt_hip.so", "PLO_GPU_DEVICES": "97,98"})
    assert rc == 2 and "shard failed" in err and out == ""


@pytest.mark.gpu
def test_optimizer_two_shards_on_the_gpu(hip):
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    rc1, out1, err1 = _run([OPT, "-q", str(P), "-D", "-O", "20001", "--seed", "3", path])
    rc2, out2, err2 = _run([OPT, "-q", str(P), "-D", "-O", "20001", "--seed", "3", "--gpu", "2", path], {"PLO_GPU_DEVICES": "0,0"})
    assert rc1 == 0 and rc2 == 0, err1 + err2
    assert out1 == out2 and _found(err1, "# Found D") == _found(err2, "# Found D")
    assert "# 2 shards (one GPU and one host thread each, one process): 20001 candidates" in err2       # plo_cse_search_multi
    rc3, out3, err3 = _run([OPT, "-q", str(P), "-D", "-O", "20001", "--seed", "3", "--gpu", "2", "--fork-shards", path], {"PLO_GPU_DEVICES": "0,0"})
    assert rc3 == 0 and out3 == out1 and "# 2 shards (one forked process and one GPU each): 20001 candidates" in err3, err3   # one forked child per device


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--fork-shards"]])
def test_optimizer_default_methods_with_two_shards(hip, extra):
    """`bin/optimizer -q p --gpu 2 file` with the default methods (-D, -K and -G all on): -D and -K are sharded, nothing is forked
    after this process has touched the HIP runtime (in-library threads by default; with --fork-shards every fork comes first),
    and the result is the one device's."""
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    rc1, out1, err1 = _run([OPT, "-q", str(P), "-O", "2001", "--seed", "3", path])
    rc2, out2, err2 = _run([OPT, "-q", str(P), "-O", "2001", "--seed", "3", "--gpu", "2"] + extra + [path], {"PLO_GPU_DEVICES": "0,0"})
    assert rc1 == 0 and rc2 == 0, err1 + err2
    assert out1 == out2
    for tag in ("# Found D", "# Found K", "# Found G"):
        assert _found(err1, tag) == _found(err2, tag) and _found(err1, tag), tag
    assert "2 shards" in err2 and "-K" in err2
    assert ("one forked process" in err2) == bool(extra)


@pytest.mark.gpu
def test_in_library_kernel_and_tril_searches_over_devices(hip, monkeypatch):
    """plo_kernel_search_multi and plo_tril_search_multi: shards on device 0 (1, 2, 3 threads) give the single-device winner;
    with PLO_MULTI_REDUCE=rccl a one-rank communicator runs the two MIN all-reduces, is built ONCE for the device set and reused."""
    from plinopt_amd import TrilPlan, capi, kernel_search, kernel_search_multi, tril_search_multi
    from plo_testlib import OracleMatrix, OracleTril
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    one = kernel_search((M.m, M.n, M.rowptr, M.col, M.val), P, 5, 2000, want_costs=False)[3]
    for devs in ([0], [0, 0], [0, 0, 0]):
        got, st = kernel_search_multi((M.m, M.n, M.rowptr, M.col, M.val), P, 5, 2000, devs)
        assert got == one and st["candidates"] == 2000 and st["reduce"] == 0
    O = OracleTril.from_sms(*(os.path.join(DATA, "4x4x4_49_156" + x) for x in ("_L.sms", "_R.sms", "_P.sms")))
    mats = [(n, rp, col, [int(x) for x in num]) for n, (rp, col, num, den) in zip(O.dims, O.csr)]
    G = TrilPlan(O.m, mats)
    tone = G.search(11, 1500)
    for devs in ([0], [0, 0], [0, 0, 0]):
        got, st = tril_search_multi(O.m, mats, 11, 1500, devs)
        assert got == tone and st["candidates"] == 1500
    monkeypatch.setenv("PLO_MULTI_REDUCE", "rccl")
    n0 = capi.lib().plo_multi_comm_inits()
    got, st = tril_search_multi(O.m, mats, 11, 1500, [0])
    assert got == tone and st["reduce"] == 1 and st["reduce_seconds"] > 0
    got, st = kernel_search_multi((M.m, M.n, M.rowptr, M.col, M.val), P, 5, 2000, [0])
    assert got == one and st["reduce"] == 1
    assert capi.lib().plo_multi_comm_inits() - n0 <= 1            # one communicator for the device set {0}, reused by the second call
    assert G.search(11, 1500) == tone                              # the caller's device and context are untouched


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_in_library_multi_device_search_equals_one_device(hip, ndev):
    """plo_cse_search_multi: one process, ndev host threads, one device each (all device 0 on the test box), contiguous seed
    blocks, minimum under (cmpOpCount key, seed) -- the same winner as the search on one device and as the oracle."""
    from plinopt_amd import CSEPlan, cse_search_multi
    from plo_testlib import OracleMatrix
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    for mode in (0, 1, 2):
        got, st = cse_search_multi(M.m, M.n, M.rowptr, M.col, M.val, P, 7, 3001, [0] * ndev, cost_mode=mode)
        assert got == M.search(7, 3001, cost_mode=mode, nthreads=8)
        assert st["candidates"] == 3001
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, P)          # the process-wide context still works after the threads' own
    assert plan.search(7, 3001) == cse_search_multi(M.m, M.n, M.rowptr, M.col, M.val, P, 7, 3001, [0, 0])[0]


@pytest.mark.gpu
def test_multi_device_minimum_through_rccl(hip, monkeypatch):
    """north_star: "a single RCCL MIN all-reduce over xGMI".  plo_cse_search_multi reduces the shards' packed words
    key << 32 | (seed - seed0) with ONE ncclAllReduce(ncclUint64, ncclMin) over a communicator of its devices (librccl loaded at
    run time) and checks the result against the host minimum; the test box has one GPU, so the communicator has one rank here
    (PLO_MULTI_REDUCE=rccl asks for the all-reduce even then) -- same winner as the single-device search, stats say `reduce`."""
    from plinopt_amd import CSEPlan, cse_search_multi
    from plo_testlib import OracleMatrix
    M = OracleMatrix.from_sms(os.path.join(DATA, "4x4x4_49_156_L.sms"), P)
    monkeypatch.setenv("PLO_MULTI_REDUCE", "rccl")
    got, st = cse_search_multi(M.m, M.n, M.rowptr, M.col, M.val, P, 7, 3001, [0])
    assert st["reduce"] == 1
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, P)
    assert plan.search(7, 3001) == got
    monkeypatch.setenv("PLO_MULTI_REDUCE", "host")
    got2, st2 = cse_search_multi(M.m, M.n, M.rowptr, M.col, M.val, P, 7, 3001, [0])
    assert got2 == got and st2["reduce"] == 0


@pytest.mark.gpu
def test_trilplacer_two_shards_on_the_gpu(hip):
    files = [os.path.join(DATA, "4x4x4_49_156" + x) for x in ("_L.sms", "_R.sms", "_P.sms")]
    rc1, out1, err1 = _run([TRIL] + files + ["-O", "3001", "--seed", "9"])
    rc2, out2, err2 = _run([TRIL] + files + ["-O", "3001", "--seed", "9", "--gpu", "2"], {"PLO_GPU_DEVICES": "0,0"})
    assert rc1 == 0 and rc2 == 0, err1 + err2
    assert out1 == out2 and _found(err1, "# Found") == _found(err2, "# Found")
    assert "# 2 shards (one GPU and one host thread each, one process)" in err2                           # plo_tril_search_multi
    rc3, out3, err3 = _run([TRIL] + files + ["-O", "3001", "--seed", "9", "--gpu", "2", "--fork-shards"], {"PLO_GPU_DEVICES": "0,0"})
    assert rc3 == 0 and out3 == out1 and "one forked process" in err3, err3
