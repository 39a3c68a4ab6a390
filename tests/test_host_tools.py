"""Host tools (C++): bin/optimizer keeps the reference CLI contract (stdout = SLP only,
'#' lines on stderr), bin/SLPchecker verifies programs; the host replay of a seed is
text-identical to the CPU oracle's program for the same seed."""
import glob
import os
import re
import subprocess

import pytest

from plo_testlib import DATA, ROOT, OracleMatrix

OPT = os.path.join(ROOT, "bin", "optimizer")
CHK = os.path.join(ROOT, "bin", "SLPchecker")
P = 131071
ALL = sorted(os.path.basename(f) for f in glob.glob(os.path.join(DATA, "*.sms")) if "-X_" not in f)


@pytest.fixture(scope="session", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])


def run(cmd, stdin=None):
    r = subprocess.run(cmd, input=stdin, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("name", ["cyclic.sms", "2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "2x2x2_7_DPS-accurate_L.sms",
                                  "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms", "2o2o4_5_Toom3_P.sms"])
def test_config1_plumbing_over_Q(name):
    """BASELINE configs[0]: bin/optimizer data/cyclic.sms, single CPU pass; then the
    reference's own check `optimizer | SLPchecker -M` (bin/FDT.sh:58)."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, path, "-O", "10"])
    assert rc == 0, err
    assert all(re.match(r"^[a-z]\d+:=", ln) for ln in out.strip().splitlines())         # stdout is SLP only
    assert all(ln.startswith("#") for ln in err.strip().splitlines())                    # stderr is '#' lines
    m = re.search(r"# \S*?(\d+)\tadditions\tinstead of (\d+)", err)
    assert m and int(m.group(1)) <= int(m.group(2))
    rc, _, err2 = run([CHK, "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2
    # the count printed by the checker (lineOperations) equals the count printed by the optimizer
    assert int(re.search(r"# \S*?(\d+)\tadditions", err2).group(1)) == int(m.group(1))


@pytest.mark.parametrize("name", ALL)
def test_host_replay_text_equals_oracle_text(name):
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    for seed in (0, 3, 12345678901234):
        a, mu, txt = M.optimizer(seed)
        rc, out, err = run([OPT, "-q", str(P), "--replay", "--seed", str(seed), path])
        assert rc == 0, err
        assert out == txt
        assert "# %d\tadditions" % a in err and "# %d\tmultiplications" % mu in err


def test_mod_p_host_search_matches_oracle_and_verifies():
    path = os.path.join(DATA, "4x4x4_49_156_L.sms")
    M = OracleMatrix.from_sms(path, P)
    rc, out, err = run([OPT, "-q", str(P), "-D", "-O", "300", "--gpu", "0", "--seed", "42", path])
    assert rc == 0, err
    a, mu, seed = M.search(42, 300)
    assert "# Found D: %d|%d" % (a, mu) in err and "[seed %d]" % seed in err
    assert out == M.optimizer(seed)[2]
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2


def test_checker_rejects_a_wrong_program():
    path = os.path.join(DATA, "2x2x2_7_Winograd_L.sms")
    good = open(os.path.join(DATA, "2x2x2_7_Winograd_L.slp")).read()
    assert run([CHK, "-M", path], stdin=good)[0] == 0
    bad = good.replace("o1:=i3+o0;", "o1:=i3-o0;")
    rc, _, err = run([CHK, "-M", path], stdin=bad)
    assert rc == 1 and "ERROR" in err


def test_regenerates_32x32x32_matrix_from_its_slp():
    """data/32x32x32_15096_L.sms is absent upstream (.MISSING_LARGE_BLOBS); the reference rule
    (Makefile:79-80) rebuilds it from the stored SLP.  Sizes: SURVEY.md 8a."""
    rc, out, err = run([CHK, os.path.join(DATA, "32x32x32_15096_L.slp")])
    assert rc == 0
    lines = out.splitlines()
    assert lines[0].split()[:2] == ["15096", "1024"]
    assert len(lines) - 2 == 1257376
    assert "34750\tadditions" in err
    vals = {}
    for ln in lines[1:-1]:
        v = ln.split()[2]
        vals[v] = vals.get(v, 0) + 1
    assert vals["1/17"] + vals["-1/17"] == 1052576


def test_stored_slps_verify_against_their_matrices():
    """GDT-style sanity on the data pairs (data/Makefile:37-38)."""
    n = 0
    for slp in sorted(glob.glob(os.path.join(DATA, "*.slp"))):
        sms = slp[:-4] + ".sms"
        if "32x32x32" in slp or "-X_" in slp or not os.path.exists(sms):
            continue
        rc, _, err = run([CHK, "-M", sms, slp])
        if rc != 0:      # algorithms over F_32 / F_243 (characteristic 2 / 3) only verify modulo their characteristic
            q = "2" if ("F32" in slp or "S32" in slp) else "3" if ("F243" in slp or "S243" in slp) else None
            assert q, (slp, err)
            rc, _, err = run([CHK, "-q", q, "-M", sms, slp])
        assert rc == 0 and "SUCCESS" in err, (slp, err)
        n += 1
    assert n >= 50


@pytest.mark.parametrize("engine", ["literal", "fast"])
def test_both_host_engines_on_synthetic_triangle_cases(engine, tmp_path):
    """The scalable engine (plo_fast.hpp) and the literal replay must print the oracle's text on
    matrices that exercise FactorOutColumns/Rows and Triangle (incl. its never-reset `found`)."""
    import synth
    for s in range(120):
        m, n, rows = synth.small_valued(s, P)
        rp, c, v = synth.to_csr(rows, P)
        M = OracleMatrix(m, n, rp, c, v, P)
        path = tmp_path / ("s%d.sms" % s)
        with open(path, "w") as fh:
            fh.write("%d %d M\n" % (m, n))
            for i in range(m):
                for k in range(rp[i], rp[i + 1]):
                    fh.write("%d %d %d\n" % (i + 1, c[k] + 1, v[k]))
            fh.write("0 0 0\n")
        a, mu, txt = M.optimizer(s)
        rc, out, err = run([OPT, "-q", str(P), "--replay", "--seed", str(s), "--engine", engine, str(path)])
        assert rc == 0, err
        assert out == txt, (s, rows)


def test_config5_host_engine_program_verifies(tmp_path):
    """32x32x32_15096_L end to end on the host: the scalable engine's program for seed 6 computes the
    matrix (SLPchecker) and its cost equals the committed golden value (~25 s)."""
    import json
    from plo_testlib import GOLDEN
    sms = tmp_path / "l32.sms"
    rc, out, err = run([CHK, "-q", str(P), os.path.join(DATA, "32x32x32_15096_L.slp")])
    assert rc == 0
    sms.write_text(out)
    rc, prog, err = run([OPT, "-q", str(P), "--replay", "--seed", "6", str(sms)])
    assert rc == 0, err
    G = json.load(open(os.path.join(GOLDEN, "config5_costs.json")))
    a, mu = G["costs"]["6"]
    assert "# %d\tadditions" % a in err and "# %d\tmultiplications" % mu in err
    rc, _, err2 = run([CHK, "-q", str(P), "-M", str(sms)], stdin=prog)
    assert rc == 0 and "SUCCESS" in err2 and ("%d,%d" % (a, mu)) in err2, err2


@pytest.mark.parametrize("name", ["cyclic.sms", "2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "2x2x2_7_DPS-accurate_L.sms",
                                  "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms", "3x4x7_63_rational-ALT_L.sms", "4o4o4_F32_Standard_L.sms"])
@pytest.mark.parametrize("field", ["Q", "p"])
def test_lu_method_programs_verify(name, field):
    """-G (LUOptimiser, plinopt_optimize.inl:1021-1109) on the host: factor with the build's pivot rule, optimise U
    and L with one random stream, glue with the two permutations; the program must compute the matrix and the
    printed count must equal the checker's count whenever -G wins."""
    path = os.path.join(DATA, name)
    q = ["-q", str(P), "--gpu", "0"] if field == "p" else []
    rc, out, err = run([OPT, "-G", "-O", "12"] + q + [path])
    assert rc == 0, err
    assert "# Found G:" in err
    rc, _, err2 = run([CHK] + (["-q", str(P)] if field == "p" else []) + ["-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2
    m = re.search(r"# \S*?(\d+)\tadditions\tinstead of (\d+)", err)
    final_adds = int(m.group(1)) if m else 0          # the statistics lines are omitted for a 0|0 program (src/optimizer.cpp:89)
    assert int(re.search(r"# \S*?(\d+)\tadditions", err2).group(1)) == final_adds


SPS = os.path.join(ROOT, "bin", "sparsifier")


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "4x4x4_49_156_P.sms", "2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms",
                                  "3x3x3_23_58_R.sms", "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms", "cyclic.sms"])
@pytest.mark.parametrize("args", [["--gpu", "0"], ["-q", "7", "--gpu", "0"], ["-q", str(P), "--gpu", "0", "-b", "1"], ["-U", "0", "--host-q"]])   # host loops (over Q the GPU is the default)
def test_sparsifier_factorization_is_consistent(name, args):
    """bin/FDT.sh:64-66: `sparsifier [-q 7] -c 5 file` must print SUCCESS (M == Res . CoB); the residue is never
    denser than the input; stdout carries the change of basis only."""
    path = os.path.join(DATA, name)
    rc, out, err = run([SPS, "-c", "5", "-S"] + args + [path])
    assert rc == 0, err
    assert "SUCCESS: consistent factorization" in err and "ERROR" not in err
    mm = re.search(r"with (\d+) non-zeroes \((\d+) alt\.\) instead of (\d+)", err)
    assert mm and int(mm.group(1)) <= int(mm.group(3))
    lines = out.strip().splitlines()
    n = int(lines[0].split()[0])
    assert lines[0].split()[1] == str(n) and lines[-1] == "0 0 0"          # an n x n matrix in SMS format


def test_sparsifier_over_q_needs_the_gpu_unless_told_otherwise():
    """no silent fallback: without a device `sparsifier file` (rationals, GPU enumeration by default) stops with an error, `--gpu 0` and
    `--host-q` run the host loops"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    path = os.path.join(DATA, "2x2x2_7_Winograd_L.sms")
    rc, out, err = run([SPS, "-c", "4", "-S", path])
    assert rc != 0 and "cannot use the GPU" in err and out == ""
    for flag in (["--gpu", "0"], ["--host-q"]):
        rc, out, err = run([SPS, "-c", "4", "-S"] + flag + [path])
        assert rc == 0 and "SUCCESS" in err


def test_sparsifier_finds_the_known_sparse_basis_of_winograd():
    """Winograd's L matrix (14 non-zeros) has an alternative basis with 10 (data/2x2x2_7_DPS-accurate-ALT_L.sms)."""
    rc, out, err = run([SPS, "-c", "4", "-S", "--gpu", "0", os.path.join(DATA, "2x2x2_7_Winograd_L.sms")])
    assert rc == 0 and re.search(r"with (\d+) non-zeroes", err).group(1) == "10"


# ----------------------------------------------------------------------------- trilplacer (host loop)
TRIL_CASES = ["2x2x2_7_Winograd", "4x4x4_49_156", "1o1o2_3_Karatsuba", "2x2x2_7_DPS-accurate"]


@pytest.mark.parametrize("name", TRIL_CASES)
def test_trilplacer_host_search_equals_oracle_and_program_runs_in_place(name):
    """bin/trilplacer --gpu 0: the host restart loop picks the oracle's argmin, and the printed program, run by the
    independent in-place interpreter, adds the bilinear map to c and restores a and b."""
    import random
    from plo_testlib import TRIL_BASE_SEED, OracleTril
    from test_tril_oracle import check_program
    files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
    if not all(os.path.exists(f) for f in files):
        pytest.skip("fixture triple absent")
    T = OracleTril.from_sms(*files)
    n = 60
    r = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "--gpu", "0", "-O", str(n), "--seed", "7"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ops, seed, var = T.search(7, n)
    (base, _), = T.cost_many(seeds=[TRIL_BASE_SEED])
    want = ops if (ops[0], ops[1]) < (base[0], base[1]) else base
    got = tuple(int(x) for x in re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", r.stderr))
    assert got == want, (got, want, r.stderr)
    check_program(T, r.stdout, want, random.Random(11))


@pytest.mark.parametrize("name", TRIL_CASES)
def test_trilplacer_expanded_host_search_equals_oracle(name):
    """bin/trilplacer -e --gpu 0 (README: `trilplacer data/1o1o2_3_Karatsuba_{L,R,P}.sms -e`): the host loop picks the
    oracle's argmin of the expanded programs, prints the oracle's text for that (seed, variant), and the program passes
    the double-size in-place check."""
    import random
    from plo_testlib import TRIL_BASE_SEED, OracleTril
    from test_tril_oracle import check_expanded_program
    files = [os.path.join(DATA, name + s) for s in ("_L.sms", "_R.sms", "_P.sms")]
    T = OracleTril.from_sms(*files)
    n = 40
    r = subprocess.run([os.path.join(ROOT, "bin", "trilplacer"), "-e", "--gpu", "0", "-O", str(n), "--seed", "7"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ops, seed, var = T.search(7, n, expanded=True)
    (base, _), = T.cost_many(seeds=[TRIL_BASE_SEED], expanded=True)
    better = (ops[0], ops[1]) < (base[0], base[1])
    want = ops if better else base
    got = tuple(int(x) for x in re.findall(r"(\d+)\t(?:ADD|SCA|AXPY)", r.stderr))
    assert got == want and "AXPY (double size)" in r.stderr, (got, want, r.stderr)
    assert r.stdout == T.program(seed if better else TRIL_BASE_SEED, var if better else 0, expanded=True)[1]
    check_expanded_program(T, r.stdout, want, random.Random(13))


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "4x4x4_49_156_R.sms", "3x3x6_40_L.sms",
                                  "4x4x4_48_rational_L.sms", "2x2x2_7_DPS-accurate_L.sms", "cyclic.sms"])
@pytest.mark.parametrize("field", ["Q", "p"])
def test_ab_method_programs_verify(name, field):
    """-A (ABOptimiser, plinopt_optimize.inl:1114-1186, inner dimension = column count): M = Alt.CoB by the back-solver
    with a random row order, then Optimizer on CoB and on Alt with one random stream; `--only A` prints that program,
    which must compute the matrix with exactly the reported operation count."""
    path = os.path.join(DATA, name)
    q = ["-q", str(P), "--gpu", "0"] if field == "p" else []
    rc, out, err = run([OPT, "--only", "A", "-O", "24"] + q + [path])
    assert rc == 0, err
    g = re.search(r"# Found A: \((\d+)x(\d+)x(\d+) \d+/\d+\)\t(\d+)\|(\d+) instead of", err)
    assert g, err
    rc, _, err2 = run([CHK] + (["-q", str(P)] if field == "p" else []) + ["-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2
    m = re.search(r"# \S*?(\d+)\tadditions\tinstead of (\d+)", err)
    final_adds = int(m.group(1)) if m else 0
    assert int(re.search(r"# \S*?(\d+)\tadditions", err2).group(1)) == final_adds


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "4x4x4_49_156_R.sms", "3x3x6_40_L.sms",
                                  "4x4x4_48_rational_L.sms", "2x2x2_7_DPS-accurate_L.sms", "6x3x3_40_DPS-accurate_R.sms"])
@pytest.mark.parametrize("field", ["Q", "p"])
def test_kernel_method_programs_verify(name, field):
    """-K (KernelOptimiser / nullspacedecomp, plinopt_optimize.inl:689-884, :1288-1353) with the build's decomposition rule:
    a greedy row basis in a random order, the other rows as combinations of the basis outputs, some of them kept direct.
    `--only K` prints that program, which must compute the matrix with exactly the reported operation count; 600
    restarts = 600 decompositions (one per restart, as the reference)."""
    path = os.path.join(DATA, name)
    q = ["-q", str(P), "--gpu", "0"] if field == "p" else []
    rc, out, err = run([OPT, "--only", "K", "-O", "600"] + q + [path])
    assert rc == 0, err
    assert re.search(r"# Found K: (\d+)\|(\d+) instead of", err), err
    rc, _, err2 = run([CHK] + (["-q", str(P)] if field == "p" else []) + ["-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2
    m = re.search(r"# \S*?(\d+)\tadditions\tinstead of (\d+)", err)
    final_adds = int(m.group(1)) if m else 0
    assert int(re.search(r"# \S*?(\d+)\tadditions", err2).group(1)) == final_adds


def test_kernel_method_reports_a_zero_dimensional_kernel():
    rc, out, err = run([OPT, "--only", "K", "-O", "10", "-q", str(P), "--gpu", "0", os.path.join(DATA, "cyclic.sms")])
    assert rc == 0 and "Zero dimensional kernel" in err
    rc, _, err2 = run([CHK, "-q", str(P), "-M", os.path.join(DATA, "cyclic.sms")], stdin=out)
    assert rc == 0 and "SUCCESS" in err2                      # the direct program is printed instead (:1473-1485)


@pytest.mark.parametrize("name,schedules,best", [("2x2x2_7_Winograd_L.sms", 3, (4, 0)), ("2x2x2_7_Winograd_R.sms", None, (4, 0)),
                                                 ("2x2x2_7_Winograd_P.sms", 10, (7, 0)), ("2x2x2_7_DPS-accurate_L.sms", 180, None)])
@pytest.mark.parametrize("field", ["Q", "p"])
def test_exhaustive_method_walks_the_whole_tree_of_small_inputs(name, schedules, best, field):
    """-E (AllCSEOpt / RecOptimizer / RecSub, plinopt_optimize.inl:889-1013, :1252-1281): every greedy CSE schedule of a
    small input, addressed by index; the best program computes the matrix with the reported count."""
    path = os.path.join(DATA, name)
    q = ["-q", str(P), "--gpu", "0"] if field == "p" else []
    rc, out, err = run([OPT, "--only", "E"] + q + [path])
    assert rc == 0, err
    g = re.search(r"# Found E: (\d+)\|(\d+) instead of \d+\|\d+\t\[schedule (\d+)\] \((\d+) schedules: the whole tree", err)
    assert g, err
    if schedules is not None:
        assert int(g.group(4)) == schedules
    if best is not None:
        assert (int(g.group(1)), int(g.group(2))) == best
    rc, _, err2 = run([CHK] + (["-q", str(P)] if field == "p" else []) + ["-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2 and ("%s,%s" % (g.group(1), g.group(2))) in err2, err2


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms", "2x2x2_7_Strassen_R.sms"])
def test_all_row_orders_of_the_kernel_method(name):
    """-N (AllKernelOpt, plinopt_optimize.inl:1357-1418): every order of the 7 rows (5040), distinct decompositions searched
    once; the program computes the matrix."""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "--only", "N", "-q", str(P), "--gpu", "0", "-O", "1000", path])
    assert rc == 0, err
    g = re.search(r"# Found N: (\d+)\|(\d+) instead of \d+\|\d+\t\[order (\d+), seed (\d+)\] \(5040 row orders, (\d+) distinct decompositions", err)
    assert g and 1 < int(g.group(5)) < 5040, err
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def test_direct_search_against_every_stored_program():
    """Every data/*.slp is an achievable bound for its matrix (reference `make opcheck`, bin/GDT.sh:41-65 counts them).  The
    direct method alone (`-D`, 100 restarts mod 131071) always beats the naive program and stays within 2x of the stored
    one (the stored programs come from all methods plus hand tuning: e.g. 4x4x4_48_accurate_R 76 operations, -D 138);
    on the +-1 matrices of 4x4x4_49_156 it is within 8 %.  A quality pin over all 80 pairs, not parity."""
    from concurrent.futures import ThreadPoolExecutor
    from plo_testlib import count_ops
    pairs = [s for s in sorted(glob.glob(os.path.join(DATA, "*.slp")))
             if "32x32x32" not in s and "-X_" not in s and os.path.exists(s[:-4] + ".sms")]
    assert len(pairs) >= 75

    def one(slp):
        sa, sm = count_ops(open(slp).read())
        rc, _, err = run([OPT, "-q", str(P), "-D", "-O", "100", "--gpu", "0", slp[:-4] + ".sms"])
        mm = re.search(r"# Found D: (\d+)\|(\d+) instead of (\d+)\|(\d+)", err)
        assert rc == 0 and mm, (slp, err)
        a, mu, na, nm = map(int, mm.groups())
        return os.path.basename(slp), sa + sm, a + mu, na + nm

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, pairs))
    for name, stored, found, naive in res:
        assert found <= naive and found <= 2 * stored, (name, stored, found, naive)
        if name.startswith("4x4x4_49_156"):
            assert found <= stored * 1.08, (name, stored, found)


# ----------------------------------------------------------------------------- compacter (post-pass on the winner)
CMP = os.path.join(ROOT, "bin", "compacter")


def _ops(err):
    mm = re.search(r"SUCCESS: correct SLP for \S+ \(\S+\) : (\d+),(\d+) ", err)
    return (int(mm.group(1)), int(mm.group(2))) if mm else None


@pytest.mark.parametrize("q", [None, 7])
def test_fdt_pipeline_optimizer_compacter_checker(q):
    """bin/FDT.sh:58-62: `optimizer -O 10 [-q 7] file | compacter -s | SLPchecker [-q 7] -M file` on every data matrix the
    reference's own driver takes (no -X_ placeholders, no 32x32x32; three inputs have a denominator divisible by 7).
    The compacted program computes the matrix, has no more operations and fewer words than the optimizer's."""
    from concurrent.futures import ThreadPoolExecutor
    files = [f for f in sorted(glob.glob(os.path.join(DATA, "*.sms"))) if "-X_" not in f and "32x32x32" not in f]
    qa = ["-q", str(q)] if q else []
    skip7 = ("2x2x2_7_DPS-integral-12.0662_P", "2x2x2_7_DPS-integral-12.0662_R", "4o4o8_Toom5_P")

    def one(f):
        if q == 7 and any(s in f for s in skip7):
            return None
        rc, prog, err = run([OPT, "-O", "10", "--gpu", "0"] + qa + [f])
        assert rc == 0, (f, err)
        rc, _, e0 = run([CHK] + qa + ["-M", f], stdin=prog)
        assert rc == 0 and "SUCCESS" in e0, (f, e0)
        rc, comp, ec = run([CMP, "-s"], stdin=prog)
        assert rc == 0, (f, ec)
        rc, _, e1 = run([CHK] + qa + ["-M", f], stdin=comp)
        assert rc == 0 and "SUCCESS" in e1, (f, e1, comp)
        b, a = _ops(e0), _ops(e1)
        assert a[0] <= b[0] and a[1] <= b[1], (f, b, a)
        mm = re.findall(r"# \S*?(\d+)\telements\tinstead of (\d+)", ec)[-1]      # the last line: whole run
        assert int(mm[0]) <= int(mm[1])
        return int(mm[1]) - int(mm[0])

    with ThreadPoolExecutor(max_workers=8) as ex:
        saved = [s for s in ex.map(one, files) if s is not None]
    assert len(saved) >= 140 and sum(1 for s in saved if s > 0) >= 0.9 * len(saved)     # the `t#:=i#;` preamble alone goes away


def test_compacter_rewrites():
    """hand-made program through `compacter -s` / `-n`: copies replaced, single uses written in place (signs folded,
    parentheses only where a product or quotient needs them), constant factors combined, leading minus rotated; the text is
    the literal oracle's (tests/test_compacter.py holds the tool to it at large); values unchanged (SLPchecker)."""
    src = ("t0:=i0;\nt1:=i1;\nt2:=i2;\nx0:=t0-t1;\nx1:=-t1+t2;\nx2:=x0*3;\nc0:=4/5;\nc2:=7;\nx3:=t2*c0;\nx4:=x3*5;\n"
           "o0:=t2-x1;\no1:=x2+x4;\no3:=-t1+t0;\no4:=t0*c2+t1*c2;\n")
    rc, out, err = run([CMP, "-s"], stdin=src)
    assert rc == 0, err
    assert out == "o0:=i2+i1-i2;\no1:=(i0-i1)*3+i2*4;\no3:=i0-i1;\no4:=i0*7+i1*7;\n", out
    rc, outn, _ = run([CMP, "-n"], stdin=src)
    assert "x0:=i0-i1;" in outn and "t0" not in outn and "c0" not in outn
    for prog in (src, out, outn):
        rc, sms, e = run([CHK], stdin=prog)
        assert rc == 0
        assert sms == run([CHK], stdin=src)[1]


# ----------------------------------------------------------------------------- factorizer
FCT = os.path.join(ROOT, "bin", "factorizer")


def test_factorizer_on_every_tall_matrix():
    """bin/FDT.sh:68-76: for every data matrix with at least as many rows as columns, `factorizer -q 7 file` and
    `factorizer file` must print SUCCESS (M == Alt . CoB, plinopt_sparsify.inl:872-907); Alt is never denser than M."""
    from concurrent.futures import ThreadPoolExecutor
    files = []
    for f in sorted(glob.glob(os.path.join(DATA, "*.sms"))):
        if "-X_" in f or "32x32x32" in f:
            continue
        m, n = [int(x) for x in next(l for l in open(f) if l.strip() and not l.startswith("#")).split()[:2]]
        if m >= n:
            files.append(f)
    assert len(files) >= 60
    skip7 = ("2x2x2_7_DPS-integral-12.0662_P", "2x2x2_7_DPS-integral-12.0662_R", "4o4o8_Toom5_P")

    def one(f):
        out = []
        for q in ([], ["-q", "7"]):
            if q and any(s in f for s in skip7):
                continue
            rc, _, err = run([FCT, "-O", "10"] + q + [f])
            assert rc == 0 and "SUCCESS: consistent factorization" in err and "ERROR" not in err, (f, q, err)
            mm = re.search(r"with (\d+) non-zeroes \((\d+) alt\.\) instead of (\d+)", err)
            assert int(mm.group(1)) <= int(mm.group(3)), (f, q, err)
            out.append(1)
        return len(out)

    with ThreadPoolExecutor(max_workers=8) as ex:
        assert sum(ex.map(one, files)) >= 2 * len(files) - 3


@pytest.mark.parametrize("name", ["3x4x7_63_rational", "4x4x4_48_accurate", "4x4x4_48_rational"])
@pytest.mark.parametrize("side", ["L", "R"])
def test_factorizer_reaches_the_stored_alternative_bases(name, side):
    """The reference holds factorizer outputs (data/*-ALT_*.sms with *-CoB_*.sms, inner dimension = rows - 1, pinned in
    tests/test_oracle_golden.py).  `bin/factorizer -k <that dimension>` on the same matrix finds an alternative matrix at
    least as sparse (the sparsest possible here: one entry per row plus the one solved row), its factorization is
    consistent, and the printed CoB times the printed Alt is the input (checked again in Python)."""
    from fractions import Fraction
    from plo_testlib import read_sms
    from test_oracle_golden import _spmul
    orig = os.path.join(DATA, "%s_%s.sms" % (name, side))
    ma, na, A = read_sms(os.path.join(DATA, "%s-ALT_%s.sms" % (name, side)))
    rc, cob, err = run([FCT, "-k", str(na), "-O", "300", "-S", orig])
    assert rc == 0 and "SUCCESS: consistent factorization" in err, err
    mm = re.search(r"with (\d+) non-zeroes \((\d+) alt\.\) instead of (\d+)", err)
    assert int(mm.group(1)) <= len(A), (mm.groups(), len(A))
    # the two printed matrices (CoB on stdout, Alt on the log stream, SMS format) multiply back to the input
    def parse(txt):
        lines = [l for l in txt.splitlines() if l.strip() and not l.startswith("#")]
        m, n = int(lines[0].split()[0]), int(lines[0].split()[1])
        ent = {}
        for l in lines[1:]:
            i, j, v = l.split()
            if i == "0":
                break
            ent[(int(i) - 1, int(j) - 1)] = Fraction(v)
        return m, n, ent
    alt_txt = err[err.index("residuum profile"):]
    alt_txt = alt_txt[alt_txt.index("\n") + 1:]
    mc, nc, C = parse(cob)
    mA, nA, Al = parse(alt_txt)
    mo, no, X = read_sms(orig)
    assert (mA, nA, mc, nc) == (mo, na, na, no) and _spmul(Al, C) == X


# ----------------------------------------------------------------------------- -E with RecSub's own accounting
RECSUB_TOYS = ["2x2x2_7_Winograd_L.sms", "2x2x2_7_Winograd_P.sms", "2x2x2_7_DPS-accurate_L.sms", "2x2x2_7_DPS-accurate_P.sms", "2x2x2_7_DPS-accurate_R.sms"]


@pytest.mark.parametrize("name", RECSUB_TOYS)
def test_exhaustive_method_with_recsub_accounting_equals_literal_recsub(name):
    """`bin/optimizer --only E --recsub`: the schedule tree walked by index, chosen as RecSub chooses (additions, then its
    own multiplication count: naiveOps minus the savings of the steps, plinopt_optimize.inl:950-959) gives the values of the
    LITERAL recursive restatement in the oracle (every pair instance of every row is a child there, :936-939: the same set
    of schedules, walked with repetitions).  The additions also equal those of the printed program (the savings
    accounting and Optimizer's count are the same number)."""
    from plo_testlib import OracleMatrix
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    adds, muls_r, muls_final, nodes = M.recsub()
    rc, out, err = run([OPT, "-q", str(P), "--only", "E", "--recsub", "--gpu", "0", path])
    assert rc == 0, err
    mm = re.search(r"# RecSub accounting: (\d+)\|(\d+) .*program (\d+)\|(\d+)", err)
    assert mm, err
    assert (int(mm.group(1)), int(mm.group(2))) == (adds, muls_r), (mm.groups(), adds, muls_r)
    assert int(mm.group(3)) == adds and "the whole tree" in err and nodes >= 3
    rc, _, err2 = run([CHK, "-q", str(P), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2


# ----------------------------------------------------------------------------- -K against the oracle's restatement
def kernel_oracle_argmin(M, seed0, n):
    """best restart of the kernel method under (cmpOpCount, seed), from oracle/plo_oracle.c `plo_oracle_kernel_restart`"""
    best = None
    for s in range(seed0, seed0 + n):
        r = M.kernel_restart(s)
        assert r is not None
        key = (r[0] + r[1], r[0], s)
        if best is None or key < best[0]:
            best = (key, r, s)
    return best[1], best[2]


def test_oracle_reports_a_zero_dimensional_kernel():
    from plo_testlib import OracleMatrix
    assert OracleMatrix.from_sms(os.path.join(DATA, "2x2x2_7_DPS-accurate_P.sms"), P).kernel_restart(3) is None      # 4 x 7, full row rank (:883)


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms"])
def test_kernel_method_equals_the_oracle_restatement(name):
    """`bin/optimizer --only K` (one decomposition per restart, as the reference :1299-1340): winner, counts, rank, NotIndep
    and number of dependent rows are those of the oracle's independent restatement of the decomposition rule followed by
    the oracle's chained Optimizer (not a comparison of the product with itself)."""
    from plo_testlib import OracleMatrix
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 120
    (a, mu, rank, ni, nd), seed = kernel_oracle_argmin(M, 40, n)
    rc, out, err = run([OPT, "-q", str(P), "--only", "K", "-O", str(n), "--seed", "40", "--gpu", "0", path])
    assert rc == 0, err
    g = re.search(r"# Found K: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\+(\d+), (\d+) dependent rows\)", err)
    assert g and tuple(int(x) for x in g.groups()) == (a, mu, seed, rank, ni, nd), (g and g.groups(), (a, mu, seed, rank, ni, nd))


@pytest.mark.parametrize("name", ["2x2x2_7_Winograd_L.sms", "4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "2x2x2_7_DPS-accurate_L.sms", "cyclic.sms"])
@pytest.mark.parametrize("field", ["Q", "p"])
def test_kernel_method_with_identity_goals(name, field):
    """-K -F (KFI, reference include/plinopt_optimize.inl:637-685): the kernel method on [M ; I]; the printed program (identity
    goals removed, outputs copied from the goals) computes M with exactly the reported operation count.  cyclic.sms is
    square and invertible: every row of [M ; I] beyond the basis is dependent, the method still runs."""
    path = os.path.join(DATA, name)
    q = ["-q", str(P), "--gpu", "0"] if field == "p" else []
    rc, out, err = run([OPT, "--only", "K", "-F", "-O", "200"] + q + [path])
    assert rc == 0, err
    mm = re.search(r"# Found K with identity goals \(-F\): (\d+)\|(\d+) after cleaning", err)
    assert mm, err
    rc, _, err2 = run([CHK] + (["-q", str(P)] if field == "p" else []) + ["-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2
    got = re.search(r": (\d+),(\d+) Matrix-Vector", err2)
    fin = re.search(r"# \S*?(\d+)\tadditions\tinstead of (\d+)", err)
    assert fin and int(got.group(1)) == int(fin.group(1))


# ----------------------------------------------------------------------------- -G against the oracle's restatement of the LU rule
def lu_oracle_argmin(M, seed0, n):
    """best restart of the LU method under (cmpOpCount, seed): the oracle's own LU factors (plo_oracle_lu, dense restatement of the
    build's pivot rule) through the oracle's chained Optimizer"""
    from plo_testlib import oracle_chain
    U, L, rank = M.lu_factors()
    best = None
    for s in range(seed0, seed0 + n):
        a, mu = oracle_chain(U, L, s)
        key = (a + mu, a, s)
        if best is None or key < best:
            best = key
    return best[1], best[0] - best[1], best[2], rank


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "2x2x2_7_Winograd_P.sms", "2x2x2_7_DPS-accurate_L.sms", "cyclic.sms"])
def test_lu_method_equals_the_oracle_restatement(name):
    """`bin/optimizer --only G` (host loop): winner, counts and rank are those of the oracle's independent LU restatement followed by the
    oracle's chained Optimizer -- not a comparison of the product with itself"""
    from plo_testlib import OracleMatrix
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 60
    a, mu, seed, rank = lu_oracle_argmin(M, 3, n)
    rc, out, err = run([OPT, "-q", str(P), "--only", "G", "-O", str(n), "--seed", "3", "--gpu", "0", path])
    assert rc == 0, err
    g = re.search(r"# Found G: (\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\] \(rank (\d+)\)", err)
    assert g and tuple(int(x) for x in g.groups()) == (a, mu, seed, rank), (g and g.groups(), (a, mu, seed, rank))


# ----------------------------------------------------------------------------- -A against the oracle's restatement of the back-solver rule
def ab_oracle_argmin(M, seed0, n, k=None):
    """the factorization the oracle finds with 1 + n/8 back-solves (ABOptimiser :1129-1130), then its chained Optimizer on CoB and Alt
    minimised over n seeds: ((rows, inner, cols, nnz Alt, nnz CoB), adds, muls, seed)"""
    from plo_testlib import oracle_chain
    r = M.ab_factors(seed0, 1 + (n >> 3), k)
    if r is None:
        return None
    CoB, Alt, sc = r
    best = None
    for s in range(seed0, seed0 + n):
        a, mu = oracle_chain(CoB, Alt, s)
        key = (a + mu, a, s)
        if best is None or key < best:
            best = key
    return (Alt.m, Alt.n, CoB.n, sc[0], sc[2]), best[1], best[0] - best[1], best[2]


AB_PAT = r"# Found A: \((\d+)x(\d+)x(\d+) (\d+)/(\d+)\)\t(\d+)\|(\d+) instead of \d+\|\d+\t\[seed (\d+)\]"


@pytest.mark.parametrize("name", ["4x4x4_49_156_L.sms", "3x3x6_40_L.sms", "4x4x4_48_rational_L.sms", "2x2x2_7_Winograd_L.sms", "2x2x2_7_DPS-accurate_L.sms", "3x3x3_23_58_L.sms"])
def test_ab_method_equals_the_oracle_restatement(name):
    """`bin/optimizer --only A` (host loop), first inner dimension (= column count): factor shapes and sparsities, winner, counts and seed
    are those of the oracle's independent restatement of the back-solver rule followed by the oracle's chained Optimizer"""
    from plo_testlib import OracleMatrix
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    n = 64
    exp = ab_oracle_argmin(M, 5, n)
    assert exp is not None
    rc, out, err = run([OPT, "-q", str(P), "--only", "A", "-O", str(n), "--seed", "5", "--gpu", "0", path])
    assert rc == 0, err
    g = re.search(AB_PAT, err)
    assert g, err
    got = tuple(int(x) for x in g.groups())
    assert (got[:5], got[5], got[6], got[7]) == exp, (got, exp)


# ----------------------------------------------------------------------------- -N against the oracle
def allkernels_oracle(M, seed0, loops):
    """AllKernelOpt as bin/optimizer runs it, from the oracle alone: every order of the rows (next_permutation order), equal
    decompositions recognised by their signature (first order kept), the restarts shared out, best under (cmpOpCount, seed)"""
    import itertools
    distinct = {}
    npi = 0
    for pi, order in enumerate(itertools.permutations(range(M.m))):
        r = M.kernel_order(order, seed0 + pi, seed0)
        assert r is not None
        distinct.setdefault(r[4], (order, pi))
        npi = pi + 1
    per = max(1, loops // len(distinct))
    best = None
    for k, sig in enumerate(sorted(distinct)):
        order, pi = distinct[sig]
        for j in range(per):
            s = seed0 + k * per + j
            a, mu, _, _, _ = M.kernel_order(order, seed0 + pi, s)
            key = (a + mu, a, s)
            if best is None or key < best[0]:
                best = (key, pi)
    return best[0][1], best[0][0] - best[0][1], best[1], best[0][2], npi, len(distinct), per


N_PAT = r"# Found N: (\d+)\|(\d+) instead of \d+\|\d+\t\[order (\d+), seed (\d+)\] \((\d+) row orders, (\d+) distinct decompositions, (\d+) restarts each"


@pytest.mark.parametrize("name", ["2x2x2_7_DPS-accurate_L.sms", "2x2x2_7_Winograd_L.sms"])
def test_all_row_orders_equal_the_oracle(name):
    """`bin/optimizer --only N` (host loop) against the oracle's decomposition with prescribed orders (plo_oracle_kernel_order): number of
    distinct decompositions, restarts each, winner (counts, order, seed)"""
    from plo_testlib import OracleMatrix
    path = os.path.join(DATA, name)
    M = OracleMatrix.from_sms(path, P)
    exp = allkernels_oracle(M, 13, 3000)
    rc, out, err = run([OPT, "-q", str(P), "--only", "N", "-O", "3000", "--seed", "13", "--gpu", "0", path])
    assert rc == 0, err
    g = re.search(N_PAT, err)
    assert g and tuple(int(x) for x in g.groups()) == exp, (g and g.groups(), exp)


# ----------------------------------------------------------------------------- moduli above 2^31 (host loops; the reference's field is Modular<Integer>)
@pytest.mark.parametrize("q", [4294967311, 2305843009213693951])          # just above 2^32; 2^61 - 1
@pytest.mark.parametrize("name", ["4x4x4_48_rational_L.sms", "2x2x2_7_DPS-accurate_L.sms", "3x3x3_23_58_P.sms"])
def test_large_moduli_run_the_host_loops_and_verify(name, q):
    """the HIP kernels hold 31-bit residues; `-q p` with p >= 2^31 runs every method on the host over Z/pZ with 128-bit products, says so,
    and the program verifies modulo the same p (bin/SLPchecker)"""
    path = os.path.join(DATA, name)
    rc, out, err = run([OPT, "-q", str(q), "-O", "40", "-A", path])
    assert rc == 0 and "modulus above 2^31: host loops" in err and "GPU" not in err.replace("GPU kernels hold", ""), err
    rc, _, err2 = run([CHK, "-q", str(q), "-M", path], stdin=out)
    assert rc == 0 and "SUCCESS" in err2, err2


def test_large_modulus_sparsifier_and_factorizer():
    path = os.path.join(DATA, "2x2x2_7_Winograd_L.sms")
    rc, out, err = run([SPS, "-q", "4294967311", "-c", "4", "-S", path])
    assert rc == 0 and "host enumeration" in err and "SUCCESS: consistent factorization" in err
    rc, out, err = run([os.path.join(ROOT, "bin", "factorizer"), "-q", "4294967311", "-O", "50", "-S", path])
    assert rc == 0 and "SUCCESS" in err
