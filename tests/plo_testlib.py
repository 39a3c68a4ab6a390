"""Independent test-side checkers (pure Python, no product code):

* SMS reader (format: /root/reference/README.md:71-77, data/README.md:10-16),
* SLP evaluator -> matrix (what /root/reference/src/SLPchecker.cpp:22-105 checks),
* op counter with `lineOperations` rules (/root/reference/include/plinopt_programs.inl:116-133),
* ctypes binding of oracle/libplo_oracle.so (the CPU checker).

Nothing here reads /root/reference at run time; data fixtures live in tests/golden/data.
"""
import ctypes
import os
import re
import subprocess
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


# ----------------------------------------------------------------------------- SMS
def read_sms(path):
    """Returns (m, n, {(i,j): Fraction}) with 0-based indices."""
    ent = {}
    m = n = None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or line.startswith("%"):
                continue
            tok = line.split()
            if m is None:
                m, n = int(tok[0]), int(tok[1])
                continue
            i, j = int(tok[0]), int(tok[1])
            if i == 0 and j == 0:
                break
            v = Fraction(tok[2])
            if v != 0:
                ent[(i - 1, j - 1)] = v
    return m, n, ent


def to_csr_mod(m, n, ent, p):
    """CSR over Z_p (rational a/b -> a*b^-1 mod p; entries vanishing mod p dropped)."""
    rows = [[] for _ in range(m)]
    for (i, j), v in ent.items():
        r = (v.numerator % p) * pow(v.denominator % p, -1, p) % p
        if r:
            rows[i].append((j, r))
    rowptr, col, val = [0], [], []
    for r in rows:
        r.sort()
        for j, v in r:
            col.append(j)
            val.append(v)
        rowptr.append(len(col))
    return rowptr, col, val


# ----------------------------------------------------------------------------- SLP
_TOK = re.compile(r"\s*(:=|[A-Za-z_][A-Za-z_0-9]*|\d+|[-+*/();])")


class _Parser:
    """Recursive-descent evaluation of `lhs:=expr;` into sparse linear forms
    {input_index or 'c': coeff} over a field given by (add, mul, inv, from_int)."""

    def __init__(self, field):
        self.F = field
        self.vars = {}

    def lin_add(self, x, y, sign=1):
        F = self.F
        out = dict(x)
        for k, v in y.items():
            nv = F.add(out.get(k, F.zero), v if sign > 0 else F.neg(v))
            if nv == F.zero:
                out.pop(k, None)
            else:
                out[k] = nv
        return out

    def lin_scale(self, x, c):
        F = self.F
        out = {}
        for k, v in x.items():
            nv = F.mul(v, c)
            if nv != F.zero:
                out[k] = nv
        return out

    def is_const(self, x):
        return all(k == "c" for k in x)

    def const(self, x):
        return x.get("c", self.F.zero)

    def parse_line(self, toks):
        self.t = toks
        self.i = 0
        lhs = self.next()
        assert self.next() == ":=", toks
        val = self.expr()
        assert self.peek() in (";", None), toks
        self.vars[lhs] = val
        return lhs

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def expr(self):
        sign = 1
        if self.peek() in "+-":
            sign = -1 if self.next() == "-" else 1
        acc = self.term()
        if sign < 0:
            acc = self.lin_scale(acc, self.F.neg(self.F.one))
        while self.peek() in ("+", "-"):
            op = self.next()
            acc = self.lin_add(acc, self.term(), 1 if op == "+" else -1)
        return acc

    def term(self):
        acc = self.factor()
        while self.peek() in ("*", "/"):
            op = self.next()
            rhs = self.factor()
            if op == "*":
                if self.is_const(rhs):
                    acc = self.lin_scale(acc, self.const(rhs))
                else:
                    assert self.is_const(acc), "non-linear product"
                    acc = self.lin_scale(rhs, self.const(acc))
            else:
                assert self.is_const(rhs), "division by non-constant"
                acc = self.lin_scale(acc, self.F.inv(self.const(rhs)))
        return acc

    def factor(self):
        tok = self.next()
        if tok == "(":
            v = self.expr()
            assert self.next() == ")"
            return v
        if tok == "-":
            return self.lin_scale(self.factor(), self.F.neg(self.F.one))
        if tok.isdigit():
            c = self.F.from_int(int(tok))
            return {"c": c} if c != self.F.zero else {}
        if tok in self.vars:
            return self.vars[tok]
        if tok[0] == "i" and tok[1:].isdigit():
            return {int(tok[1:]): self.F.one}
        raise KeyError("undefined variable %s" % tok)


class FieldQ:
    zero, one = Fraction(0), Fraction(1)
    add = staticmethod(lambda a, b: a + b)
    mul = staticmethod(lambda a, b: a * b)
    neg = staticmethod(lambda a: -a)
    inv = staticmethod(lambda a: 1 / a)
    from_int = staticmethod(lambda n: Fraction(n))


class FieldP:
    def __init__(self, p):
        self.p = p
        self.zero, self.one = 0, 1 % p

    def add(self, a, b):
        return (a + b) % self.p

    def mul(self, a, b):
        return a * b % self.p

    def neg(self, a):
        return (-a) % self.p

    def inv(self, a):
        return pow(a, -1, self.p)

    def from_int(self, n):
        return n % self.p


def tokenize_slp(text):
    lines = []
    for raw in text.splitlines():
        raw = raw.split("#", 1)[0].strip()
        if not raw or ":=" not in raw:
            continue
        toks = _TOK.findall(raw)
        assert "".join(toks) == re.sub(r"\s+", "", raw), raw
        lines.append(toks)
    return lines


def eval_slp(text, field, outchar="o"):
    """Returns {(i,j): coeff} of the matrix computed on outputs `o#` from inputs `i#`."""
    P = _Parser(field)
    for toks in tokenize_slp(text):
        P.parse_line(toks)
    mat = {}
    for name, lin in P.vars.items():
        if name[0] == outchar and name[1:].isdigit():
            i = int(name[1:])
            for k, v in lin.items():
                assert k != "c", "constant term in output"
                mat[(i, k)] = v
    return mat


def count_ops(text):
    """(adds, muls) with lineOperations semantics: a sign right after ':=' or '('
    is a negation, not an addition; every '*' or '/' *operator* is one multiplication
    (a rational constant a/b is one token, plinopt_programs.inl:646-659)."""
    adds = muls = 0
    for toks in tokenize_slp(text):
        # merge natural '/' natural into one rational token, as programParser does
        merged = []
        k = 0
        while k < len(toks):
            if (toks[k].isdigit() and k + 2 < len(toks) and toks[k + 1] == "/" and toks[k + 2].isdigit()):
                merged.append(toks[k] + "/" + toks[k + 2])
                k += 3
            else:
                merged.append(toks[k])
                k += 1
        negator = False
        for w in merged:
            if w in ("+", "-") and not negator:
                adds += 1
            elif w in ("*", "/"):
                muls += 1
            negator = w in (":=", "(")
    return adds, muls


# ----------------------------------------------------------------------------- oracle binding
_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ROOT, "oracle", "libplo_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(so)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.plo_oracle_optimizer.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32,
                                           ctypes.c_uint64, ctypes.c_char_p, u32p, u32p,
                                           ctypes.POINTER(ctypes.c_void_p)]
        L.plo_oracle_cost_many.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32,
                                           u64p, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p, ctypes.c_int]
        L.plo_oracle_cse_search.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32,
                                            ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, u32p, u32p, u64p,
                                            ctypes.c_int]
        L.plo_oracle_first_ties.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32,
                                            u32p, ctypes.c_int, u32p]
        L.plo_oracle_chain.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p,
                                       ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32, ctypes.c_uint64,
                                       u32p, u32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
        L.plo_oracle_cob_search.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), u64p, u32p]
        i64p = ctypes.POINTER(ctypes.c_int64)
        mat = [ctypes.c_uint32, u32p, u32p, i64p, i64p]
        L.plo_oracle_tril_cost_many.argtypes = [ctypes.c_uint32] + mat * 3 + [u64p, ctypes.c_uint64, ctypes.c_uint64, u32p, ctypes.c_int]
        L.plo_oracle_tril_program.argtypes = [ctypes.c_uint32] + mat * 3 + [ctypes.c_uint64, ctypes.c_int, u32p, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_oracle_tril_search.argtypes = [ctypes.c_uint32] + mat * 3 + [ctypes.c_uint64, ctypes.c_uint64, u32p, u64p, u32p]
        L.plo_oracle_enum_optimizer.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p,
                                                u32p, u32p, u64p, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_oracle_enum_cost_many.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                                u32p, u32p, u64p, ctypes.c_int]
        L.plo_oracle_sp_coeffs.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p]
        L.plo_oracle_sp_local.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, ctypes.c_uint32]
        L.plo_oracle_sparsify.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, u32p, u32p, u64p]
        L.plo_oracle_sp_coeffs_q.argtypes = [ctypes.c_uint32, ctypes.c_uint32, i64p, i64p, ctypes.c_uint32, i64p, i64p, u32p]
        L.plo_oracle_sparsify_q.argtypes = [ctypes.c_uint32, ctypes.c_uint32, i64p, i64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, i64p, i64p, i64p, i64p, u64p]
        L.plo_oracle_naive_ops.argtypes = [ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, u32p, u32p]
        L.plo_oracle_naive_ops.restype = None
        L.plo_oracle_free.argtypes = [ctypes.c_void_p]
        L.plo_oracle_free.restype = None
        L.plo_oracle_rng_state0.argtypes = [ctypes.c_uint64]
        L.plo_oracle_rng_state0.restype = ctypes.c_uint32
        L.plo_oracle_rng_next.argtypes = [u32p]
        L.plo_oracle_rng_next.restype = ctypes.c_uint32
        _oracle = L
    return _oracle


def _arr(xs, ty=ctypes.c_uint32):
    return (ty * max(len(xs), 1))(*xs)


class OracleMatrix:
    """CSR over Z_p handed to the oracle."""

    def __init__(self, m, n, rowptr, col, val, p):
        self.m, self.n, self.p = m, n, p
        self.rowptr, self.col, self.val = list(rowptr), list(col), list(val)
        self._rp, self._c, self._v = _arr(self.rowptr), _arr(self.col), _arr(self.val)

    @classmethod
    def from_sms(cls, path, p):
        m, n, ent = read_sms(path)
        rp, c, v = to_csr_mod(m, n, ent, p)
        return cls(m, n, rp, c, v, p)

    def dense(self):
        d = {}
        for i in range(self.m):
            for k in range(self.rowptr[i], self.rowptr[i + 1]):
                d[(i, self.col[k])] = self.val[k]
        return d

    def optimizer(self, seed, letters=b"otri", text=True):
        a, mu = ctypes.c_uint32(), ctypes.c_uint32()
        tp = ctypes.c_void_p()
        rc = oracle().plo_oracle_optimizer(self.m, self.n, self._rp, self._c, self._v, self.p, seed, letters,
                                           ctypes.byref(a), ctypes.byref(mu),
                                           ctypes.byref(tp) if text else None)
        assert rc == 0
        s = None
        if text:
            s = ctypes.string_at(tp).decode()
            oracle().plo_oracle_free(tp)
        return a.value, mu.value, s

    def cost_many(self, seeds=None, seed0=0, nseeds=0, nthreads=1):
        if seeds is not None:
            nseeds = len(seeds)
            sp = _arr(seeds, ctypes.c_uint64)
        else:
            sp = None
        adds, muls = (ctypes.c_uint32 * max(nseeds, 1))(), (ctypes.c_uint32 * max(nseeds, 1))()
        rc = oracle().plo_oracle_cost_many(self.m, self.n, self._rp, self._c, self._v, self.p, sp, seed0, nseeds,
                                           adds, muls, nthreads)
        assert rc == 0
        return list(adds[:nseeds]), list(muls[:nseeds])

    def search(self, seed0, nseeds, cost_mode=0, nthreads=1):
        a, mu, s = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        rc = oracle().plo_oracle_cse_search(self.m, self.n, self._rp, self._c, self._v, self.p, seed0, nseeds,
                                            cost_mode, ctypes.byref(a), ctypes.byref(mu), ctypes.byref(s), nthreads)
        assert rc == 0
        return a.value, mu.value, s.value

    def kernel_restart(self, seed):
        """one restart of the kernel method with the build's decomposition rule restated in the oracle:
        (adds, muls, rank, notindep, dependent rows) or None for a zero dimensional kernel"""
        a, mu, r, ni, nd = (ctypes.c_uint32() for _ in range(5))
        rc = oracle().plo_oracle_kernel_restart(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, ctypes.c_uint64(seed),
                                                ctypes.byref(a), ctypes.byref(mu), ctypes.byref(r), ctypes.byref(ni), ctypes.byref(nd))
        if rc == -2:
            return None
        assert rc == 0, rc
        return a.value, mu.value, r.value, ni.value, nd.value

    def lu_factors(self):
        """(U, L, rank) of the -G method by the oracle's dense restatement of the build's pivot rule (plo_oracle_lu): two OracleMatrix"""
        U = (ctypes.c_uint32 * (self.m * self.n))(); L = (ctypes.c_uint32 * (self.m * self.m))(); rk = ctypes.c_uint32()
        rc = oracle().plo_oracle_lu(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, U, L, ctypes.byref(rk))
        assert rc == 0

        def csr(D, rows, cols):
            rp, c, v = [0], [], []
            for i in range(rows):
                for j in range(cols):
                    if D[i * cols + j]:
                        c.append(j); v.append(D[i * cols + j])
                rp.append(len(c))
            return OracleMatrix(rows, cols, rp, c, v, self.p)
        return csr(U, self.m, self.n), csr(L, self.m, self.m), rk.value

    def ab_factors(self, seed0, loops, k=None):
        """(CoB, Alt, score) of the -A method by the oracle's restatement of the build's back-solver rule (plo_oracle_ab_factor); None when m <= n"""
        k = self.n if k is None else k
        Alt = (ctypes.c_uint32 * (self.m * k))(); CoB = (ctypes.c_uint32 * (k * self.n))(); sc = (ctypes.c_uint32 * 3)()
        rc = oracle().plo_oracle_ab_factor(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, ctypes.c_uint64(seed0), loops, k, Alt, CoB, sc)
        if rc == -2:
            return None
        assert rc == 0

        def csr(D, rows, cols):
            rp, c, v = [0], [], []
            for i in range(rows):
                for j in range(cols):
                    if D[i * cols + j]:
                        c.append(j); v.append(D[i * cols + j])
                rp.append(len(c))
            return OracleMatrix(rows, cols, rp, c, v, self.p)
        return csr(CoB, k, self.n), csr(Alt, self.m, k), tuple(sc)

    def kernel_order(self, order, dseed, cseed):
        """one decomposition of -N with the prescribed row order (plo_oracle_kernel_order): (adds, muls, rank, notindep, signature) where the
        signature is the one bin/optimizer uses to recognise equal decompositions: kept dependent rows, m + NotIndep, column pattern of Dep"""
        m = self.m
        a, mu, r, ni, nd = (ctypes.c_uint32() for _ in range(5))
        dep = (ctypes.c_uint32 * m)(); cols = (ctypes.c_ubyte * (m * m))()
        rc = oracle().plo_oracle_kernel_order(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, _arr(list(order)),
                                              ctypes.c_uint64(dseed), ctypes.c_uint64(cseed), ctypes.byref(a), ctypes.byref(mu), ctypes.byref(r),
                                              ctypes.byref(ni), ctypes.byref(nd), dep, cols)
        if rc == -2:
            return None
        assert rc == 0, rc
        kept = nd.value
        sig = tuple(dep[j] for j in range(kept)) + (m + ni.value,) + tuple(i for j in range(kept) for i in range(m) if cols[j * m + i])
        return a.value, mu.value, r.value, ni.value, sig

    def recsub(self):
        """literal RecSub / RecOptimizer (plinopt_optimize.inl:889-1013): (adds, muls before ProgramGen, muls after, nodes)"""
        a, mr, mf, nd = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        rc = oracle().plo_oracle_recsub(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p,
                                        ctypes.byref(a), ctypes.byref(mr), ctypes.byref(mf), ctypes.byref(nd))
        assert rc == 0
        return a.value, mr.value, mf.value, nd.value

    def enum_optimizer(self, index, letters=b"otri"):
        """one schedule of RecSub's tree (-E): -> (adds, muls, radix product, text)"""
        a = ctypes.c_uint32(); mu = ctypes.c_uint32(); pr = ctypes.c_uint64(); txt = ctypes.c_void_p()
        rc = oracle().plo_oracle_enum_optimizer(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, index, letters,
                                                ctypes.byref(a), ctypes.byref(mu), ctypes.byref(pr), ctypes.byref(txt))
        assert rc == 0
        text = ctypes.string_at(txt).decode(); oracle().plo_oracle_free(txt)
        return a.value, mu.value, pr.value, text

    def enum_cost_many(self, first, n, nthreads=1):
        a = (ctypes.c_uint32 * max(n, 1))(); mu = (ctypes.c_uint32 * max(n, 1))(); pr = (ctypes.c_uint64 * max(n, 1))()
        rc = oracle().plo_oracle_enum_cost_many(self.m, self.n, _arr(self.rowptr), _arr(self.col), _arr(self.val), self.p, first, n, a, mu, pr, nthreads)
        assert rc == 0
        return list(a[:n]), list(mu[:n]), list(pr[:n])

    def first_ties(self, cap=4096):
        buf = (ctypes.c_uint32 * (3 * cap))()
        mf = ctypes.c_uint32()
        nt = oracle().plo_oracle_first_ties(self.m, self.n, self._rp, self._c, self._v, self.p, buf, cap,
                                            ctypes.byref(mf))
        return [(buf[3 * k], buf[3 * k + 1], buf[3 * k + 2]) for k in range(min(nt, cap))], mf.value

    def naive_ops(self):
        a, mu = ctypes.c_uint32(), ctypes.c_uint32()
        oracle().plo_oracle_naive_ops(self.m, self._rp, self._v, self.p, ctypes.byref(a), ctypes.byref(mu))
        return a.value, mu.value


def oracle_chain(A, B, seed, text=False):
    """Optimizer() on OracleMatrix A then on B with one random stream (LU method); returns (adds, muls[, textA, textB])."""
    a, mu = ctypes.c_uint32(), ctypes.c_uint32()
    t1, t2 = ctypes.c_void_p(), ctypes.c_void_p()
    rc = oracle().plo_oracle_chain(A.m, A.n, A._rp, A._c, A._v, B.m, B.n, B._rp, B._c, B._v, A.p, seed,
                                   ctypes.byref(a), ctypes.byref(mu),
                                   ctypes.byref(t1) if text else None, ctypes.byref(t2) if text else None)
    assert rc == 0
    if not text:
        return a.value, mu.value
    s1, s2 = ctypes.string_at(t1).decode(), ctypes.string_at(t2).decode()
    oracle().plo_oracle_free(t1)
    oracle().plo_oracle_free(t2)
    return a.value, mu.value, s1, s2


def oracle_cob_search(n, m, TM, Cand, row, off, coeffs, p, w0=-1, w1=-1):
    """TM, Cand: flat row-major lists.  Returns (zeros_v, zeros_w, index, found)."""
    zv, zw = ctypes.c_int32(), ctypes.c_int32()
    idx, fnd = ctypes.c_uint64(), ctypes.c_uint32()
    rc = oracle().plo_oracle_cob_search(n, m, _arr(TM), _arr(Cand), row, off, _arr(coeffs), len(coeffs), p, w0, w1,
                                        ctypes.byref(zv), ctypes.byref(zw), ctypes.byref(idx), ctypes.byref(fnd))
    assert rc == 0
    return zv.value, zw.value, idx.value, fnd.value


def oracle_sp_coeffs(TM, p, maxnumcoeff):
    """Coefficient set of localSparsifier (plinopt_sparsify.inl:256-268) for the dense matrix TM (list of rows)."""
    n, m = len(TM), len(TM[0])
    out = (ctypes.c_uint32 * (maxnumcoeff + 8))()
    cnt = ctypes.c_uint32()
    assert oracle().plo_oracle_sp_coeffs(n, m, _arr([x for r in TM for x in r]), p, maxnumcoeff, out, ctypes.byref(cnt)) == 0
    return list(out[:cnt.value])


def oracle_sparsify(M, p, blocksize=4, maxnumcoeff=11, initial_elimination=True):
    """blockSparsifier (plinopt_sparsify.inl:667-748) of the dense m x n matrix M (list of rows) over Z_p: (CoB, Res, candidates)."""
    m, n = len(M), len(M[0])
    cob = (ctypes.c_uint32 * (n * n))()
    res = (ctypes.c_uint32 * (m * n))()
    cand = ctypes.c_uint64()
    rc = oracle().plo_oracle_sparsify(m, n, _arr([x for r in M for x in r]), p, blocksize, maxnumcoeff, 1 if initial_elimination else 0, cob, res, ctypes.byref(cand))
    assert rc == 0
    return [list(cob[i * n:(i + 1) * n]) for i in range(n)], [list(res[i * n:(i + 1) * n]) for i in range(m)], cand.value


def oracle_sp_coeffs_q(TM, maxnumcoeff):
    """Coefficient set of localSparsifier over Q (plinopt_sparsify.inl:256-268 with QField<Rational>) for the dense matrix TM of Fractions."""
    from fractions import Fraction
    n, m = len(TM), len(TM[0])
    flat = [Fraction(x) for r in TM for x in r]
    on, od = (ctypes.c_int64 * (maxnumcoeff + 8))(), (ctypes.c_int64 * (maxnumcoeff + 8))()
    cnt = ctypes.c_uint32()
    assert oracle().plo_oracle_sp_coeffs_q(n, m, _arr([x.numerator for x in flat], ctypes.c_int64), _arr([x.denominator for x in flat], ctypes.c_int64),
                                           maxnumcoeff, on, od, ctypes.byref(cnt)) == 0
    return [Fraction(on[k], od[k]) for k in range(cnt.value)]


def oracle_sparsify_q(M, blocksize=4, maxnumcoeff=11, initial_elimination=True):
    """blockSparsifier (plinopt_sparsify.inl:667-748) of the dense m x n matrix M of Fractions over Q: (CoB, Res, candidates)."""
    from fractions import Fraction
    m, n = len(M), len(M[0])
    flat = [Fraction(x) for r in M for x in r]
    cn, cd = (ctypes.c_int64 * (n * n))(), (ctypes.c_int64 * (n * n))()
    rn, rd = (ctypes.c_int64 * (m * n))(), (ctypes.c_int64 * (m * n))()
    cand = ctypes.c_uint64()
    rc = oracle().plo_oracle_sparsify_q(m, n, _arr([x.numerator for x in flat], ctypes.c_int64), _arr([x.denominator for x in flat], ctypes.c_int64),
                                        blocksize, maxnumcoeff, 1 if initial_elimination else 0, cn, cd, rn, rd, ctypes.byref(cand))
    assert rc == 0
    return ([[Fraction(cn[i * n + j], cd[i * n + j]) for j in range(n)] for i in range(n)],
            [[Fraction(rn[i * n + j], rd[i * n + j]) for j in range(n)] for i in range(m)], cand.value)


def dense_q(path):
    """dense matrix of Fractions of an SMS file (list of rows)"""
    from fractions import Fraction
    m, n, ent = read_sms(path)
    D = [[Fraction(0)] * n for _ in range(m)]
    for (i, j), v in ent.items():
        D[i][j] = v
    return D


def parse_sms_text_q(text):
    """dense matrix of Fractions of an SMS text as the tools print it over Q (entries `a` or `a/b`)"""
    from fractions import Fraction
    lines = [ln for ln in text.splitlines() if ln.strip() and not ln.lstrip().startswith("#")]
    m, n = int(lines[0].split()[0]), int(lines[0].split()[1])
    D = [[Fraction(0)] * n for _ in range(m)]
    for ln in lines[1:]:
        i, j, v = ln.split()[:3]
        if int(i) == 0:
            break
        D[int(i) - 1][int(j) - 1] = Fraction(v)
    return D


def dense_mod(path, p):
    """dense image mod p of an SMS file (list of rows)"""
    m, n, ent = read_sms(path)
    rp, c, v = to_csr_mod(m, n, ent, p)
    D = [[0] * n for _ in range(m)]
    for i in range(m):
        for k in range(rp[i], rp[i + 1]):
            D[i][c[k]] = v[k]
    return D


def parse_sms_text(text, p=None):
    """dense matrix (list of rows) of an SMS text as the tools print it"""
    lines = [ln for ln in text.splitlines() if ln.strip() and not ln.lstrip().startswith("#")]
    m, n = int(lines[0].split()[0]), int(lines[0].split()[1])
    D = [[0] * n for _ in range(m)]
    for ln in lines[1:]:
        t = ln.split()
        if len(t) < 3 or (t[0] == "0" and t[1] == "0"):
            break
        D[int(t[0]) - 1][int(t[1]) - 1] = int(t[2]) if p is None else int(t[2]) % p
    return D


# ----------------------------------------------------------------------------- trilplacer
TRIL_BASE_SEED = (1 << 64) - 1


def csr_rational(m, n, ent):
    """CSR with rational values: (rowptr, col, num, den), columns sorted per row."""
    rows = [[] for _ in range(m)]
    for (i, j), v in ent.items():
        rows[i].append((j, v))
    rp, col, num, den = [0], [], [], []
    for r in rows:
        r.sort()
        for j, v in r:
            col.append(j); num.append(v.numerator); den.append(v.denominator)
        rp.append(len(col))
    return rp, col, num, den


class OracleTril:
    """The three matrices of an in-place trilinear search handed to the oracle: A, B (m x .) and T = C^T (m x .)."""

    def __init__(self, A, B, C):
        (ma, na, ea), (mb, nb, eb), (mc, nc, ec) = A, B, C
        et = {(j, i): v for (i, j), v in ec.items()}
        self.m = ma
        self.dims = (na, nb, mc)
        self.ent = (ea, eb, et)
        self.csr = [csr_rational(ma, na, ea), csr_rational(mb, nb, eb), csr_rational(nc, mc, et)]
        assert ma == mb == nc

    @classmethod
    def from_sms(cls, pa, pb, pc):
        return cls(read_sms(pa), read_sms(pb), read_sms(pc))

    def _args(self):
        out = [ctypes.c_uint32(self.m)]
        self._keep = []
        for n, (rp, col, num, den) in zip(self.dims, self.csr):
            a = (_arr(rp), _arr(col), _arr(num, ctypes.c_int64), _arr(den, ctypes.c_int64))
            self._keep.append(a)
            out += [ctypes.c_uint32(n), a[0], a[1], a[2], a[3]]
        return out

    def cost_many(self, seeds=None, seed0=0, nseeds=0, expanded=False):
        """-> list of ((ADD,SCA,MUL) oriented, (ADD,SCA,MUL) unoriented); expanded: `trilplacer -e`"""
        if seeds is not None:
            nseeds = len(seeds); sp = _arr(seeds, ctypes.c_uint64)
        else:
            sp = None
        ops = (ctypes.c_uint32 * (6 * max(nseeds, 1)))()
        rc = oracle().plo_oracle_tril_cost_many_x(*self._args(), ctypes.c_int(int(expanded)), sp, ctypes.c_uint64(seed0), ctypes.c_uint64(nseeds), ops)
        assert rc == 0, rc
        return [(tuple(ops[6 * k:6 * k + 3]), tuple(ops[6 * k + 3:6 * k + 6])) for k in range(nseeds)]

    def program(self, seed, variant, expanded=False):
        ops = (ctypes.c_uint32 * 6)()
        txt = ctypes.c_void_p()
        rc = oracle().plo_oracle_tril_program_x(*self._args(), ctypes.c_int(int(expanded)), ctypes.c_uint64(seed), ctypes.c_int(variant), ops, ctypes.byref(txt))
        assert rc == 0, rc
        text = ctypes.string_at(txt).decode()
        oracle().plo_oracle_free(txt)
        return tuple(ops[3 * variant:3 * variant + 3]), text

    def search(self, seed0, nseeds, expanded=False):
        best = (ctypes.c_uint32 * 3)(); bs = ctypes.c_uint64(); bv = ctypes.c_uint32()
        rc = oracle().plo_oracle_tril_search_x(*self._args(), ctypes.c_int(int(expanded)), ctypes.c_uint64(seed0), ctypes.c_uint64(nseeds), best, ctypes.byref(bs), ctypes.byref(bv))
        assert rc == 0, rc
        return tuple(best), bs.value, bv.value


_INPL_LINE = re.compile(r"^([a-z])(\d+):=(.*?);")


class LowHigh:
    """A value p + l*low + h*hig with symbolic low/hig: what an entry of c holds in a `trilplacer -e` program (the
    double-size products are split into a low and a high half, reference plinopt_inplace.h:111, checker :1047)."""

    def __init__(self, p, l=0, h=0):
        self.v = (Fraction(p), Fraction(l), Fraction(h))

    def _w(self, o):
        return o.v if isinstance(o, LowHigh) else (Fraction(o), Fraction(0), Fraction(0))

    def __add__(self, o):
        return LowHigh(*(x + y for x, y in zip(self.v, self._w(o))))

    __radd__ = __add__

    def __sub__(self, o):
        return LowHigh(*(x - y for x, y in zip(self.v, self._w(o))))

    def __neg__(self):
        return LowHigh(*(-x for x in self.v))

    def __mul__(self, f):
        return LowHigh(*(x * f for x in self.v))

    def __truediv__(self, f):
        return LowHigh(*(x / f for x in self.v))

    def __eq__(self, o):
        return self.v == self._w(o)

    def __repr__(self):
        return "LowHigh%r" % (self.v,)


def run_inplace_program(text, a, b, c):
    """Independent interpreter of a trilplacer program (the reference checks these with Maple, -DINPLACE_CHECKER):
    executes `x:=x op y[*v|/d];`, `x:=-x;`, `x:=x*v;`, `x:=x/d;` and `c:=c +- a * b;` on Fractions, in place.
    Returns the operation counts (ADD, SCA, MUL) seen."""
    env = {"a": a, "b": b, "c": c}
    nadd = nsca = nmul = 0

    def operand(tok):
        tok = tok.strip()
        mm = re.match(r"^([a-z])(\d+)(?:([*/])(-?\d+(?:/\d+)?))?$", tok)
        assert mm, tok
        v = env[mm.group(1)][int(mm.group(2))]
        sca = 0
        if mm.group(3):
            f = Fraction(mm.group(4)); sca = 1
            v = v * f if mm.group(3) == "*" else v / f
        return v, sca, (mm.group(1), int(mm.group(2)))

    for line in text.splitlines():
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        mm = _INPL_LINE.match(line)
        assert mm, line
        var, idx, rhs = mm.group(1), int(mm.group(2)), mm.group(3).strip()
        if " * " in rhs and (rhs.endswith(")*low") or rhs.endswith(")*hig")):      # -e: c:=c +/- (a * b)*low|hig; two lines = one AXPY
            mq = re.match(r"^([a-z])(\d+) ([+-]) \(([a-z])(\d+) \* ([a-z])(\d+)\)\*(low|hig)$", rhs)
            assert mq and mq.group(1) == var and int(mq.group(2)) == idx, line
            prod = env[mq.group(4)][int(mq.group(5))] * env[mq.group(6)][int(mq.group(7))]
            if mq.group(3) == "-":
                prod = -prod
            env[var][idx] = env[var][idx] + (LowHigh(0, prod, 0) if mq.group(8) == "low" else LowHigh(0, 0, prod))
            if mq.group(8) == "low":
                nmul += 1
            continue
        if " * " in rhs:                                    # AXPY: c:=c +/- a * b
            mq = re.match(r"^([a-z])(\d+) ([+-]) ([a-z])(\d+) \* ([a-z])(\d+)$", rhs)
            assert mq and mq.group(1) == var and int(mq.group(2)) == idx, line
            prod = env[mq.group(4)][int(mq.group(5))] * env[mq.group(6)][int(mq.group(7))]
            env[var][idx] += prod if mq.group(3) == "+" else -prod
            nmul += 1
            continue
        neg = rhs.startswith("-")
        body = rhs[1:] if neg else rhs
        mq = re.match(r"^([a-z]\d+)([+-])(.*)$", body)
        if mq and not neg:                                   # ADD: x:=x+y*v
            v0, s0, who = operand(mq.group(1))
            assert who == (var, idx) and s0 == 0, line
            v1, s1, _ = operand(mq.group(3))
            env[var][idx] = v0 + v1 if mq.group(2) == "+" else v0 - v1
            nadd += 1; nsca += s1
        else:                                                # SCA: x:=[-]x[*v|/d]
            v0, s0, who = operand(body)
            assert who == (var, idx), line
            env[var][idx] = -v0 if neg else v0
            nsca += 1
    return nadd, nsca, nmul


# ----------------------------------------------------------------------------- cuts of 32x32x32_15096_L
def l32_rows(p=131071):
    """32x32x32_15096_L regenerated from its stored SLP by bin/SLPchecker (reference Makefile:79-80): list of rows,
    each a sorted list of (column, residue)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    out = subprocess.run([os.path.join(ROOT, "bin", "SLPchecker"), "-q", str(p), os.path.join(DATA, "32x32x32_15096_L.slp")],
                         capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    m, n = int(lines[0].split()[0]), int(lines[0].split()[1])
    rows = [[] for _ in range(m)]
    for ln in lines[1:-1]:
        i, j, v = ln.split()
        rows[int(i) - 1].append((int(j) - 1, int(v)))
    for r in rows:
        r.sort()
    return m, n, rows


def l32_cut(row_lo, row_hi, p=131071, rows=None):
    """The row block [row_lo, row_hi) of 32x32x32_15096_L as CSR over Z_p: (m, n, rowptr, col, val)."""
    if rows is None:
        _, n, rows = l32_rows(p)
    else:
        n = 1 + max(c for r in rows for c, _ in r)
    rp, c, v = [0], [], []
    for i in range(row_lo, row_hi):
        for j, x in rows[i]:
            c.append(j)
            v.append(x)
        rp.append(len(c))
    return row_hi - row_lo, n, rp, c, v


def write_sms(path, m, n, rp, c, v):
    with open(path, "w") as f:
        f.write("%d %d M\n" % (m, n))
        for i in range(m):
            for k in range(rp[i], rp[i + 1]):
                f.write("%d %d %d\n" % (i + 1, c[k] + 1, v[k]))
        f.write("0 0 0\n")


def l32_cut_b(p=131071, rows=None):
    """Second cut: rows [0,24) and the first two rows of >= 768 entries of 32x32x32_15096_L, each cut to its first 256
    entries (so that the literal oracle still walks it): rows of up to four 64-lane chunks on the device."""
    if rows is None:
        _, _, rows = l32_rows(p)
    sel = list(range(24)) + [i for i, r in enumerate(rows) if len(r) >= 768][:2]
    rp, c, v = [0], [], []
    for i in sel:
        for j, x in rows[i][:256]:
            c.append(j)
            v.append(x)
        rp.append(len(c))
    return len(sel), 1024, rp, c, v


def longrow_matrix(p=131071, seed=2026, m=9, n=640, block=40):
    """Synthetic matrix with rows BEYOND 512 entries that the literal oracle can still walk (tests/golden/make_longrow_costs.py): m rows
    of ~530-600 entries over n columns that share most of their support -- column blocks of `block`, each present in 5 to 9 of the rows --
    with entries u_i * w_j from four residues (so that a triple (a, b, w_b/w_a) is shared by every row holding both columns: frequencies
    up to m, a first block present in all rows keeps the top level small), 3 % holes and 2 % entries with a ratio of their own."""
    import random
    rng = random.Random(seed)
    inv2 = (p + 1) // 2
    w = [rng.choice([1, 1, 1, p - 1, p - 1, 2, inv2]) for _ in range(n)]
    u = [rng.choice([1, p - 1, 2]) for _ in range(m)]
    rows = [[] for _ in range(m)]
    for k in range(n // block):
        sub = list(range(m)) if k == 0 else rng.sample(range(m), rng.choice([5, 7, 8, 9, 9, 9, 9, 9]))
        for j in range(k * block, (k + 1) * block):
            for i in sub:
                if rng.random() < 0.03:
                    continue
                v = u[i] * w[j] % p
                if rng.random() < 0.02:
                    v = rng.choice([1, p - 1, 2, inv2])
                rows[i].append((j, v))
    rp, c, v = [0], [], []
    for r in rows:
        for j, x in r:
            c.append(j)
            v.append(x)
        rp.append(len(c))
    return m, n, rp, c, v
