// ===========================================================================
// bin/factorizer -- M = Alt . CoB with a prescribed inner dimension (reference src/factorizer.cpp:31-204: `TFactorizer`,
// flags -k # -q # -O # -V [1|0] -b # -c # -U [1|0] -M/-P/-S/-L; `Factorizer` / `backSolver`,
// include/plinopt_sparsify.inl:756-867, :924-984).  The change of basis CoB goes to stdout, the alternative matrix Alt, the
// density profiles and the "SUCCESS: consistent factorization" line to stderr (profileConsistency :132-155 with
// showA = -1, showB = 1).  With -V 1 the matrix is first sparsified (blockSparsifier :667-748), A = M . Cs, then
// M = Alt . Ca and CoB = Ca . Cs (:62-88).  Host only: one factorization is a handful of eliminations of a small matrix
// (it is also the first step of `bin/optimizer -A`, whose restart loop then runs on the GPU).
// Randomness: the build's per-candidate stream (--seed #, default 0); see plo_host.hpp `ab_backsolve` for the rule that
// replaces LinBox's unspecified choices.
// ===========================================================================
#include "plo_sparsify.hpp"

#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace plo;

namespace {
enum Fmt { PRETTY, SMS, MAPLE, LINALG };

template <class F> void write_matrix(std::ostream &os, const F &f, const SparseMat<typename F::Elt> &S, Fmt fmt) {
    if (fmt == SMS) { write_sms(os, f, S, std::is_same<F, QField>::value ? 'R' : 'M'); return; }
    const auto A = to_dense(f, S);
    const size_t r = A.size(), c = S.coldim();
    if (fmt == MAPLE || fmt == LINALG) {
        os << (fmt == MAPLE ? "Matrix(" : "matrix(") << r << ',' << c << ",[";
        for (size_t i = 0; i < r; ++i) { os << (i ? ",[" : "["); for (size_t j = 0; j < c; ++j) { if (j) os << ','; f.write(os, A[i][j]); } os << ']'; }
        os << "])";
        return;
    }
    for (size_t i = 0; i < r; ++i) { os << "  [ "; for (size_t j = 0; j < c; ++j) { f.write(os, A[i][j]); os << ' '; } os << "]\n"; }
}
template <class E> size_t profile_line(std::ostream &os, const char *code, const char *tag, const SparseMat<E> &A, bool colour) {
    size_t s = 0; os << "# " << code << tag << (colour ? "\033[1;36m" : "");
    for (auto &r : A.rows) { s += r.size(); os << r.size() << ' '; }
    os << '=' << s << (colour ? "\033[0m" : "") << std::endl;
    return s;
}
// M == R . C (consistency :872-907)
template <class F> bool consistent(const F &f, const SparseMat<typename F::Elt> &M, const SparseMat<typename F::Elt> &R, const SparseMat<typename F::Elt> &C) {
    using E = typename F::Elt;
    if (M.rowdim() != R.rowdim() || R.coldim() != C.rowdim() || M.coldim() != C.coldim()) return false;
    for (size_t i = 0; i < M.rowdim(); ++i) {
        std::vector<E> acc(M.coldim(), f.zero());
        for (auto &e : R.rows[i]) for (auto &g : C.rows[e.first]) acc[g.first] = f.add(acc[g.first], f.mul(e.second, g.second));
        for (auto &e : M.rows[i]) acc[e.first] = f.add(acc[e.first], f.neg(e.second));
        for (auto &x : acc) if (!f.isZero(x)) return false;
    }
    return true;
}
template <class F> size_t profile_consistency(const F &f, Fmt fmt, double secs, const SparseMat<typename F::Elt> &C, size_t sc, const char *code,
                                              const SparseMat<typename F::Elt> &A, int showA, const SparseMat<typename F::Elt> &B, int showB, bool &ok) {
    const size_t sb = profile_line(std::clog, code, " chgobase profile: ", B, true);
    if (showB) { write_matrix(showB == 1 ? std::cout : std::clog, f, B, fmt); (showB == 1 ? std::cout : std::clog) << std::endl; }
    const size_t sa = profile_line(std::clog, code, " residuum profile: ", A, true);
    if (showA) { write_matrix(showA == 1 ? std::cout : std::clog, f, A, fmt); (showA == 1 ? std::cout : std::clog) << std::endl; }
    ok = consistent(f, C, A, B);
    if (ok) std::clog << "# \033[1;32mSUCCESS: consistent factorization!\033[0m";
    else std::cerr << "# \033[1;31m****** ERROR inconsistency ******\033[0m" << std::endl;
    std::clog << " \033[1;36m" << A.rowdim() << 'x' << A.coldim() << " by " << B.rowdim() << 'x' << B.coldim() << " with " << sa << " non-zeroes (" << sb
              << " alt.) instead of " << sc << "\033[0m: " << secs << "s" << std::endl;
    return sa;
}

// TFactorizer, src/factorizer.cpp:31-99
template <class F> int tfactorizer(const F &f, const SparseMat<typename F::Elt> &A, Fmt fmt, size_t innerdim, size_t loops, uint64_t seed0,
                                   size_t blocksize, size_t maxnumcoeff, bool initialElimination, bool initialSparsification) {
    using E = typename F::Elt;
    const size_t sc = profile_line(std::clog, "[FCTZ]", " Initial profile: ", A, false);
    const size_t kdim = innerdim ? innerdim : A.coldim();
    if (kdim > A.rowdim() || kdim < A.coldim()) {                                           // plinopt_sparsify.inl:936-941
        std::cerr << "# \033[1;36mFail: inner dimension has to be between " << A.coldim() << " and " << A.rowdim() << ".\033[0m\n";
        return -1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    ABFactors<F> ab; bool ok = true, ok1 = true;
    if (initialSparsification) {
        CobHostBackend<F> backend;
        Sparsifier<F> S(f, backend, std::clog);
        DMat<E> Cs, Md;
        S.block_sparsifier(Cs, Md, to_dense(f, A), blocksize, maxnumcoeff, initialElimination);          // A = M . Cs
        const SparseMat<E> M = to_sparse(f, Md), CsS = to_sparse(f, Cs);
        const size_t sm = profile_consistency(f, fmt, secs(), A, sc, "[SPRB]", M, 0, CsS, 0, ok1);
        std::clog << std::string(30, '#') << std::endl;
        ab = ab_factorize(f, M, loops, seed0, kdim);                                                     // M = Alt . Ca
        bool ok2 = true;
        profile_consistency(f, fmt, secs(), M, sm, "[FCTA]", ab.Alt, 0, ab.CoB, 0, ok2);
        std::clog << std::string(30, '#') << std::endl;
        ok1 = ok1 && ok2;
        SparseMat<E> CoB(ab.CoB.rowdim(), A.coldim());                                                   // CoB = Ca . Cs
        for (size_t i = 0; i < ab.CoB.rowdim(); ++i) {
            std::vector<E> acc(A.coldim(), f.zero());
            for (auto &e : ab.CoB.rows[i]) for (auto &g : CsS.rows[e.first]) acc[g.first] = f.add(acc[g.first], f.mul(e.second, g.second));
            for (size_t j = 0; j < acc.size(); ++j) if (!f.isZero(acc[j])) CoB.rows[i].emplace_back(j, acc[j]);
        }
        ab.CoB = CoB;
    } else {
        ab = ab_factorize(f, A, loops, seed0, kdim);
        if (ab.identity) { std::clog << std::string(30, '#') << std::endl; std::clog << "# \033[1;36mWARNING: identity factorization\033[0m\n"; }
    }
    profile_consistency(f, fmt, secs(), A, sc, "[FCTZ]", ab.Alt, -1, ab.CoB, 1, ok);
    return ok && ok1 ? 0 : 1;
}
} // namespace

int main(int argc, char **argv)
{
#ifdef _OPENMP
    if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(std::min(omp_get_max_threads(), 64));
#endif
    Fmt fmt = PRETTY; std::string filename; size_t innerdim = 0, loops = 100, maxnumcoeff = 11, blocksize = 4; uint64_t q = 0, seed0 = 0;
    bool initialSparsification = false, initialElimination = true;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") {
            std::clog << "Usage: " << argv[0] << " [-h|-M|-P|-S|-L|[-k|-O|-c|-b|-U|-V #]] [stdin|matrixfile.sms]\n"
                      << "  -k #: inner dimension (default is column dimension)\n"
                      << "  -M/-P/-S/-L: selects the ouput format\n"
                      << "  -V [1|0]: initial sparsification or not (default 0)\n"
                      << "  -b #: states the blocking dimension (default " << blocksize << ")\n"
                      << "  -c #: max number of coefficients per iteration (default " << maxnumcoeff << ")\n"
                      << "  -U [1|0]: initial LU factorization or not (default 1) \n"
                      << "  -q #: search modulo (default is Rationals)\n"
                      << "  -O #: search for reduced randomized sparsity (default " << loops << " loops)\n"
                      << "  --seed #: first candidate seed (default 0)\n";
            exit(-1);
        } else if (a == "-M") fmt = MAPLE;
        else if (a == "-S") fmt = SMS;
        else if (a == "-P") fmt = PRETTY;
        else if (a == "-L") fmt = LINALG;
        else if (a == "-k" && i + 1 < argc) innerdim = strtoull(argv[++i], nullptr, 10);
        else if (a == "-q" && i + 1 < argc) q = strtoull(argv[++i], nullptr, 10);
        else if (a == "-V" && i + 1 < argc) initialSparsification = atoi(argv[++i]) != 0;
        else if (a == "-b" && i + 1 < argc) blocksize = strtoull(argv[++i], nullptr, 10);
        else if (a == "-c" && i + 1 < argc) maxnumcoeff = strtoull(argv[++i], nullptr, 10);
        else if (a == "-U" && i + 1 < argc) initialElimination = atoi(argv[++i]) != 0;
        else if (a == "-O" && i + 1 < argc) loops = strtoull(argv[++i], nullptr, 10);
        else if (a == "--seed" && i + 1 < argc) seed0 = strtoull(argv[++i], nullptr, 10);
        else filename = a;
    }
    try {
        QMat MQ;
        if (filename.empty()) MQ = read_sms(std::cin);
        else { std::ifstream in(filename); if (!in) return -1; MQ = read_sms(in); }
        if (q != 0) {
            if (q < 2 || q >= (1ull << 62)) { std::cerr << "# ERROR: modulus must be a prime below 2^62 in this build" << std::endl; return -1; }
            if (q >= (1ull << 31)) { Zp64Field f(q); return tfactorizer(f, rebind(MQ, f), fmt, innerdim, loops, seed0, blocksize, maxnumcoeff, initialElimination, initialSparsification); }
            ZpField f((uint32_t)q);
            return tfactorizer(f, rebind(MQ, f), fmt, innerdim, loops, seed0, blocksize, maxnumcoeff, initialElimination, initialSparsification);
        }
        QField f;
        return tfactorizer(f, rebind(MQ, f), fmt, innerdim, loops, seed0, blocksize, maxnumcoeff, initialElimination, initialSparsification);
    } catch (const std::exception &e) {
        std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return -1;
    }
}
