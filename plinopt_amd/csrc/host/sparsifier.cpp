// ===========================================================================
// bin/sparsifier -- front-end keeping the reference CLI contract of
// src/sparsifier.cpp:89-136: flags -q # -c # -b # -U [1|0] -M/-S/-P/-L; the change of basis CoB goes to
// stdout, the residue, the density profiles and the "SUCCESS: consistent factorization" line to stderr (:52,
// plinopt_sparsify.inl:132-155).  New flag: --gpu N (0 = host only).
//
// The exhaustive |Coeffs|^4 enumeration of localSparsifier (plinopt_sparsify.inl:299-314) runs on the GPU through
// plo_cob_search (include/plinopt_hip.h): with -q p modulo p, and without -q (the rationals, the reference's default)
// modulo two 31-bit primes with every winner re-evaluated over Q (CobGpuQBackend).  A failing GPU call is fatal: there is
// no silent fallback; --gpu 0 / --host-q keep the host loops.
// ===========================================================================
#include "plo_sparsify.hpp"
#include "../../../include/plinopt_hip.h"

#include <chrono>
#include <dlfcn.h>
#include <omp.h>
#include <libgen.h>
#include <unistd.h>

using namespace plo;

namespace {

struct HipCob {
    void *h = nullptr;
    decltype(&plo_init) init = nullptr;
    decltype(&plo_last_error) last_error = nullptr;
    decltype(&plo_cob_search) cob_search = nullptr;
    decltype(&plo_cob_search_batch) cob_search_batch = nullptr;
    bool load() {
        std::vector<std::string> cand;
        for (const char *v : {"PLO_HIP_LIB", "PLINOPT_HIP_LIB"}) if (const char *e = getenv(v)) cand.emplace_back(e);   // (one name for the tools and plinopt_amd/capi.py; the older one still works)
        char buf[4096]; ssize_t k = readlink("/proc/self/exe", buf, sizeof buf - 1);
        if (k > 0) { buf[k] = 0; std::string d = dirname(buf); cand.push_back(d + "/../plinopt_amd/libplinopt_hip.so"); cand.push_back(d + "/libplinopt_hip.so"); }
        cand.emplace_back("libplinopt_hip.so");
        for (auto &c : cand) { h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { std::cerr << "# \033[1;31mERROR: cannot load libplinopt_hip.so: " << dlerror() << "\033[0m\n"; return false; }
        init = (decltype(init))dlsym(h, "plo_init"); last_error = (decltype(last_error))dlsym(h, "plo_last_error");
        cob_search = (decltype(cob_search))dlsym(h, "plo_cob_search");
        cob_search_batch = (decltype(cob_search_batch))dlsym(h, "plo_cob_search_batch");
        return init && last_error && cob_search;
    }
};

// An enumeration of fewer candidate rows than this is walked on the host even with --gpu 1: the (block, row) enumerations of a run
// are sequential by definition (each tests independence against every row chosen before, include/plinopt_sparsify.inl:172-175,
// 282-314), so a small one costs a launch + two copies (~0.1 ms) for microseconds of work -- measured on BASELINE configs[2] as
// written, `sparsifier -c 4` (32 enumerations of 256 rows): 0.175 s on the GPU against 0.008 s on the host.  The GPU pays from
// about 12 coefficients (2e4 rows) on; --gpu-min-rows N moves the threshold (0: everything on the GPU).
uint64_t g_gpu_min_rows = 20000;

// GPU backend of the enumeration (Z_p only)
struct CobGpuBackend : CobBackend<ZpField> {
    HipCob &L; double kernel_ms = 0; uint64_t launches = 0, small_on_host = 0;
    CobHostBackend<ZpField> host;
    explicit CobGpuBackend(HipCob &l) : L(l) {}
    CobBest best(const ZpField &f, const DMat<uint32_t> &TM, const DMat<uint32_t> &Cand, size_t row, size_t off,
                 const std::vector<uint32_t> &coeffs, int w0, int w1) override {
        const size_t n = TM.size(), m = n ? TM[0].size() : 0;
        if ((uint64_t)coeffs.size() * coeffs.size() * coeffs.size() * coeffs.size() < g_gpu_min_rows) {
            const uint64_t c0 = host.candidates;
            CobBest r = host.best(f, TM, Cand, row, off, coeffs, w0, w1);
            this->candidates += host.candidates - c0; ++small_on_host;
            return r;
        }
        std::vector<uint32_t> tm(n * m), cd(n * n);
        for (size_t i = 0; i < n; ++i) { std::copy(TM[i].begin(), TM[i].end(), tm.begin() + i * m); std::copy(Cand[i].begin(), Cand[i].end(), cd.begin() + i * n); }
        plo_cob_best_t b{}; plo_stats_t st{};
        int rc = L.cob_search((uint32_t)n, (uint32_t)m, tm.data(), cd.data(), (uint32_t)row, (uint32_t)off, coeffs.data(), (uint32_t)coeffs.size(), f.p, w0, w1, &b, &st);
        if (rc != PLO_OK) throw std::runtime_error(std::string("GPU CoB search failed: ") + L.last_error());
        this->candidates += st.candidates; kernel_ms += st.kernel_ms; launches += st.launches;
        CobBest r; r.zv = b.zeros_v; r.zw = b.zeros_w; r.index = b.index; r.found = b.found != 0;
        return r;
    }
};

// GPU backend of the enumeration over the RATIONALS (the reference's default field for bin/sparsifier, src/sparsifier.cpp:66-83).
// The zero pattern of TM^T w and the independence of w from the rows already chosen do not change when TM, the coefficient set
// and each chosen row are scaled to integers; the enumeration then runs on the device modulo TWO 31-bit primes.  A candidate is
// judged differently modulo p only when p divides a non-zero integer of the computation (a sum of at most four products for
// the zero counts -- impossible when 4 max|w| max|TM| < p, which is checked -- or a minor of the chosen rows for the
// independence test).  The two runs must name the same candidate and that candidate is re-evaluated over Q on the host
// (independence by rank, both zero counts); anything else sends this (block, row) to the host enumeration over Q.
struct CobGpuQBackend : CobBackend<QField> {
    HipCob &L; double kernel_ms = 0; uint64_t fallbacks = 0, gpu_calls = 0, launches = 0, small_on_host = 0;
    CobHostBackend<QField> host;
    explicit CobGpuQBackend(HipCob &l) : L(l) {}
    static int64_t lcm64(int64_t a, int64_t b) { int64_t x = a, y = b; while (y) { int64_t t = x % y; x = y; y = t; } __int128 r = (__int128)a / x * b; if (r > ((__int128)1 << 40)) throw std::overflow_error("denominators too large"); return (int64_t)r; }
    CobBest best(const QField &f, const DMat<Rat> &TM, const DMat<Rat> &Cand, size_t row, size_t off, const std::vector<Rat> &coeffs, int w0, int w1) override {
        const size_t n = TM.size(), m = n ? TM[0].size() : 0, C = coeffs.size();
        static const uint32_t primes[2] = {2147483647u, 2147483629u};
        if ((uint64_t)C * C * C * C < g_gpu_min_rows) {                           // too small to pay for a launch: see g_gpu_min_rows
            const uint64_t c0 = host.candidates;
            CobBest r = host.best(f, TM, Cand, row, off, coeffs, w0, w1);
            this->candidates += host.candidates - c0; ++small_on_host;
            return r;
        }
        try {
            // integer images: TM by one common factor, the coefficients by one common factor, every chosen row by its own
            int64_t dT = 1, dC = 1;
            for (auto &r : TM) for (auto &e : r) dT = lcm64(dT, e.d);
            for (auto &e : coeffs) dC = lcm64(dC, e.d);
            std::vector<__int128> tmi(n * m), cfi(C), cdi(n * n, 0);
            __int128 tmax = 0, cmax = 0;
            for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < m; ++j) { __int128 v = (__int128)TM[i][j].n * (dT / TM[i][j].d); tmi[i * m + j] = v; tmax = std::max(tmax, v < 0 ? -v : v); }
            for (size_t k = 0; k < C; ++k) { __int128 v = (__int128)coeffs[k].n * (dC / coeffs[k].d); cfi[k] = v; cmax = std::max(cmax, v < 0 ? -v : v); }
            for (size_t i = 0; i < row; ++i) { int64_t dr = 1; for (auto &e : Cand[i]) dr = lcm64(dr, e.d); for (size_t j = 0; j < n; ++j) cdi[i * n + j] = (__int128)Cand[i][j].n * (dr / Cand[i][j].d); }
            if (4 * tmax * cmax >= (__int128)primes[1]) throw std::overflow_error("entries too large for an exact zero test modulo a 31-bit prime");
            // both moduli in ONE launch (plo_cob_search_batch: one upload, one kernel, one download)
            plo_cob_best_t b[2]{}; plo_stats_t st[2]{};
            std::vector<uint32_t> tm[2], cd[2], cf[2]; plo_cob_problem_t pr[2];
            for (int q = 0; q < 2; ++q) {
                const __int128 P = primes[q];
                auto red = [&](__int128 v) { __int128 r = v % P; if (r < 0) r += P; return (uint32_t)r; };
                tm[q].resize(n * m); cd[q].resize(n * n); cf[q].resize(C);
                for (size_t k = 0; k < n * m; ++k) tm[q][k] = red(tmi[k]);
                for (size_t k = 0; k < n * n; ++k) cd[q][k] = red(cdi[k]);
                for (size_t k = 0; k < C; ++k) cf[q][k] = red(cfi[k]);
                pr[q] = plo_cob_problem_t{tm[q].data(), cd[q].data(), cf[q].data(), (uint32_t)C, primes[q], w0, w1};
            }
            {
                int rc;
                if (L.cob_search_batch) { rc = L.cob_search_batch(2, (uint32_t)n, (uint32_t)m, (uint32_t)row, (uint32_t)off, pr, b, &st[0]); st[0].candidates /= 2; }
                else { rc = PLO_OK; for (int q = 0; q < 2 && rc == PLO_OK; ++q) { rc = L.cob_search((uint32_t)n, (uint32_t)m, tm[q].data(), cd[q].data(), (uint32_t)row, (uint32_t)off, cf[q].data(), (uint32_t)C, primes[q], w0, w1, &b[q], &st[q]); if (q) { st[0].kernel_ms += st[1].kernel_ms; st[0].launches += st[1].launches; } } }
                if (rc == PLO_E_CAPACITY || rc == PLO_E_UNSUPPORTED) throw std::range_error(std::string("the device refused this enumeration (") + L.last_error() + ")");   // e.g. more than 255 coefficients: the host enumerates, as it does for entries too large for the primes
                if (rc != PLO_OK) throw std::runtime_error(std::string("GPU CoB search failed: ") + L.last_error());
                kernel_ms += st[0].kernel_ms; launches += st[0].launches;
            }
            this->candidates += st[0].candidates; ++gpu_calls;
            if (b[0].found != b[1].found || (b[0].found && (b[0].index != b[1].index || b[0].zeros_v != b[1].zeros_v || b[0].zeros_w != b[1].zeros_w)))
                throw std::runtime_error("the two moduli disagree");
            CobBest r; r.zv = w0; r.zw = w1;
            if (!b[0].found) return r;
            // the winner over Q
            uint64_t idx = b[0].index; size_t ix[4];
            for (int t = 3; t >= 0; --t) { ix[t] = (size_t)(idx % C); idx /= C; }
            std::vector<Rat> w(n, f.zero());
            for (size_t t = 0; t < 4; ++t) if (off + t < n) w[off + t] = coeffs[ix[t]];
            DMat<Rat> prev(Cand.begin(), Cand.begin() + (long)row); prev.push_back(w);
            if (drank(f, prev) != row + 1) throw std::runtime_error("winner is dependent over Q");
            int zv = 0, zw = (int)n;
            for (size_t t = 0; t < 4 && off + t < n; ++t) if (!f.isZero(w[off + t])) --zw;
            for (size_t c = 0; c < m; ++c) { Rat sacc = f.zero(); for (size_t t = 0; t < 4 && off + t < n; ++t) sacc = f.add(sacc, f.mul(w[off + t], TM[off + t][c])); if (f.isZero(sacc)) ++zv; }
            if (zv != b[0].zeros_v || zw != b[0].zeros_w) throw std::runtime_error("winner's zero counts differ over Q");
            r.zv = zv; r.zw = zw; r.index = b[0].index; r.found = true;
            return r;
        } catch (const std::exception &e) {
            if (std::string(e.what()).rfind("GPU CoB search failed", 0) == 0) throw;           // a failing GPU call is fatal, as with -q
            ++fallbacks;
            std::clog << "# CoB over Q on the host for block " << off << ", row " << row << ": " << e.what() << std::endl;
            const uint64_t before = host.candidates;
            CobBest r = host.best(f, TM, Cand, row, off, coeffs, w0, w1);
            this->candidates += host.candidates - before;
            return r;
        }
    }
};

enum Fmt { PRETTY, SMS, MAPLE, LINALG };

template <class F> void write_matrix(std::ostream &os, const F &f, const DMat<typename F::Elt> &A, Fmt fmt) {
    const size_t r = A.size(), c = r ? A[0].size() : 0;
    if (fmt == SMS) { write_sms(os, f, to_sparse(f, A), std::is_same<F, QField>::value ? 'R' : 'M'); return; }
    if (fmt == MAPLE || fmt == LINALG) {
        os << (fmt == MAPLE ? "Matrix(" : "matrix(") << r << ',' << c << ",[";
        for (size_t i = 0; i < r; ++i) { os << (i ? ",[" : "["); for (size_t j = 0; j < c; ++j) { if (j) os << ','; f.write(os, A[i][j]); } os << ']'; }
        os << "])";
        return;
    }
    for (size_t i = 0; i < r; ++i) { os << "  [ "; for (size_t j = 0; j < c; ++j) { f.write(os, A[i][j]); os << ' '; } os << "]\n"; }
}

template <class F> size_t profile_line(std::ostream &os, const F &f, const char *tag, const DMat<typename F::Elt> &A, bool colour) {
    size_t s = 0; os << "# " << tag << (colour ? "\033[1;36m" : "");
    for (auto &r : A) { size_t k = 0; for (auto &e : r) if (!f.isZero(e)) ++k; s += k; os << k << ' '; }
    os << '=' << s << (colour ? "\033[0m" : "") << std::endl;
    return s;
}

// TSparsifier, src/sparsifier.cpp:21-56
template <class F> int tsparsifier(const F &f, const SparseMat<typename F::Elt> &Ms, CobBackend<F> &backend, Fmt fmt, size_t blocksize, size_t maxnumcoeff, bool initialElimination) {
    const auto M = to_dense(f, Ms);
    const size_t sc = profile_line(std::clog, f, "[SPRF] Initial profile: ", M, false);
    std::clog << std::string(30, '#') << std::endl;
    auto t0 = std::chrono::steady_clock::now();
    Sparsifier<F> S(f, backend, std::clog);
    DMat<typename F::Elt> CoB, Res;
    S.block_sparsifier(CoB, Res, M, blocksize, maxnumcoeff, initialElimination);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::clog << std::string(30, '#') << std::endl;
    const size_t sb = profile_line(std::clog, f, "[SPRF] chgobase profile: ", CoB, true);          // profileConsistency :132-155
    write_matrix(std::cout, f, CoB, fmt); std::cout << std::endl;
    const size_t sa = profile_line(std::clog, f, "[SPRF] residuum profile: ", Res, true);
    write_matrix(std::clog, f, Res, fmt); std::clog << std::endl;
    const bool ok = S.consistent(M, Res, CoB);
    if (ok) std::clog << "# \033[1;32mSUCCESS: consistent factorization!\033[0m";
    else std::cerr << "# \033[1;31m****** ERROR inconsistency ******\033[0m" << std::endl;
    std::clog << " \033[1;36m" << Res.size() << 'x' << (Res.empty() ? 0 : Res[0].size()) << " by " << CoB.size() << 'x' << (CoB.empty() ? 0 : CoB[0].size())
              << " with " << sa << " non-zeroes (" << sb << " alt.) instead of " << sc << "\033[0m: " << secs << "s" << std::endl;
    std::clog << "# CoB enumeration: " << backend.candidates << " candidate rows" << std::endl;
    return ok ? 0 : 1;
}

} // namespace

int main(int argc, char **argv)
{
    if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(std::min(omp_get_max_threads(), 64));   // cgroup-limited boxes report all host cores
    Fmt fmt = PRETTY; std::string filename; size_t maxnumcoeff = 11, blocksize = 4; bool initialElimination = true; uint64_t q = 0; int gpu = 1; bool gpu_q = true;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") {
            std::clog << "Usage: " << argv[0] << " [-h|-q|-M|-P|-S|-L|-c #|-b #|-U [1|0]|--gpu #] [stdin|matfile.sms]\n"
                      << "  -c #: max number of coefficients per iteration (default " << maxnumcoeff << ")\n"
                      << "  -b #: states the blocking dimension (default " << blocksize << ")\n"
                      << "  -U [1|0]: initial LU factorization (default) or not\n"
                      << "  -M/-P/-S/-L: selects the ouput format\n"
                      << "  --gpu #: 1 = enumerate the candidate rows on the MI355X (default), 0 = host only\n"
                      << "  --gpu-min-rows #: enumerations of fewer candidate rows stay on the host (default 20000; 0 = all on the GPU)\n"
                      << "  --host-q: without -q (rationals) enumerate on the host (default: on the GPU modulo two primes, winners checked over Q)\n";
            exit(-1);
        } else if (a == "-q" && i + 1 < argc) q = strtoull(argv[++i], nullptr, 10);
        else if (a == "-M") fmt = MAPLE;
        else if (a == "-S") fmt = SMS;
        else if (a == "-P") fmt = PRETTY;
        else if (a == "-L") fmt = LINALG;
        else if (a == "-c" && i + 1 < argc) maxnumcoeff = strtoull(argv[++i], nullptr, 10);
        else if (a == "-b" && i + 1 < argc) blocksize = strtoull(argv[++i], nullptr, 10);
        else if (a == "-U" && i + 1 < argc) initialElimination = atoi(argv[++i]) != 0;
        else if (a == "--gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (a == "--host-q") gpu_q = false;
        else if (a == "--gpu-min-rows" && i + 1 < argc) g_gpu_min_rows = strtoull(argv[++i], nullptr, 10);
        else filename = a;
    }
    try {
        QMat MQ;
        if (filename.empty()) MQ = read_sms(std::cin);
        else { std::ifstream in(filename); if (!in) return -1; MQ = read_sms(in); }
        if (q != 0) {
            if (q < 3 || q >= (1ull << 62)) { std::cerr << "# ERROR: modulus must be an odd prime below 2^62 in this build" << std::endl; return -1; }
            if (q >= (1ull << 31)) {                       // the enumeration kernel holds 31-bit residues: host loops
                std::clog << "# modulus above 2^31: host enumeration" << std::endl;
                Zp64Field f(q); CobHostBackend<Zp64Field> B;
                return tsparsifier(f, rebind(MQ, f), B, fmt, blocksize, maxnumcoeff, initialElimination);
            }
            ZpField f((uint32_t)q);
            if (gpu > 0) {
                HipCob L;
                if (!L.load() || L.init(0) != PLO_OK) { std::cerr << "# \033[1;31mERROR: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return 2; }
                CobGpuBackend B(L);
                int rc = tsparsifier(f, rebind(MQ, f), B, fmt, blocksize, maxnumcoeff, initialElimination);
                std::clog << "# GPU: " << B.launches << " launches, enumeration kernels " << B.kernel_ms << " ms, " << (B.kernel_ms > 0 ? B.candidates / (B.kernel_ms * 1e-3) : 0.0) << " candidate rows/s"
                          << "; " << B.small_on_host << " enumerations of fewer than " << g_gpu_min_rows << " rows on the host (--gpu-min-rows)" << std::endl;
                return rc;
            }
            CobHostBackend<ZpField> B;
            return tsparsifier(f, rebind(MQ, f), B, fmt, blocksize, maxnumcoeff, initialElimination);
        }
        QField f;
        if (gpu > 0 && gpu_q) {
            // over the rationals, as the reference runs it (src/sparsifier.cpp:66-83): the enumeration on the GPU modulo two primes, winners checked over Q
            HipCob L;
            if (!L.load() || L.init(0) != PLO_OK) { std::cerr << "# \033[1;31mERROR: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return 2; }
            CobGpuQBackend B(L);
            int rc = tsparsifier(f, rebind(MQ, f), B, fmt, blocksize, maxnumcoeff, initialElimination);
            std::clog << "# GPU (Q, two 31-bit primes + check over Q): " << B.gpu_calls << " enumerations in " << B.launches << " launches (both moduli of an enumeration in one), kernels " << B.kernel_ms << " ms, " << B.fallbacks << " on the host; "
                      << B.small_on_host << " enumerations of fewer than " << g_gpu_min_rows << " rows on the host (--gpu-min-rows)" << std::endl;
            return rc;
        }
        CobHostBackend<QField> B;
        return tsparsifier(f, rebind(MQ, f), B, fmt, blocksize, maxnumcoeff, initialElimination);
    } catch (const std::exception &e) {
        std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return -1;
    }
}
