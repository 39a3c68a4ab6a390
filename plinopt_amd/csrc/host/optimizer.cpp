// ===========================================================================
// bin/optimizer -- front-end keeping the reference CLI contract of
// src/optimizer.cpp:150-228: flags -D -K -G -A -F -E -N -P -M -q # -v # -O #,
// SLP on stdout only (:87), `#`-prefixed statistics on stderr (:89-98).
// New flags (the reference has none): --gpu N (0 = host only), --seed S.
//
// With -q p the restart loop of CSEOptimiser (include/plinopt_optimize.inl:
// 1204-1238) runs on the GPU through the C-ABI of libplinopt_hip.so
// (plo_cse_search); the winning seed is then replayed on the host to produce
// the program text.  Over the rationals (no -q) the loop runs on the host, as
// in the reference.  A failing GPU call is fatal: there is no silent fallback.
// ===========================================================================
#include "plo_host.hpp"
#include "plo_fast.hpp"
#include "plo_compact.hpp"
#include "../../../include/plinopt_hip.h"

#include <chrono>
#include <memory>
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace plo;
using Ops = std::pair<size_t, size_t>;

namespace {

static int g_failures = 0;   // GPU calls that failed and replays that disagreed with the search: the exit status is non-zero when any happened

struct HipLib {
    void *h = nullptr;
    decltype(&plo_init) init = nullptr;
    decltype(&plo_last_error) last_error = nullptr;
    decltype(&plo_cse_search) cse_search = nullptr;
    decltype(&plo_cse_search_multi) cse_search_multi = nullptr;
    decltype(&plo_cse_chain_create) chain_create = nullptr;
    decltype(&plo_cse_chain_search) chain_search = nullptr;
    decltype(&plo_cse_chain_destroy) chain_destroy = nullptr;
    decltype(&plo_shutdown) shutdown = nullptr;
    decltype(&plo_cse_plan_create) plan_create = nullptr;
    decltype(&plo_cse_plan_destroy) plan_destroy = nullptr;
    decltype(&plo_cse_enum_search_plan) enum_search = nullptr;
    decltype(&plo_cse_chain_batch) chain_batch = nullptr;
    decltype(&plo_kernel_search) kernel_search = nullptr;   // optional: -K with the decompositions on the device
    decltype(&plo_kernel_search_multi) kernel_search_multi = nullptr;   // optional: the same over N devices from this process
    bool load(const char *argv0) {
        std::vector<std::string> cand;
        for (const char *v : {"PLO_HIP_LIB", "PLINOPT_HIP_LIB"}) if (const char *e = getenv(v)) cand.emplace_back(e);   // (one name for the tools and plinopt_amd/capi.py; the older one still works)
        char buf[4096]; ssize_t k = readlink("/proc/self/exe", buf, sizeof buf - 1);
        if (k > 0) { buf[k] = 0; std::string d = dirname(buf); cand.push_back(d + "/../plinopt_amd/libplinopt_hip.so"); cand.push_back(d + "/libplinopt_hip.so"); }
        cand.emplace_back("libplinopt_hip.so");
        for (auto &c : cand) { h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        (void)argv0;
        if (!h) { ++g_failures, std::cerr << "# \033[1;31mERROR: cannot load libplinopt_hip.so: " << dlerror() << "\033[0m\n"; return false; }
        init = (decltype(init))dlsym(h, "plo_init"); last_error = (decltype(last_error))dlsym(h, "plo_last_error");
        cse_search = (decltype(cse_search))dlsym(h, "plo_cse_search"); cse_search_multi = (decltype(cse_search_multi))dlsym(h, "plo_cse_search_multi"); shutdown = (decltype(shutdown))dlsym(h, "plo_shutdown");
        chain_create = (decltype(chain_create))dlsym(h, "plo_cse_chain_create"); chain_search = (decltype(chain_search))dlsym(h, "plo_cse_chain_search");
        chain_destroy = (decltype(chain_destroy))dlsym(h, "plo_cse_chain_destroy");
        plan_create = (decltype(plan_create))dlsym(h, "plo_cse_plan_create"); plan_destroy = (decltype(plan_destroy))dlsym(h, "plo_cse_plan_destroy");
        enum_search = (decltype(enum_search))dlsym(h, "plo_cse_enum_search_plan");
        chain_batch = (decltype(chain_batch))dlsym(h, "plo_cse_chain_batch");
        kernel_search = (decltype(kernel_search))dlsym(h, "plo_kernel_search");
        kernel_search_multi = (decltype(kernel_search_multi))dlsym(h, "plo_kernel_search_multi");
        return init && last_error && cse_search && shutdown && chain_create && chain_search && chain_destroy && plan_create && plan_destroy && enum_search && chain_batch;
    }
};

bool g_fork_shards = false;   // --fork-shards: --gpu N with one forked child per device instead of one thread per device
int g_engine = 0;   // 0 auto (scalable engine over Z_p), 1 literal std::map replay, 2 scalable engine

template <class F> std::string replay_text(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed, Ops &ops) {
    std::ostringstream os;
    input2temps(os, lM, 'i', 't');                          // plinopt_optimize.inl:1211
    if constexpr (std::is_same<F, ZpField>::value) {
        if (g_engine != 1) {
            auto t0 = std::chrono::steady_clock::now();
            SharedIndex S(f, lM);
            auto t1 = std::chrono::steady_clock::now();
            FastCand C(S, seed, &os, 'o', 't', 'r');
            ops = C.optimizer();
            auto t2 = std::chrono::steady_clock::now();
            if (g_engine == 2) std::clog << "# engine: index " << std::chrono::duration<double>(t1 - t0).count() << " s, candidate " << std::chrono::duration<double>(t2 - t1).count()
                << " s; decs " << C.st.decs << ", fresh pairs " << C.st.fresh_inst << " (" << C.st.fresh_distinct << " distinct), level rebuilds " << C.st.rebuilds
                << " (scanned " << C.st.rebuild_scan << "), select scanned " << C.st.select_scan << ", candidate rows " << C.st.cand_rows << ", affected rows " << C.st.aff_rows
                << ", top frequency " << C.st.max_level0 << ", nnz at ProgramGen " << C.st.live_nnz_end << ", columns " << C.st.cols_end << ", sum of tie-set sizes " << C.st.sum_T << " (max " << C.st.max_T << "), steps at level 2: " << C.st.steps_l2 << " (sum T " << C.st.sum_T_l2 << "), steps at level<=4: " << C.st.steps_le4 << ", largest fresh batch " << C.st.max_fresh << std::endl;
            if (g_engine == 2) std::clog << "# engine: scalable, " << C.steps() << " CSE steps, " << S.keys.size() << " initial triples, " << S.pairs0 << " pair instances" << std::endl;
            return os.str();
        }
    }
    Replay<F> R(f, lM, seed, os, 'o', 't', 'r');
    ops = R.optimizer();                                    // :1212
    return os.str();
}

// host restart loop (rationals, or --gpu 0): candidates seed0..seed0+loops-1, total order (cmpOpCount, seed)
template <class F> bool host_search(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed0, size_t loops, Ops &best, uint64_t &bseed) {
    bool have = false;
    std::unique_ptr<SharedIndex> S;
    if constexpr (std::is_same<F, ZpField>::value) if (g_engine != 1) S.reset(new SharedIndex(f, lM));
#pragma omp parallel for schedule(dynamic)
    for (long long k = 0; k < (long long)loops; ++k) {
        std::ostringstream sink; Ops ops;
        if (S) { FastCand C(*S, seed0 + (uint64_t)k); ops = C.optimizer(); }
        else { Replay<F> R(f, lM, seed0 + (uint64_t)k, sink, 'o', 't', 'r'); ops = R.optimizer(); }
#pragma omp critical
        {
            uint64_t s = seed0 + (uint64_t)k;
            if (!have || cmp_op_count(ops, best) || (!cmp_op_count(best, ops) && s < bseed)) { best = ops; bseed = s; have = true; }
        }
    }
    return have;
}


// program text of the LU method for one seed (include/plinopt_optimize.inl:1058-1079)
template <class F> std::string lu_text(const F &f, const LUFactors<F> &lu, uint64_t seed, Ops &ops) {
    std::ostringstream os;
    for (size_t k = 0; k < lu.P.size(); ++k) os << 't' << k << ":=" << 'i' << lu.P[k] << ";\n";          // :1064
    CandRng rng(seed);
    Replay<F> RU(f, lu.U, rng, os, 'v', 't', 'r'); Ops uo = RU.optimizer();                                 // :1068
    Replay<F> RL(f, lu.L, rng, os, 'x', 'v', 'g'); Ops lo = RL.optimizer();                                 // :1072
    for (size_t i = 0; i < lu.Q.size(); ++i) os << 'o' << i << ":=" << 'x' << lu.Q[i] << ";\n";          // :1076
    ops = {uo.first + lo.first, uo.second + lo.second};
    return os.str();
}

template <class E> void to_csr(const SparseMat<E> &A, std::vector<uint32_t> &rp, std::vector<uint32_t> &cc, std::vector<uint32_t> &vv) {
    rp.assign(1, 0); cc.clear(); vv.clear();
    for (auto &r : A.rows) { for (auto &e : r) { cc.push_back((uint32_t)e.first); vv.push_back((uint32_t)e.second); } rp.push_back((uint32_t)cc.size()); }
}

// LUOptimiser :1021-1109.  Returns false when the method could not run (reported on stderr).
template <class F> bool lu_method(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed0, size_t loops, int gpu, uint32_t q,
                                  int verbose, Ops &gops, std::string &gtext, const char *argv0) {
    const LUFactors<F> lu = sparse_lu(f, lM);
    uint64_t seed = 0; Ops best; bool have = false;
    if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0) {
        HipLib L;
        if (!L.load(argv0) || L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -G: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return false; }
        std::vector<uint32_t> rp1, c1, v1, rp2, c2, v2;
        to_csr(lu.U, rp1, c1, v1); to_csr(lu.L, rp2, c2, v2);
        plo_csr_t A{(uint32_t)lu.U.rowdim(), (uint32_t)lu.U.coldim(), rp1.data(), c1.data(), v1.data()};
        plo_csr_t B{(uint32_t)lu.L.rowdim(), (uint32_t)lu.L.coldim(), rp2.data(), c2.data(), v2.data()};
        plo_chain_t *ch = nullptr;
        int rc = L.chain_create(&A, &B, q, &ch);
        if (rc != PLO_OK) { std::clog << "# -G skipped: " << L.last_error() << std::endl; return false; }
        plo_best_t b{}; plo_stats_t st{};
        rc = L.chain_search(ch, seed0, loops, PLO_COST_SUM_THEN_ADD, &b, &st);
        L.chain_destroy(ch);
        if (rc != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -G GPU search failed: " << L.last_error() << "\033[0m" << std::endl; return false; }
        best = {b.adds, b.muls}; seed = b.seed; have = loops > 0;
        if (verbose > 0) std::clog << "# GPU (LU): " << st.candidates << " candidates, kernel " << st.kernel_ms << " ms" << std::endl;
    }
    if (!have) {
#pragma omp parallel for schedule(dynamic)
        for (long long k = 0; k < (long long)loops; ++k) {
            Ops ops; (void)lu_text(f, lu, seed0 + (uint64_t)k, ops);
#pragma omp critical
            { uint64_t s = seed0 + (uint64_t)k; if (!have || cmp_op_count(ops, best) || (!cmp_op_count(best, ops) && s < seed)) { best = ops; seed = s; have = true; } }
        }
    }
    if (!have) return false;
    Ops rops; std::string t = lu_text(f, lu, seed, rops);
    if (rops != best) { ++g_failures, std::cerr << "# \033[1;31mERROR: -G replay of seed " << seed << " gives " << rops.first << '|' << rops.second << ", search said " << best.first << '|' << best.second << "\033[0m" << std::endl; return false; }
    if (verbose > 0) std::clog << "# Found G: " << best.first << '|' << best.second << " instead of " << gops.first << '|' << gops.second << "\t[seed " << seed << "] (rank " << lu.rank << ')' << std::endl;
    if (cmp_op_count(best, gops)) { gops = best; gtext = t; }                                               // :1103-1107
    return true;
}

// program text of the alternative-factorization method for one seed (include/plinopt_optimize.inl:1142-1153)
template <class F> std::string ab_text(const F &f, const ABFactors<F> &ab, uint64_t seed, Ops &ops) {
    std::ostringstream os;
    input2temps(os, ab.CoB, 'i', 't');                                                                      // :1146
    CandRng rng(seed);
    Replay<F> RC(f, ab.CoB, rng, os, 'v', 't', 'r'); Ops bo = RC.optimizer();                               // :1150
    Replay<F> RA(f, ab.Alt, rng, os, 'o', 'v', 'g'); Ops ao = RA.optimizer();                               // :1153
    ops = {bo.first + ao.first, bo.second + ao.second};
    return os.str();
}

// ABOptimiser :1114-1186 for the inner dimension coldim(M).  Returns false when the method could not run.
template <class F> bool ab_method(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed0, size_t loops, int gpu, uint32_t q,
                                  int verbose, Ops &gops, std::string &gtext, const char *argv0, size_t innerdim = 0) {
    if (lM.rowdim() < lM.coldim()) { std::clog << "# -A skipped: fewer rows than columns" << std::endl; return false; }
    const ABFactors<F> ab = ab_factorize(f, lM, 1 + (loops >> 3), seed0, innerdim);                         // :1129-1130
    uint64_t seed = 0; Ops best; bool have = false;
    if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0) {
        HipLib L;
        if (!L.load(argv0) || L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -A: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return false; }
        std::vector<uint32_t> rp1, c1, v1, rp2, c2, v2;
        to_csr(ab.CoB, rp1, c1, v1); to_csr(ab.Alt, rp2, c2, v2);
        plo_csr_t A{(uint32_t)ab.CoB.rowdim(), (uint32_t)ab.CoB.coldim(), rp1.data(), c1.data(), v1.data()};
        plo_csr_t B{(uint32_t)ab.Alt.rowdim(), (uint32_t)ab.Alt.coldim(), rp2.data(), c2.data(), v2.data()};
        plo_chain_t *ch = nullptr;
        int rc = L.chain_create(&A, &B, q, &ch);
        if (rc != PLO_OK) { std::clog << "# -A skipped: " << L.last_error() << std::endl; return false; }
        plo_best_t b{}; plo_stats_t st{};
        rc = L.chain_search(ch, seed0, loops, PLO_COST_SUM_THEN_ADD, &b, &st);
        L.chain_destroy(ch);
        if (rc != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -A GPU search failed: " << L.last_error() << "\033[0m" << std::endl; return false; }
        best = {b.adds, b.muls}; seed = b.seed; have = loops > 0;
        if (verbose > 0) std::clog << "# GPU (A): " << st.candidates << " candidates, kernel " << st.kernel_ms << " ms" << std::endl;
    }
    if (!have) {
#pragma omp parallel for schedule(dynamic)
        for (long long k = 0; k < (long long)loops; ++k) {
            Ops ops; (void)ab_text(f, ab, seed0 + (uint64_t)k, ops);
#pragma omp critical
            { uint64_t s = seed0 + (uint64_t)k; if (!have || cmp_op_count(ops, best) || (!cmp_op_count(best, ops) && s < seed)) { best = ops; seed = s; have = true; } }
        }
    }
    if (!have) return false;
    Ops rops; std::string t = ab_text(f, ab, seed, rops);
    if (rops != best) { ++g_failures, std::cerr << "# \033[1;31mERROR: -A replay of seed " << seed << " gives " << rops.first << '|' << rops.second << ", search said " << best.first << '|' << best.second << "\033[0m" << std::endl; return false; }
    if (verbose > 0) std::clog << "# Found A: (" << ab.Alt.rowdim() << 'x' << ab.Alt.coldim() << 'x' << ab.CoB.coldim() << ' ' << ab.score[0] << '/' << ab.score[2] << ")\t"
                               << best.first << '|' << best.second << " instead of " << gops.first << '|' << gops.second << "\t[seed " << seed << ']' << std::endl;
    if (cmp_op_count(best, gops)) { gops = best; gtext = t; }                                               // :1180-1184
    return true;
}

// program text of the kernel method for one decomposition and one seed (include/plinopt_optimize.inl:836-872)
template <class F> std::string kernel_text(const F &f, const KernelDecomp<F> &kd, uint64_t seed, Ops &ops) {
    std::ostringstream os;
    input2temps(os, kd.Free, 'i', 't');                                                                    // :836
    CandRng rng(seed);
    Replay<F> RF(f, kd.Free, rng, os, 'o', 't', 'r'); Ops fo = RF.optimizer();                             // :841
    input2temps(os, kd.Dep, 'o', 'v');                                                                     // :859
    Replay<F> RK(f, kd.Dep, rng, os, 'x', 'v', 'g'); Ops ko = RK.optimizer();                              // :864
    for (size_t j = 0; j < kd.dep.size(); ++j) os << 'o' << kd.dep[j] << ":=" << 'x' << j << ";\n";       // :871-874
    ops = {fo.first + ko.first, fo.second + ko.second};
    return os.str();
}

// KernelOptimiser :1288-1353 with this build's decomposition rule.  The restarts are spent as blocks of PLO_KERNEL_BLOCK
// seeds: block d uses the decomposition drawn from its first seed, and every seed of the block is one run of the two
// Optimizer calls on it (the reference draws a new decomposition for every restart; with a per-restart elimination on
// the host the GPU would idle).  Returns false when the method could not run.
struct KShard { bool done = false; Ops ops; uint64_t seed = 0, ncand = 0; double kms = 0; int shards = 0; std::string how; } g_kshard;   // -K searched by the --gpu N shards (run(): plo_kernel_search_multi, or forked children with --fork-shards)
bool g_host_decomp = false;        // --host-decomp: -K eliminates on the host and ships the images (plo_cse_chain_batch)
uint64_t g_kernel_block = 1;      // restarts per decomposition (--kernel-block; the reference draws one decomposition per restart, :1299-1340)
#define PLO_KERNEL_BLOCK g_kernel_block
// restarts s0 .. s0+cnt-1 of the two Optimizer calls on one decomposition: GPU (chained-candidate kernel) or host loop
template <class F> bool kernel_block_serial(const F &f, const KernelDecomp<F> &kd, uint64_t s0, uint64_t cnt, Ops &bops, uint64_t &bs) {
    bool have = false;
    for (uint64_t k = 0; k < cnt; ++k) {
        Ops ops; (void)kernel_text(f, kd, s0 + k, ops);
        if (!have || cmp_op_count(ops, bops)) { bops = ops; bs = s0 + k; have = true; }        // increasing seeds: the first one keeps ties
    }
    return have;
}
template <class F> bool kernel_block(const F &f, const KernelDecomp<F> &kd, uint64_t s0, uint64_t cnt, HipLib *L, uint32_t q, Ops &bops, uint64_t &bs, double &kms, uint64_t &ncand) {
    bool bhave = false;
    if constexpr (std::is_same<F, ZpField>::value) if (L) {
        std::vector<uint32_t> rp1, c1, v1, rp2, c2, v2;
        to_csr(kd.Free, rp1, c1, v1); to_csr(kd.Dep, rp2, c2, v2);
        plo_csr_t A{(uint32_t)kd.Free.rowdim(), (uint32_t)kd.Free.coldim(), rp1.data(), c1.data(), v1.data()};
        plo_csr_t B{(uint32_t)kd.Dep.rowdim(), (uint32_t)kd.Dep.coldim(), rp2.data(), c2.data(), v2.data()};
        plo_chain_t *ch = nullptr;
        if (L->chain_create(&A, &B, q, &ch) != PLO_OK) throw std::runtime_error(std::string("chain plan: ") + L->last_error());
        plo_best_t b{}; plo_stats_t st{};
        const int rc = L->chain_search(ch, s0, cnt, PLO_COST_SUM_THEN_ADD, &b, &st);
        L->chain_destroy(ch);
        if (rc != PLO_OK) throw std::runtime_error(std::string("GPU search failed: ") + L->last_error());
        bops = {b.adds, b.muls}; bs = b.seed; kms += st.kernel_ms; ncand += st.candidates;
        return true;
    }
#pragma omp parallel for schedule(dynamic)
    for (long long k = 0; k < (long long)cnt; ++k) {
        Ops ops; (void)kernel_text(f, kd, s0 + (uint64_t)k, ops);
#pragma omp critical
        { uint64_t s = s0 + (uint64_t)k; if (!bhave || cmp_op_count(ops, bops) || (!cmp_op_count(bops, ops) && s < bs)) { bops = ops; bs = s; bhave = true; } }
    }
    (void)q;
    return bhave;
}


// restarts of many decompositions in ONE launch (plo_cse_chain_batch): decomposition j takes the seeds s0 + j*per .. + per - 1
template <class F> bool kernel_batch_gpu(HipLib &L, const std::vector<KernelDecomp<F>> &kds, uint64_t s0, uint32_t per, uint32_t q,
                                         Ops &bops, uint64_t &bs, double &kms, uint64_t &ncand) {
    if constexpr (std::is_same<F, ZpField>::value) {
        const size_t np = kds.size();
        std::vector<std::vector<uint32_t>> rp(2 * np), cc(2 * np), vv(2 * np);
        std::vector<plo_csr_t> A(np), B(np);
        for (size_t j = 0; j < np; ++j) {
            to_csr(kds[j].Free, rp[2 * j], cc[2 * j], vv[2 * j]); to_csr(kds[j].Dep, rp[2 * j + 1], cc[2 * j + 1], vv[2 * j + 1]);
            A[j] = plo_csr_t{(uint32_t)kds[j].Free.rowdim(), (uint32_t)kds[j].Free.coldim(), rp[2 * j].data(), cc[2 * j].data(), vv[2 * j].data()};
            B[j] = plo_csr_t{(uint32_t)kds[j].Dep.rowdim(), (uint32_t)kds[j].Dep.coldim(), rp[2 * j + 1].data(), cc[2 * j + 1].data(), vv[2 * j + 1].data()};
        }
        plo_best_t b{}; plo_stats_t st{};
        if (L.chain_batch((uint32_t)np, A.data(), B.data(), q, s0, per, PLO_COST_SUM_THEN_ADD, nullptr, nullptr, &b, &st) != PLO_OK)
            throw std::runtime_error(std::string("batched chain search: ") + L.last_error());
        bops = {b.adds, b.muls}; bs = b.seed; kms += st.kernel_ms; ncand += st.candidates;
        return true;
    } else { (void)L; (void)kds; (void)s0; (void)per; (void)q; (void)bops; (void)bs; (void)kms; (void)ncand; return false; }
}

template <class F> bool kernel_method(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed0, size_t loops, int gpu, uint32_t q,
                                      int verbose, Ops &gops, std::string &gtext, const char *argv0) {
    uint64_t seed = 0, bdec = 0; Ops best; bool have = false, on_device = false; double kms = 0; uint64_t ncand = 0;
    const uint64_t nblocks = (loops + PLO_KERNEL_BLOCK - 1) / PLO_KERNEL_BLOCK;
    bool use_gpu = false;
    HipLib L;
    if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0 && !g_kshard.done) {
        if (!L.load(argv0) || L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -K: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return false; }
        use_gpu = true;
    }
    const uint64_t full = loops / PLO_KERNEL_BLOCK;                    // full blocks go to the GPU in batches of one launch each
    const uint64_t BATCH = 4096;
    uint64_t d = 0;
    if (g_kshard.done) {                                               // searched by N forked shards, one device each (main): only the replay is left
        best = g_kshard.ops; seed = g_kshard.seed; bdec = seed; have = true; kms = g_kshard.kms; ncand = g_kshard.ncand; d = nblocks; on_device = true;
        g_kshard.done = false;
        if (verbose > 0) std::clog << "# " << g_kshard.shards << " shards (" << g_kshard.how << "): -K" << std::endl;
    }
    // All of it on the device (plo_kernel_search: the wave of a restart eliminates, builds both images and runs both
    // Optimizer calls): matrices of at most 128 rows and 64 columns with at most 64 dependent rows.  --host-decomp keeps the decompositions on the host (round-1 path).
    if constexpr (std::is_same<F, ZpField>::value) if (!have && use_gpu && L.kernel_search && !g_host_decomp && lM.rowdim() <= 128 && lM.coldim() <= 64) {
        KernelDecomp<F> kd0;
        if (!kernel_decomp(f, lM, seed0, kd0)) { std::clog << "# \033[1;36mZero dimensional kernel.\033[0m" << std::endl; return false; }   // :1343-1346 (the rank does not depend on the order)
        std::vector<uint32_t> rp, cc, vv; to_csr(lM, rp, cc, vv);
        plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
        plo_best_t b{}; plo_stats_t st{};
        const int rc = L.kernel_search(&A, q, seed0, loops, (uint32_t)std::min<uint64_t>(PLO_KERNEL_BLOCK, 0xFFFFFFFFull), PLO_COST_SUM_THEN_ADD, nullptr, nullptr, nullptr, &b, &st);
        if (rc == PLO_OK) {
            best = {b.adds, b.muls}; seed = b.seed; bdec = seed0 + ((b.seed - seed0) / PLO_KERNEL_BLOCK) * PLO_KERNEL_BLOCK; have = true;
            kms = st.kernel_ms; ncand = st.candidates; d = nblocks; on_device = true;
        } else if (rc != PLO_E_UNSUPPORTED && rc != PLO_E_CAPACITY) {
            ++g_failures; std::cerr << "# \033[1;31mERROR: -K on the GPU: " << L.last_error() << "\033[0m" << std::endl; return false;
        }
    }
    if (use_gpu) for (; d < full; ) {
        const uint64_t nb = std::min<uint64_t>(BATCH, full - d), s0 = seed0 + d * PLO_KERNEL_BLOCK;
        std::vector<KernelDecomp<F>> kds(nb); bool zero = false;
#pragma omp parallel for schedule(dynamic, 8)
        for (long long j = 0; j < (long long)nb; ++j) if (!kernel_decomp(f, lM, s0 + (uint64_t)j * PLO_KERNEL_BLOCK, kds[(size_t)j])) {
#pragma omp atomic write
            zero = true;
        }
        if (zero) { std::clog << "# \033[1;36mZero dimensional kernel.\033[0m" << std::endl; return false; }              // :1343-1346
        Ops bops; uint64_t bs = 0;
        try { kernel_batch_gpu(L, kds, s0, (uint32_t)PLO_KERNEL_BLOCK, q, bops, bs, kms, ncand); }
        catch (const std::exception &e) { ++g_failures; std::cerr << "# \033[1;31mERROR: -K on the GPU: " << e.what() << "\033[0m" << std::endl; return false; }
        if (!have || cmp_op_count(bops, best)) { best = bops; seed = bs; bdec = seed0 + ((bs - seed0) / PLO_KERNEL_BLOCK) * PLO_KERNEL_BLOCK; have = true; }
        d += nb;
    }
    if (!use_gpu) {                                                    // host path: the blocks are shared out between the threads
        bool zero = false;
#pragma omp parallel for schedule(dynamic, 1)
        for (long long dd = (long long)d; dd < (long long)nblocks; ++dd) {
            const uint64_t s0 = seed0 + (uint64_t)dd * PLO_KERNEL_BLOCK, cnt = std::min<uint64_t>(PLO_KERNEL_BLOCK, loops - (uint64_t)dd * PLO_KERNEL_BLOCK);
            KernelDecomp<F> kd; Ops bops; uint64_t bs = 0;
            if (!kernel_decomp(f, lM, s0, kd)) {
#pragma omp atomic write
                zero = true;
                continue;
            }
            const bool bhave = kernel_block_serial(f, kd, s0, cnt, bops, bs);
#pragma omp critical
            if (bhave && (!have || cmp_op_count(bops, best) || (!cmp_op_count(best, bops) && bs < seed))) { best = bops; seed = bs; bdec = s0; have = true; }
        }
        if (zero) { std::clog << "# \033[1;36mZero dimensional kernel.\033[0m" << std::endl; return false; }              // :1343-1346
        d = nblocks;
    }
    for (; d < nblocks; ++d) {                                         // GPU: the last partial block
        const uint64_t s0 = seed0 + d * PLO_KERNEL_BLOCK, cnt = std::min<uint64_t>(PLO_KERNEL_BLOCK, loops - d * PLO_KERNEL_BLOCK);
        KernelDecomp<F> kd;
        if (!kernel_decomp(f, lM, s0, kd)) { std::clog << "# \033[1;36mZero dimensional kernel.\033[0m" << std::endl; return false; }   // :1343-1346
        Ops bops; uint64_t bs = 0;
        bool bhave;
        try { bhave = kernel_block(f, kd, s0, cnt, &L, q, bops, bs, kms, ncand); }
        catch (const std::exception &e) { ++g_failures; std::cerr << "# \033[1;31mERROR: -K on the GPU: " << e.what() << "\033[0m" << std::endl; return false; }
        if (bhave && (!have || cmp_op_count(bops, best))) { best = bops; seed = bs; bdec = s0; have = true; }      // earlier block wins ties
    }
    if (!have) return false;
    KernelDecomp<F> kd;
    if (!kernel_decomp(f, lM, bdec, kd)) return false;
    Ops rops; std::string t = kernel_text(f, kd, seed, rops);
    if (rops != best) { ++g_failures, std::cerr << "# \033[1;31mERROR: -K replay of seed " << seed << " gives " << rops.first << '|' << rops.second << ", search said " << best.first << '|' << best.second << "\033[0m" << std::endl; return false; }
    if ((use_gpu || on_device) && verbose > 0) std::clog << "# GPU (K): " << ncand << " candidates on " << nblocks << " decompositions, kernel " << kms << " ms" << (on_device ? " (decompositions on the device)" : " (decompositions on the host)") << std::endl;
    if (verbose > 0) std::clog << "# Found K: " << best.first << '|' << best.second << " instead of " << gops.first << '|' << gops.second << "\t[seed " << seed
                               << "] (rank " << kd.rank << '+' << kd.notindep << ", " << kd.dep.size() << " dependent rows)" << std::endl;
    if (cmp_op_count(best, gops)) { gops = best; gtext = t; }                                              // :1347-1351
    return true;
}

// AllKernelOpt (-N, :1357-1418): every order of the rows (m <= 9 here: m! eliminations on the host; the reference allows 12).
// Orders that give the same decomposition (same computed rows, same combinations) are searched once; the restarts are
// shared out between the distinct decompositions.
template <class F> bool allkernels_method(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t seed0, size_t loops, int gpu, uint32_t q,
                                          int verbose, Ops &gops, std::string &gtext, const char *argv0) {
    const size_t m = lM.rowdim();
    if (m > 9) { std::clog << "# -N skipped: " << m << "! row orders (this build walks up to 9!)" << std::endl; return false; }
    std::vector<size_t> ord(m);
    for (size_t i = 0; i < m; ++i) ord[i] = i;
    std::map<std::vector<size_t>, std::pair<std::vector<size_t>, uint64_t>> distinct;     // signature -> (order, permutation index)
    uint64_t pi = 0;
    do {
        KernelDecomp<F> kd; CandRng rng(seed0 + pi);
        if (!kernel_decomp_order(f, lM, ord, rng, kd)) { std::clog << "# \033[1;36mZero dimensional kernel.\033[0m" << std::endl; return false; }
        std::vector<size_t> sig(kd.dep); sig.push_back(m + kd.notindep);
        for (auto &r : kd.Dep.rows) for (auto &e : r) sig.push_back(e.first);             // the basis rows used
        distinct.emplace(sig, std::make_pair(ord, pi));
        ++pi;
    } while (std::next_permutation(ord.begin(), ord.end()));
    bool use_gpu = false; HipLib L;
    if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0) {
        if (!L.load(argv0) || L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -N: cannot use the GPU\033[0m" << std::endl; return false; }
        use_gpu = true;
    }
    const uint64_t per = std::max<uint64_t>(1, loops / distinct.size());
    Ops best; uint64_t seed = 0, bpi = 0; std::vector<size_t> bord; bool have = false; double kms = 0; uint64_t ncand = 0;
    std::vector<const std::pair<const std::vector<size_t>, std::pair<std::vector<size_t>, uint64_t>> *> order;
    for (auto &kv : distinct) order.push_back(&kv);
    if (use_gpu) {                                                   // all distinct decompositions in one launch
        std::vector<KernelDecomp<F>> kds(order.size());
        for (size_t k = 0; k < order.size(); ++k) { CandRng rng(seed0 + order[k]->second.second); kernel_decomp_order(f, lM, order[k]->second.first, rng, kds[k]); }
        Ops bops; uint64_t bs = 0;
        try { kernel_batch_gpu(L, kds, seed0, (uint32_t)per, q, bops, bs, kms, ncand); }
        catch (const std::exception &e) { std::clog << "# -N skipped: " << e.what() << std::endl; return false; }
        const size_t w = (size_t)((bs - seed0) / per);
        best = bops; seed = bs; bord = order[w]->second.first; bpi = order[w]->second.second; have = true;
    } else {
#pragma omp parallel for schedule(dynamic, 1)
        for (long long k = 0; k < (long long)order.size(); ++k) {
            const auto &kv = *order[(size_t)k];
            KernelDecomp<F> kd; CandRng rng(seed0 + kv.second.second);
            kernel_decomp_order(f, lM, kv.second.first, rng, kd);
            Ops bops; uint64_t bs = 0;
            const bool bhave = kernel_block_serial(f, kd, seed0 + (uint64_t)k * per, per, bops, bs);
#pragma omp critical
            if (bhave && (!have || cmp_op_count(bops, best) || (!cmp_op_count(best, bops) && bs < seed))) { best = bops; seed = bs; bord = kv.second.first; bpi = kv.second.second; have = true; }
        }
    }
    if (!have) return false;
    KernelDecomp<F> kd; CandRng rng(seed0 + bpi);
    kernel_decomp_order(f, lM, bord, rng, kd);
    Ops rops; std::string t = kernel_text(f, kd, seed, rops);
    if (rops != best) { ++g_failures, std::cerr << "# \033[1;31mERROR: -N replay gives " << rops.first << '|' << rops.second << ", search said " << best.first << '|' << best.second << "\033[0m" << std::endl; return false; }
    if (verbose > 0) std::clog << "# Found N: " << best.first << '|' << best.second << " instead of " << gops.first << '|' << gops.second << "\t[order " << bpi << ", seed " << seed << "] ("
                               << pi << " row orders, " << distinct.size() << " distinct decompositions, " << per << " restarts each" << (use_gpu ? ", GPU kernel " : ", host ") ;
    if (verbose > 0) { if (use_gpu) std::clog << kms << " ms"; std::clog << ')' << std::endl; }
    if (cmp_op_count(best, gops)) { gops = best; gtext = t; }
    else std::clog << "# \033[1;36mNo kernel permutation has less additions.\033[0m" << std::endl;        // :1409-1413
    return true;
}

// program text of one schedule of the exhaustive CSE tree
template <class F> std::string schedule_text(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t index, Ops &ops, uint64_t &prod, size_t *recsub_muls = nullptr) {
    std::ostringstream os;
    input2temps(os, lM, 'i', 't');                                                                         // :1265
    Replay<F> R(f, lM, 0, os, 'o', 't', 'r'); R.set_schedule(index);
    ops = R.optimizer(); prod = R.eprod;
    if (recsub_muls) *recsub_muls = R.recsub_muls;
    return os.str();
}

// AllCSEOpt / RecOptimizer / RecSub (-E, include/plinopt_optimize.inl:889-1013, :1252-1281): the best of ALL greedy CSE
// schedules (every pair of frequency > 1 is a child, order additions then multiplications :958-959).  The tree is walked
// by schedule index (include/plinopt_hip.h): ranges 0..N-1 with N growing to the largest radix product seen; exhaustive
// when N reaches it, otherwise stopped at `budget` schedules (the reference has no bound and no termination on anything
// but toy inputs).  Returns false when the method could not run.
bool g_recsub_order = false;   // --recsub: choose the schedule as RecSub does (:950-959): additions, then ITS multiplication count (before ProgramGen)
template <class F> bool exhaustive_method(const F &f, const SparseMat<typename F::Elt> &lM, uint64_t budget, int gpu, uint32_t q,
                                          int verbose, Ops &gops, std::string &gtext, const char *argv0) {
    Ops best; uint64_t bidx = 0, maxprod = 1, done = 0; bool have = false, on_gpu = false; double kms = 0;
    // `best` holds the SELECTION key: (adds, muls of the program), or with --recsub (adds, RecSub's multiplication count)
    auto better = [](const Ops &a, const Ops &b) { return cmp_op_count(a, b, 1); };
    HipLib L; plo_plan_t *plan = nullptr;
    std::vector<uint32_t> rp, cc, vv;
    if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0) {
        if (!L.load(argv0) || L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: -E: cannot use the GPU: " << (L.last_error ? L.last_error() : "library missing") << "\033[0m" << std::endl; return false; }
        to_csr(lM, rp, cc, vv);
        plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
        if (L.plan_create(&A, q, &plan) != PLO_OK) { std::clog << "# -E skipped: " << L.last_error() << std::endl; return false; }
        on_gpu = true;
    }
    uint64_t target = 1;
    while (done < target) {
        const uint64_t cnt = target - done;
        Ops rb; uint64_t ri = 0, rp_ = 1; bool rh = false;
        if (on_gpu) {
            plo_best_t b{}; plo_stats_t st{}; uint64_t mp = 0;
            if (L.enum_search(plan, done, cnt, g_recsub_order ? PLO_COST_RECSUB : PLO_COST_ADD_THEN_MUL, &b, &mp, &st) != PLO_OK) {
                std::clog << "# -E skipped: " << L.last_error() << std::endl; L.plan_destroy(plan); return false;
            }
            rb = {b.adds, b.muls}; ri = b.seed; rp_ = mp; rh = true; kms += st.kernel_ms;
        } else {
#pragma omp parallel for schedule(dynamic, 8)
            for (long long k = 0; k < (long long)cnt; ++k) {
                Ops ops; uint64_t pr = 1; size_t rm = 0; (void)schedule_text(f, lM, done + (uint64_t)k, ops, pr, &rm);
                if (g_recsub_order) ops.second = rm;
#pragma omp critical
                { const uint64_t ix = done + (uint64_t)k; if (!rh || better(ops, rb) || (!better(rb, ops) && ix < ri)) { rb = ops; ri = ix; rh = true; } rp_ = std::max(rp_, pr); }
            }
        }
        if (rh && (!have || better(rb, best))) { best = rb; bidx = ri; have = true; }
        maxprod = std::max(maxprod, rp_);
        done = target;
        target = std::min<uint64_t>(std::max<uint64_t>(maxprod, done), budget);      // grow to the largest tree size seen so far
    }
    if (on_gpu) L.plan_destroy(plan);
    if (!have) return false;
    Ops rops; uint64_t pr = 1; size_t rm = 0; std::string t = schedule_text(f, lM, bidx, rops, pr, &rm);
    const Ops prog = rops;                                   // what the printed program costs
    if (g_recsub_order) rops.second = rm;
    if (rops != best) { ++g_failures, std::cerr << "# \033[1;31mERROR: -E replay of schedule " << bidx << " gives " << rops.first << '|' << rops.second << ", search said " << best.first << '|' << best.second << "\033[0m" << std::endl; return false; }
    const bool complete = maxprod <= done;
    if (verbose > 0) {
        if (g_recsub_order) std::clog << "# RecSub accounting: " << best.first << '|' << best.second << " (additions | multiplications before ProgramGen); program " << prog.first << '|' << prog.second << std::endl;
        std::clog << "# Found E: " << prog.first << '|' << prog.second << " instead of " << gops.first << '|' << gops.second << "\t[schedule " << bidx << "] ("
                  << done << (complete ? " schedules: the whole tree" : " schedules of a tree of at least ") ;
        if (!complete) std::clog << maxprod;
        std::clog << (on_gpu ? ", GPU" : ", host");
        if (on_gpu) std::clog << " kernel " << kms << " ms";
        std::clog << ')' << std::endl;
    }
    if (cmp_op_count(prog, gops)) { gops = prog; gtext = t; }                                              // :1270-1274
    else if (complete) std::clog << "# \033[1;36mNo greedy CSE schedule has less additions.\033[0m" << std::endl;       // :1275-1278
    if (!complete) std::clog << "# \033[1;36m-E stopped after " << done << " schedules of a tree of at least " << maxprod << ": the result is the best of those only.\033[0m" << std::endl;
    return true;
}

template <class F>
int run(const F &f, const QMat &MQ, size_t loops, uint64_t seed0, int gpu, bool tryDirect, bool tryKernel, bool tryLU,
        bool tryAB, bool mostCSE, bool allkernels, bool kfi, int verbose, uint32_t q, const char *argv0)
{
    auto t0 = std::chrono::steady_clock::now();
    std::clog << std::string(40, '#') << std::endl;
    const Ops opsinit = naive_ops(QField(), MQ);            // src/optimizer.cpp:77: computed over Q
    auto lM = rebind(MQ, f);
    Ops nbops = opsinit; std::string text;

    // the eliminations of -K, -G, -A and -N run on the host with dense or map-based rows: not attempted on inputs like
    // 32x32x32_15096_L (15096 x 1024), where only the direct method is meant to run (BASELINE config 5 uses -D)
    const double elim_cost = (double)lM.rowdim() * (double)lM.coldim() * (double)std::min(lM.rowdim(), lM.coldim());
    if (elim_cost > 2e9 && (tryAB || tryKernel || tryLU || allkernels)) {
        std::clog << "# -K/-G/-A/-N skipped: host elimination of a " << lM.rowdim() << 'x' << lM.coldim() << " matrix is not attempted (use -D)" << std::endl;
        tryAB = tryKernel = tryLU = allkernels = false;
    }
    // --gpu N, N >= 2: the seed range of -D and of -K in N contiguous shards, the minimum under (cmpOpCount, seed).
    // Default: ONE process, one host thread and one device per shard inside the library (plo_cse_search_multi,
    // plo_kernel_search_multi), the minimum by RCCL MIN all-reduces -- the seed space sharded over the GPUs of the node as
    // north_star names it; this process forks nothing.  --fork-shards (and the host-engine test knob) keep one forked child per
    // device instead: every fork happens here, BEFORE anything in this process touches the HIP runtime (a runtime does not
    // survive fork), and the library's multi-device entries are not used at all then.
    struct { bool done = false; Ops ops; uint64_t seed = 0; } sharded;
    const bool shard_host_engine = getenv("PLO_SHARD_ENGINE") && std::string(getenv("PLO_SHARD_ENGINE")) == "host";   // test knob: every shard on the host engine
    const bool use_forks = g_fork_shards || shard_host_engine;
    bool d_refused = false;                                 // the device refused -D's plan: the restarts run on the host, said aloud
    auto reduce_note = [](const plo_stats_t &st) {
        std::ostringstream os;
        if (st.reduce) os << ", minimum by RCCL MIN all-reduce in " << st.reduce_seconds * 1e3 << " ms"; else os << ", minimum on the host";
        return os.str();
    };
    if constexpr (std::is_same<F, ZpField>::value) if (tryDirect && q != 0 && gpu >= 2 && loops > 0 && !use_forks) {
        HipLib L;
        if (!L.load(argv0) || !L.cse_search_multi) { ++g_failures; std::cerr << "# \033[1;31mERROR: shard failed: libplinopt_hip.so " << (L.cse_search ? "lacks plo_cse_search_multi" : "cannot be loaded") << "\033[0m" << std::endl; return 2; }
        std::vector<uint32_t> rp(1, 0), cc, vv;
        for (auto &r : lM.rows) { for (auto &e : r) { cc.push_back((uint32_t)e.first); vv.push_back((uint32_t)e.second); } rp.push_back((uint32_t)cc.size()); }
        std::vector<int> devs((size_t)gpu); for (int r = 0; r < gpu; ++r) devs[(size_t)r] = shard_device(r);
        plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
        plo_best_t b{}; plo_stats_t st{};
        const int rc = L.cse_search_multi(&A, q, seed0, loops, PLO_COST_SUM_THEN_ADD, gpu, devs.data(), &b, &st);
        if (rc == PLO_E_CAPACITY || rc == PLO_E_UNSUPPORTED) {                          // as one device does: a limit of the kernels for this input
            std::clog << "# -D on the GPUs refused (" << L.last_error() << "): host search" << std::endl;
            d_refused = true;
        } else if (rc != PLO_OK) { ++g_failures; std::cerr << "# \033[1;31mERROR: shard failed: " << L.last_error() << "\033[0m" << std::endl; return 2; }
        else {
            sharded.ops = {b.adds, b.muls}; sharded.seed = b.seed; sharded.done = b.seed != ~0ull;
            if (verbose > 0) std::clog << "# " << gpu << " shards (one GPU and one host thread each, one process): " << st.candidates << " candidates, slowest kernel " << st.kernel_ms << " ms" << reduce_note(st) << std::endl;
        }
    }
    if constexpr (std::is_same<F, ZpField>::value) if (tryDirect && q != 0 && gpu >= 2 && loops > 0 && use_forks) {
        const bool host_engine = shard_host_engine;
        std::vector<uint32_t> rp(1, 0), cc, vv;
        for (auto &r : lM.rows) { for (auto &e : r) { cc.push_back((uint32_t)e.first); vv.push_back((uint32_t)e.second); } rp.push_back((uint32_t)cc.size()); }
        auto shard = [&](int, int device, uint64_t s0, uint64_t cnt) {
            ShardOut o{};
            if (cnt == 0) { o.ok = 1; o.a = o.b = 0xFFFFFFFFu; return o; }
            if (host_engine) {
#ifdef _OPENMP
                omp_set_num_threads(1);                                                // a forked child keeps to its own thread
#endif
                Ops b; uint64_t bs = 0;
                if (host_search(f, lM, s0, (size_t)cnt, b, bs)) { o.ok = 1; o.a = (uint32_t)b.first; o.b = (uint32_t)b.second; o.seed = bs; o.candidates = cnt; }
                return o;
            }
            HipLib L;
            if (!L.load(argv0) || L.init(device) != PLO_OK) { snprintf(o.msg, sizeof o.msg, "device %d: %s", device, L.last_error ? L.last_error() : "library missing"); return o; }
            plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
            plo_best_t b{}; plo_stats_t st{};
            const int rc = L.cse_search(&A, q, s0, cnt, PLO_COST_SUM_THEN_ADD, &b, &st);
            if (rc != PLO_OK) { o.rc = rc; snprintf(o.msg, sizeof o.msg, "device %d: %s", device, L.last_error()); return o; }
            o.ok = 1; o.a = b.adds; o.b = b.muls; o.seed = b.seed; o.candidates = st.candidates; o.kernel_ms = st.kernel_ms;
            L.shutdown();
            return o;
        };
        std::vector<ShardOut> outs;
        if (!forked_shards(gpu, seed0, loops, shard, outs)) {
            bool refused = true;
            for (auto &o : outs) if (!o.ok && o.rc != PLO_E_UNSUPPORTED && o.rc != PLO_E_CAPACITY) refused = false;
            if (!refused) {
                for (auto &o : outs) if (!o.ok) ++g_failures, std::cerr << "# \033[1;31mERROR: shard failed: " << o.msg << "\033[0m" << std::endl;
                return 2;
            }
            std::clog << "# -D on the GPUs refused (" << outs[0].msg << "): host search" << std::endl;
            d_refused = true;
        } else {
            bool have = false; uint64_t total = 0; double kmax = 0;
            for (auto &o : outs) {
                total += o.candidates; kmax = std::max(kmax, o.kernel_ms);
                if (o.a == 0xFFFFFFFFu && o.b == 0xFFFFFFFFu) continue;
                const Ops ops{o.a, o.b};
                if (!have || cmp_op_count(ops, sharded.ops) || (!cmp_op_count(sharded.ops, ops) && o.seed < sharded.seed)) { sharded.ops = ops; sharded.seed = o.seed; have = true; }
            }
            sharded.done = have;
            if (verbose > 0) std::clog << "# " << gpu << " shards" << (host_engine ? " (host engine)" : " (one forked process and one GPU each)") << ": " << total << " candidates, slowest kernel " << kmax << " ms" << std::endl;
        }
    }
    // the same for -K (decompositions on the device, one per restart): N shards of the restart range
    if constexpr (std::is_same<F, ZpField>::value) if (tryKernel && !kfi && q != 0 && gpu >= 2 && loops > 0 && !g_host_decomp && g_kernel_block == 1 && lM.rowdim() <= 128 && lM.coldim() <= 64 && !shard_host_engine) {
        KernelDecomp<F> kd0;
        if (kernel_decomp(f, lM, seed0, kd0)) {                                       // (a zero dimensional kernel is reported by kernel_method)
            std::vector<uint32_t> rp, cc, vv; to_csr(lM, rp, cc, vv);
            if (!use_forks) {
                HipLib L;
                if (!L.load(argv0) || !L.kernel_search_multi) { ++g_failures; std::cerr << "# \033[1;31mERROR: -K shard failed: libplinopt_hip.so " << (L.cse_search ? "lacks plo_kernel_search_multi" : "cannot be loaded") << "\033[0m" << std::endl; return 2; }
                std::vector<int> devs((size_t)gpu); for (int r = 0; r < gpu; ++r) devs[(size_t)r] = shard_device(r);
                plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
                plo_best_t b{}; plo_stats_t st{};
                const int rc = L.kernel_search_multi(&A, q, seed0, loops, 1u, PLO_COST_SUM_THEN_ADD, gpu, devs.data(), &b, &st);
                if (rc == PLO_E_UNSUPPORTED || rc == PLO_E_CAPACITY) {
                    // a plan the device refuses (state beyond LDS, table bounds) is what the single-device path answers with host
                    // decompositions + the batched chain kernel: leave the method to kernel_method then
                    if (verbose > 0) std::clog << "# -K: the device refused the one-wave restart (" << L.last_error() << "): host decompositions + batched chain kernel on one device" << std::endl;
                } else if (rc != PLO_OK) { ++g_failures; std::cerr << "# \033[1;31mERROR: -K shard failed: " << L.last_error() << "\033[0m" << std::endl; return 2; }
                else if (b.seed != ~0ull) {
                    g_kshard.ops = {b.adds, b.muls}; g_kshard.seed = b.seed; g_kshard.ncand = st.candidates; g_kshard.kms = st.kernel_ms; g_kshard.done = true; g_kshard.shards = gpu;
                    g_kshard.how = "one GPU and one host thread each, one process" + reduce_note(st);
                }
            } else {
            auto shard = [&](int, int device, uint64_t s0, uint64_t cnt) {
                ShardOut o{};
                if (cnt == 0) { o.ok = 1; o.a = o.b = 0xFFFFFFFFu; return o; }
                HipLib L;
                if (!L.load(argv0) || !L.kernel_search || L.init(device) != PLO_OK) { snprintf(o.msg, sizeof o.msg, "device %d: %s", device, L.last_error ? L.last_error() : "library missing"); return o; }
                plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
                plo_best_t b{}; plo_stats_t st{};
                const int rc = L.kernel_search(&A, q, s0, cnt, 1u, PLO_COST_SUM_THEN_ADD, nullptr, nullptr, nullptr, &b, &st);
                if (rc != PLO_OK) { o.rc = rc; snprintf(o.msg, sizeof o.msg, "device %d: %s", device, L.last_error()); return o; }
                o.ok = 1; o.a = b.adds; o.b = b.muls; o.seed = b.seed; o.candidates = st.candidates; o.kernel_ms = st.kernel_ms;
                L.shutdown();
                return o;
            };
            std::vector<ShardOut> outs;
            bool kshards_ok = forked_shards(gpu, seed0, loops, shard, outs), refused = false;
            if (!kshards_ok) {
                refused = true;
                for (auto &o : outs) if (!o.ok && o.rc != PLO_E_UNSUPPORTED && o.rc != PLO_E_CAPACITY) refused = false;
                if (!refused) {
                    for (auto &o : outs) if (!o.ok) ++g_failures, std::cerr << "# \033[1;31mERROR: -K shard failed: " << o.msg << "\033[0m" << std::endl;
                    return 2;
                }
                if (verbose > 0) std::clog << "# -K: the device refused the one-wave restart (" << outs[0].msg << "): host decompositions + batched chain kernel on one device" << std::endl;
            }
            bool have = false;
            if (!refused)
            for (auto &o : outs) {
                g_kshard.ncand += o.candidates; g_kshard.kms = std::max(g_kshard.kms, o.kernel_ms);
                if (o.a == 0xFFFFFFFFu && o.b == 0xFFFFFFFFu) continue;
                const Ops ops{o.a, o.b};
                if (!have || cmp_op_count(ops, g_kshard.ops) || (!cmp_op_count(g_kshard.ops, ops) && o.seed < g_kshard.seed)) { g_kshard.ops = ops; g_kshard.seed = o.seed; have = true; }
            }
            g_kshard.done = have; g_kshard.shards = gpu; g_kshard.how = "one forked process and one GPU each";
            }
        }
    }
    if (tryAB) {                                                                      // :1436-1440 (inner dimension = column count)
        // every inner dimension from the column count to the row count - 1 (:1437-1439); a square matrix has the identity factorization only
        const size_t id0 = lM.coldim(), id1 = std::max<size_t>(lM.rowdim(), id0 + 1);
        for (size_t id = id0; id < id1; ++id) {
            try { ab_method(f, lM, seed0, loops, gpu, q, verbose, nbops, text, argv0, id); }
            catch (const std::exception &e) { std::clog << "# -A skipped: " << e.what() << std::endl; }
        }
    }
    if (tryDirect) {
        Ops dops; uint64_t seed = 0; bool have = false;
        bool on_gpu = false;
        if (sharded.done) { on_gpu = true; dops = sharded.ops; seed = sharded.seed; have = loops > 0; }
        if constexpr (std::is_same<F, ZpField>::value) if (q != 0 && gpu > 0 && !sharded.done && !d_refused) {
            on_gpu = true;
            HipLib L;
            if (!L.load(argv0)) return 2;
            if (L.init(0) != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: " << L.last_error() << "\033[0m" << std::endl; return 2; }
            std::vector<uint32_t> rp(1, 0), cc, vv;
            for (auto &r : lM.rows) { for (auto &e : r) { cc.push_back((uint32_t)e.first); vv.push_back((uint32_t)e.second); } rp.push_back((uint32_t)cc.size()); }
            plo_csr_t A{(uint32_t)lM.rowdim(), (uint32_t)lM.coldim(), rp.data(), cc.data(), vv.data()};
            plo_best_t b{}; plo_stats_t st{};
            int rc = L.cse_search(&A, q, seed0, loops, PLO_COST_SUM_THEN_ADD, &b, &st);
            if (rc == PLO_E_CAPACITY || rc == PLO_E_UNSUPPORTED) {
                // a limit of the device kernels for this input (e.g. a column whose non +-1 entries span more than 64 rows in ProgramGen): said
                // aloud, and the same restarts run on the host (as -K and bin/sparsifier do when the device refuses)
                std::clog << "# -D on the GPU refused (" << L.last_error() << "): host search" << std::endl;
                on_gpu = false;
            } else if (rc != PLO_OK) { ++g_failures, std::cerr << "# \033[1;31mERROR: GPU search failed (" << rc << "): " << L.last_error() << "\033[0m" << std::endl; return 2; }
            else {
                dops = {b.adds, b.muls}; seed = b.seed; have = loops > 0;
                if (verbose > 0)
                    std::clog << "# GPU: " << st.candidates << " candidates, kernel " << st.kernel_ms << " ms, "
                              << (st.kernel_ms > 0 ? st.candidates / (st.kernel_ms * 1e-3) : 0.0) << " candidates/s" << std::endl;
            }
            L.shutdown();
        }
        if (!on_gpu) {
            auto ts = std::chrono::steady_clock::now();
            have = host_search(f, lM, seed0, loops, dops, seed);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
            int nthr = 1;
#ifdef _OPENMP
            nthr = omp_get_max_threads();
#endif
            if (verbose > 0) std::clog << "# host search: " << loops << " candidates in " << dt << " s on " << nthr << " threads (index build included)" << std::endl;
        }
        if (have) {
            Ops rops; std::string t = replay_text(f, lM, seed, rops);
            if (rops != dops) { ++g_failures, std::cerr << "# \033[1;31mERROR: replay of seed " << seed << " gives " << rops.first << '|' << rops.second
                                          << ", search said " << dops.first << '|' << dops.second << "\033[0m" << std::endl; return 3; }
            if (verbose > 0) std::clog << "# Found D: " << dops.first << '|' << dops.second << " instead of "
                                       << nbops.first << '|' << nbops.second << "\t[seed " << seed << ']' << std::endl;
            if (cmp_op_count(dops, nbops)) { nbops = dops; text = t; }                 // :1241-1245
        }
    }
    if (tryKernel && !kfi) {                                                          // :1450-1462
        try { kernel_method(f, lM, seed0, loops, gpu, q, verbose, nbops, text, argv0); }
        catch (const std::exception &e) { std::clog << "# -K skipped: " << e.what() << std::endl; }
    }
    if (tryKernel && kfi) {
        // -F: the kernel method with the identity added to the goals (nullspacedecomp :637-685): the inputs become rows that the
        // decomposition may put in its basis, so a dependent output can be written over inputs as well as over other outputs.
        // The program of [M ; I] is searched as any other; its identity goals are then no outputs: the text is renamed
        // (goals 'q'), the real outputs are copied (`o_j := q_j`, :680-681) and the rewriting engine of bin/compacter removes
        // what only the identity goals needed (the reference does this with OpOnVars / RemoveVars, :657-663).
        try {
            auto Mx = lM; const size_t m = lM.rowdim();
            for (size_t i = 0; i < lM.coldim(); ++i) { Mx.rows.emplace_back(); Mx.rows.back().emplace_back(i, f.one()); }
            Ops kops{~(size_t)0 >> 1, 0}; std::string ktext;
            if (kernel_method(f, Mx, seed0, loops, gpu, q, verbose, kops, ktext, argv0) && !ktext.empty()) {
                std::istringstream in(ktext);
                std::vector<compact::Line> Pg = compact::parse(in);
                auto ren = [](std::string &t) { if (t.size() > 1 && t[0] == 'o' && isdigit((unsigned char)t[1])) t[0] = 'q'; };
                for (auto &l : Pg) { ren(l.lhs); for (auto &t : l.rhs) ren(t); }
                for (size_t j = 0; j < m; ++j) Pg.push_back(compact::Line{"o" + std::to_string(j), {"q" + std::to_string(j)}});
                compact::compact_program(Pg, false, compact::letter_outputs('o'));
                std::ostringstream os; compact::print(os, Pg);
                SlpEval<F> ev(f); std::istringstream chk(os.str());
                const Ops cops = ev.run(chk);
                if (!same_matrix(f, ev.matrix('o', lM.rowdim(), lM.coldim()), lM)) throw std::runtime_error("the cleaned program does not compute the matrix");
                if (verbose > 0) std::clog << "# Found K with identity goals (-F): " << cops.first << '|' << cops.second << " after cleaning (" << kops.first << '|' << kops.second
                                           << " with the identity goals) instead of " << nbops.first << '|' << nbops.second << std::endl;
                if (cmp_op_count(cops, nbops)) { nbops = cops; text = os.str(); }
            }
        } catch (const std::exception &e) { std::clog << "# -K -F skipped: " << e.what() << std::endl; }
    }
    if (tryLU) {
        try { lu_method(f, lM, seed0, loops, gpu, q, verbose, nbops, text, argv0); }
        catch (const std::exception &e) { std::clog << "# -G skipped: " << e.what() << std::endl; }
    }
    if (allkernels) {                                                                 // :1463-1465
        try { allkernels_method(f, lM, seed0, loops, gpu, q, verbose, nbops, text, argv0); }
        catch (const std::exception &e) { std::clog << "# -N skipped: " << e.what() << std::endl; }
    }
    if (mostCSE) {                                                                    // :1469-1471
        try { exhaustive_method(f, lM, std::max<uint64_t>(loops, 1ull << 22), gpu, q, verbose, nbops, text, argv0); }
        catch (const std::exception &e) { std::clog << "# -E skipped: " << e.what() << std::endl; }
    }

    if (cmp_op_count(opsinit, nbops) || opsinit == nbops) {                           // :1473-1485
        std::ostringstream os;
        input2temps(os, lM, 'i', 't');
        Replay<F> R(f, lM, 0, os, 'o', 't', 'r');
        nbops = R.direct(); text = os.str();
    }
    std::cout << text << std::flush;                                                  // src/optimizer.cpp:87
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (nbops.first != 0 || nbops.second != 0) {
        std::clog << std::string(40, '#') << std::endl;
        std::clog << "# \033[1;32m" << nbops.first << "\tadditions\tinstead of " << opsinit.first << "\033[0m \t" << secs << "s" << std::endl;
        std::clog << "# \033[1;32m" << nbops.second << "\tmultiplications\tinstead of " << opsinit.second << "\033[0m" << std::endl;
        std::clog << std::string(40, '#') << std::endl;
    }
    if (g_failures) { std::cerr << "# \033[1;31mERROR: " << g_failures << " GPU call(s) or replay check(s) failed: the program above comes from the methods that did run\033[0m" << std::endl; return 2; }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    bool printMaple = false, printPretty = false, mostCSE = false, tryKernel = false, tryLU = false, tryAB = false,
         tryDirect = false, allkernels = false, kfi = false;
    int verbose = 1, gpu = 1; size_t loops = 100; uint64_t q = 0, seed0 = 0; std::string filename, only; bool replay_only = false;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") {
            std::clog << "Usage: " << argv[0] << " [-h|-M|-P|-K|-D|-G|-E|-N|-A|-q #|-O #|--gpu #|--seed #] [stdin|matrixfile.sms]\n"
                      << "  -D/-K/-G: direct/kernel/LU methods (default is all)\n"
                      << "  -F: kernel method with the identity added to its goals (inputs may join the row basis)\n"
                      << "  -q #: search modulo (default is Rationals, on the host)\n"
                      << "  -O #: randomized search with that many loops (default " << loops << " loops)\n"
                      << "  --gpu #: 1 = run the restart loop on the MI355X (default with -q), 0 = host only, N >= 2 = the seeds of -D in N shards, one GPU each\n"
                      << "  --seed #: first candidate seed (default 0)\n"
                      << "  -A: also try the alternative factorizations M = Alt.CoB, every inner dimension from the column count to the row count - 1\n"
                      << "  -E: also walk the exhaustive tree of greedy CSE schedules (bounded by max(-O, 2^22) schedules; every DISTINCT triple of\n"
                      << "      frequency > 1 is a child, once -- the reference tries it once per row holding it); best by additions then multiplications\n"
                      << "      of the printed program, or with --recsub by RecSub's own counts (multiplications before ProgramGen)\n"
                      << "  --only D|K|G|A|E|N: run exactly that method\n"
                      << "  --host-decomp: -K makes the nullspace decompositions on the host (default: on the device when the matrix has at most 64 rows and columns)\n"
                      << "  --kernel-block #: restarts per nullspace decomposition of -K (default 1: one decomposition per restart, as the reference)\n"
                      << "  -M/-P: also print the matrix (Maple / pretty) on the log stream\n";
            exit(-1);
        } else if (a == "-M") printMaple = true;
        else if (a == "-P") printPretty = true;
        else if (a == "-G") tryLU = true;
        else if (a == "-A") tryAB = true;
        else if (a == "-D") tryDirect = true;
        else if (a == "-K") tryKernel = true;
        else if (a == "-F") kfi = true;
        else if (a == "-E") mostCSE = true;
        else if (a == "-N") allkernels = true;
        else if (a == "-q" && i + 1 < argc) q = strtoull(argv[++i], nullptr, 10);
        else if (a == "-v" && i + 1 < argc) verbose = atoi(argv[++i]);
        else if (a == "-O" && i + 1 < argc) loops = strtoull(argv[++i], nullptr, 10);
        else if (a == "--gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed0 = strtoull(argv[++i], nullptr, 10);
        else if (a == "--engine" && i + 1 < argc) { std::string e(argv[++i]); g_engine = e == "literal" ? 1 : e == "fast" ? 2 : 0; }
        else if (a == "--replay") replay_only = true;    // print the program of candidate --seed, no search
        else if (a == "--host-decomp") g_host_decomp = true;
        else if (a == "--fork-shards") g_fork_shards = true;
        else if (a == "--kernel-block" && i + 1 < argc) g_kernel_block = std::max<uint64_t>(1, strtoull(argv[++i], nullptr, 10));
        else if (a == "--recsub") g_recsub_order = true;
        else if (a == "--only" && i + 1 < argc) { only = argv[++i]; }   // run exactly one method (D, G or A): for tests and timing
        else filename = a;
    }
#ifdef _OPENMP
    if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(std::min(omp_get_max_threads(), 64));   // cgroup-limited boxes report all host cores
#endif
    if (!only.empty()) { tryDirect = only == "D"; tryLU = only == "G"; tryAB = only == "A"; tryKernel = only == "K"; mostCSE = only == "E"; allkernels = only == "N"; }
    else if (!tryKernel && !tryDirect && !tryLU) tryLU = tryDirect = tryKernel = true;      // src/optimizer.cpp:208-211
    try {
        QMat MQ;
        if (filename.empty()) MQ = read_sms(std::cin);
        else { std::ifstream in(filename); if (!in) return -1; MQ = read_sms(in); }
        if (printPretty || printMaple) {                                             // src/optimizer.cpp:65-73 (LinBox's writers are not in the tree: plain dense forms)
            auto dense = [&](const char *open, const char *rowopen, const char *sep, const char *rowclose, const char *rowsep, const char *close) {
                std::clog << open;
                for (size_t i = 0; i < MQ.rowdim(); ++i) {
                    std::clog << (i ? rowsep : "") << rowopen; size_t k = 0;
                    for (size_t j = 0; j < MQ.coldim(); ++j) {
                        if (j) std::clog << sep;
                        if (k < MQ.rows[i].size() && MQ.rows[i][k].first == j) { std::clog << MQ.rows[i][k].second; ++k; } else std::clog << 0;
                    }
                    std::clog << rowclose;
                }
                std::clog << close << ';' << std::endl << std::string(40, '#') << std::endl;
            };
            if (printPretty) dense("", "[", " ", "]", "\n", "");
            if (printMaple) dense("M:=Matrix([", "[", ",", "]", ",", "])");
        }
        if (replay_only) {
            Ops ops;
            if (q >= (1ull << 31)) { Zp64Field f(q); std::cout << replay_text(f, rebind(MQ, f), seed0, ops); }
            else if (q) { ZpField f((uint32_t)q); std::cout << replay_text(f, rebind(MQ, f), seed0, ops); }
            else { QField f; std::cout << replay_text(f, rebind(MQ, f), seed0, ops); }
            std::clog << "# " << ops.first << "\tadditions\n# " << ops.second << "\tmultiplications" << std::endl;
            return 0;
        }
        if (q != 0) {
            if (q < 3 || q >= (1ull << 62)) { std::cerr << "# ERROR: modulus must be an odd prime below 2^62 in this build" << std::endl; return -1; }
            if (q >= (1ull << 31)) {                       // the kernels keep 31-bit residues: larger moduli run the host loops (the reference's field is Modular<Integer>, src/optimizer.cpp:131)
                std::clog << "# modulus above 2^31: host loops (the GPU kernels hold 31-bit residues)" << std::endl;
                return run(Zp64Field(q), MQ, loops, seed0, 0, tryDirect, tryKernel, tryLU, tryAB, mostCSE, allkernels, kfi, verbose, 0, argv[0]);
            }
            return run(ZpField((uint32_t)q), MQ, loops, seed0, gpu, tryDirect, tryKernel, tryLU, tryAB, mostCSE, allkernels, kfi, verbose, (uint32_t)q, argv[0]);
        }
        return run(QField(), MQ, loops, seed0, 0, tryDirect, tryKernel, tryLU, tryAB, mostCSE, allkernels, kfi, verbose, 0, argv[0]);
    } catch (const std::exception &e) {
        ++g_failures, std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return -1;
    }
}
