// ==========================================================================
// bin/trilplacer -- in-place trilinear programs (reference src/trilplacer.cpp:60-190):
//   trilplacer L.sms R.sms P.sms [-e] [-m] [-O #] [--seed s] [--gpu 0|1|N]
// stdout = the program (a_i, b_j restored, c_k += the bilinear map), clog = '#' statistics
// ("ADD / SCA / AXPY" as the reference :170-180).  The restart loop of
// SearchTriLinearAlgorithm (include/plinopt_inplace.inl:837-924) runs on the GPU through
// plo_tril_search[_multi] of libplinopt_hip.so when the three matrices have no empty row and no
// row of more than 64 entries (rational coefficients as residues modulo a 31-bit prime); the winner
// is replayed on the host over Q to print it.  Other inputs, or --gpu 0, use the host loop (OpenMP).
// -e: the expanded variant (TransposedDoubleAlgorithm on DoubleExpand(P^T)), on the device too.
// ==========================================================================
#include "plo_inplace.hpp"
#include "../../../include/plinopt_hip.h"
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>
#include <chrono>
#include <fstream>
#include <omp.h>
#include <tuple>

using namespace plo;

namespace {
struct HipTril {
    void *h = nullptr;
    decltype(&plo_init) init = nullptr; decltype(&plo_last_error) last_error = nullptr;
    decltype(&plo_tril_plan_create_q) create = nullptr; decltype(&plo_tril_plan_destroy) destroy = nullptr; decltype(&plo_tril_search) search = nullptr;
    decltype(&plo_tril_search_multi) search_multi = nullptr;
    bool load() {
        std::vector<std::string> cand;
        for (const char *v : {"PLO_HIP_LIB", "PLINOPT_HIP_LIB"}) if (const char *e = getenv(v)) cand.emplace_back(e);   // (one name for the tools and plinopt_amd/capi.py; the older one still works)
        char buf[4096]; ssize_t k = readlink("/proc/self/exe", buf, sizeof buf - 1);
        if (k > 0) { buf[k] = 0; std::string d = dirname(buf); cand.push_back(d + "/../plinopt_amd/libplinopt_hip.so"); cand.push_back(d + "/libplinopt_hip.so"); }
        cand.emplace_back("libplinopt_hip.so");
        for (auto &c : cand) { h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { std::cerr << "# \033[1;31mERROR: cannot load libplinopt_hip.so: " << dlerror() << "\033[0m\n"; return false; }
        init = (decltype(init))dlsym(h, "plo_init"); last_error = (decltype(last_error))dlsym(h, "plo_last_error");
        create = (decltype(create))dlsym(h, "plo_tril_plan_create_q"); destroy = (decltype(destroy))dlsym(h, "plo_tril_plan_destroy");
        search = (decltype(search))dlsym(h, "plo_tril_search"); search_multi = (decltype(search_multi))dlsym(h, "plo_tril_search_multi");
        return init && last_error && create && destroy && search;
    }
};

bool better(const Tricount &l, const Tricount &r) { return l[0] < r[0] || (l[0] == r[0] && l[1] < r[1]); }   // :893-897

// rational CSR for plo_tril_plan_create_q (round 3: the device programs carry the coefficients modulo a 31-bit prime)
struct ICsr { std::vector<uint32_t> rp{0}, col; std::vector<int64_t> num, den; bool unit = true, full = true; };
ICsr icsr(const QMat &M) {
    ICsr c;
    for (const auto &row : M.rows) {
        if (row.empty() || row.size() > 64) c.full = false;
        for (const auto &e : row) {
            c.col.push_back((uint32_t)e.first);
            if (!(e.second.d == 1 && (e.second.n == 1 || e.second.n == -1))) c.unit = false;
            // (Rat is 128 bits wide; the C-ABI takes 64-bit numerators and denominators: a wider coefficient keeps the matrix on the host)
            if (e.second.n > (__int128)INT64_MAX || e.second.n < -(__int128)INT64_MAX || e.second.d > (__int128)INT64_MAX || e.second.d < -(__int128)INT64_MAX) c.full = false;
            c.num.push_back((int64_t)e.second.n); c.den.push_back((int64_t)e.second.d);
        }
        c.rp.push_back((uint32_t)c.col.size());
    }
    return c;
}
} // namespace

int main(int argc, char **argv) {
#ifdef _OPENMP
    if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(std::min(omp_get_max_threads(), 64));   // cgroup-limited boxes report all host cores
#endif
    size_t loops = 30; uint64_t seed0 = 0; int gpu = 1; bool expanded = false, fork_shards = false; std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") { std::clog << "Usage: " << argv[0] << " L.sms R.sms P.sms [-e] [-O #] [--seed s] [--gpu 0|1|N: N >= 2 shards the restarts over N GPUs] [--fork-shards]\n"; return 0; }
        else if (a == "-O" && i + 1 < argc) loops = (size_t)atoll(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed0 = strtoull(argv[++i], nullptr, 10);
        else if (a == "--gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
        else if (a == "-m") { }        // reference: selects the Maple check of an INPLACE_CHECKER build (src/trilplacer.cpp:79,155); nothing to do here
        else if (a == "--fork-shards") fork_shards = true;   // --gpu N with one forked child per device instead of one host thread per device inside the library
        else if (a == "-e") expanded = true;           // src/trilplacer.cpp:80,114-137: double-size products, two entries of c per AXPY
        else files.push_back(a);
    }
    if (files.size() != 3) { std::cerr << "# \033[1;31mERROR: three matrices needed (L R P)\033[0m\n"; return -1; }
    try {
        QMat M[3];
        for (int k = 0; k < 3; ++k) { std::ifstream in(files[(size_t)k]); if (!in) { std::cerr << "# \033[1;31mERROR: cannot read " << files[(size_t)k] << "\033[0m\n"; return -1; } M[k] = read_sms(in); }
        const QMat &A = M[0], &B = M[1]; QMat T = transpose(M[2]);
        if (A.rowdim() != B.rowdim() || A.rowdim() != T.rowdim()) { std::cerr << "Incorrect dimensions\n"; return 1; }
        const auto t0 = std::chrono::steady_clock::now();
        TrilCandidate basec = tril_candidate(A, B, T, ~0ull, 0, expanded);
        Tricount best = basec.ops[0]; uint64_t bseed = ~0ull; int bvar = 0;
        std::clog << "# Oriented number of operations: " << best[0] << '|' << best[1] << '|' << best[2] << std::endl;
        bool on_gpu = false; double kms = 0;
        if (loops > 0) {
            ICsr ca = icsr(A), cb = icsr(B), ct = icsr(T);
            // device path: no empty row, rows of at most 64 entries, coefficients that fit 64 bits (-e included: round 4)
            const bool device_ok = ca.full && cb.full && ct.full;
            using Key = std::tuple<size_t, size_t, uint64_t, int>;      // (ADD, SCA, seed, variant): the order of :893-897 made total
            // restarts s0 .. s0+cnt-1 on one device (plo_tril_search); throws on failure
            auto gpu_search = [&](int device, uint64_t s0, uint64_t cnt, plo_tril_best_t &r, plo_stats_t &st) {
                HipTril L;
                if (!L.load()) throw std::runtime_error("cannot load libplinopt_hip.so");        // no silent fallback: --gpu 0 selects the host loop
                if (L.init(device) != PLO_OK) throw std::runtime_error(L.last_error());
                plo_qcsr_t a{(uint32_t)A.rowdim(), (uint32_t)A.coldim(), ca.rp.data(), ca.col.data(), ca.num.data(), ca.den.data()};
                plo_qcsr_t b{(uint32_t)B.rowdim(), (uint32_t)B.coldim(), cb.rp.data(), cb.col.data(), cb.num.data(), cb.den.data()};
                plo_qcsr_t t{(uint32_t)T.rowdim(), (uint32_t)T.coldim(), ct.rp.data(), ct.col.data(), ct.num.data(), ct.den.data()};
                plo_tril_plan_t *plan = nullptr;
                if (L.create(&a, &b, &t, expanded ? 1 : 0, &plan) != PLO_OK) throw std::runtime_error(L.last_error());
                const int rc = L.search(plan, s0, cnt, &r, &st);
                const std::string msg = rc != PLO_OK ? L.last_error() : "";
                L.destroy(plan);
                if (rc != PLO_OK) throw std::runtime_error(msg);
            };
            // the same restarts on the host: best of the loop under Key
            auto host_loop = [&](uint64_t s0, uint64_t cnt) {
                Key lb{~(size_t)0, ~(size_t)0, 0, 0};
                #pragma omp parallel
                {
                    Key tb = lb;
                    #pragma omp for schedule(dynamic, 16)
                    for (long long k = 0; k < (long long)cnt; ++k) {
                        TrilCandidate c = tril_candidate(A, B, T, s0 + (uint64_t)k, -1, expanded);
                        for (int v = 0; v < 2; ++v) tb = std::min(tb, Key{c.ops[v][0], c.ops[v][1], s0 + (uint64_t)k, v});
                    }
                    #pragma omp critical
                    lb = std::min(lb, tb);
                }
                return lb;
            };
            const bool host_engine = getenv("PLO_SHARD_ENGINE") && std::string(getenv("PLO_SHARD_ENGINE")) == "host";   // test knob: every shard on the host loop
            try {
                if (gpu >= 2 && device_ok && !fork_shards && !host_engine) {
                    // --gpu N (BASELINE configs[3]): N contiguous seed shards over N devices from THIS process -- one host thread, one
                    // device and one plan per shard inside the library, the minimum under Key by RCCL MIN all-reduces (plo_tril_search_multi)
                    HipTril L;
                    if (!L.load() || !L.search_multi) throw std::runtime_error("libplinopt_hip.so cannot be loaded or lacks plo_tril_search_multi");
                    plo_qcsr_t a{(uint32_t)A.rowdim(), (uint32_t)A.coldim(), ca.rp.data(), ca.col.data(), ca.num.data(), ca.den.data()};
                    plo_qcsr_t b{(uint32_t)B.rowdim(), (uint32_t)B.coldim(), cb.rp.data(), cb.col.data(), cb.num.data(), cb.den.data()};
                    plo_qcsr_t t{(uint32_t)T.rowdim(), (uint32_t)T.coldim(), ct.rp.data(), ct.col.data(), ct.num.data(), ct.den.data()};
                    std::vector<int> devs((size_t)gpu); for (int r = 0; r < gpu; ++r) devs[(size_t)r] = shard_device(r);
                    plo_tril_best_t r{}; plo_stats_t st{};
                    if (L.search_multi(&a, &b, &t, expanded ? 1 : 0, seed0, loops, gpu, devs.data(), &r, &st) != PLO_OK) throw std::runtime_error(L.last_error());
                    on_gpu = true; kms = st.kernel_ms;
                    std::clog << "# " << gpu << " shards (one GPU and one host thread each, one process)";
                    if (st.reduce) std::clog << ", minimum by RCCL MIN all-reduce in " << st.reduce_seconds * 1e3 << " ms"; else std::clog << ", minimum on the host";
                    std::clog << std::endl;
                    const Tricount g{r.add, r.sca, r.mul};
                    if (better(g, best)) { best = g; bseed = r.seed; bvar = (int)r.variant; }
                } else if (gpu >= 2 && (device_ok || host_engine)) {
                    // --gpu N --fork-shards: one forked child and one device per shard (every fork before this process touches the HIP
                    // runtime); minimum under Key in the parent
                    auto shard = [&](int, int device, uint64_t s0, uint64_t cnt) {
                        ShardOut o{};
                        if (cnt == 0) { o.ok = 1; o.a = o.b = 0xFFFFFFFFu; return o; }
                        if (host_engine) {
#ifdef _OPENMP
                            omp_set_num_threads(1);
#endif
                            const Key k = host_loop(s0, cnt);
                            o.ok = 1; o.a = (uint32_t)std::get<0>(k); o.b = (uint32_t)std::get<1>(k); o.c = (uint32_t)A.rowdim(); o.seed = std::get<2>(k); o.variant = (uint64_t)std::get<3>(k); o.candidates = cnt;
                            return o;
                        }
                        plo_tril_best_t r{}; plo_stats_t st{};
                        gpu_search(device, s0, cnt, r, st);
                        o.ok = 1; o.a = r.add; o.b = r.sca; o.c = r.mul; o.seed = r.seed; o.variant = r.variant; o.candidates = cnt; o.kernel_ms = st.kernel_ms;
                        return o;
                    };
                    std::vector<ShardOut> outs;
                    if (!forked_shards(gpu, seed0, loops, shard, outs)) { for (auto &o : outs) if (!o.ok) std::cerr << "# \033[1;31mERROR: shard failed: " << o.msg << "\033[0m\n"; return 2; }
                    Key lb{~(size_t)0, ~(size_t)0, 0, 0}; uint32_t mul = (uint32_t)A.rowdim();
                    for (auto &o : outs) { if (o.a == 0xFFFFFFFFu && o.b == 0xFFFFFFFFu) continue; const Key k{o.a, o.b, o.seed, (int)o.variant}; if (k < lb) { lb = k; mul = o.c; } kms = std::max(kms, o.kernel_ms); }
                    on_gpu = !host_engine;
                    std::clog << "# " << gpu << " shards" << (host_engine ? " (host engine)" : " (one forked process and one GPU each)") << std::endl;
                    const Tricount g{std::get<0>(lb), std::get<1>(lb), mul};
                    if (better(g, best)) { best = g; bseed = std::get<2>(lb); bvar = std::get<3>(lb); }
                } else if (gpu && device_ok) {
                    plo_tril_best_t r{}; plo_stats_t st{};
                    gpu_search(0, seed0, loops, r, st);
                    on_gpu = true; kms = st.kernel_ms;
                    const Tricount g{r.add, r.sca, r.mul};
                    if (better(g, best)) { best = g; bseed = r.seed; bvar = (int)r.variant; }
                } else {
                    if (gpu) std::clog << "# an empty row, a row of more than 64 entries or a coefficient wider than 64 bits: host search" << std::endl;
                    // best of the loop under (ADD, SCA, seed, variant), then strictly better than the unpermuted program
                    const Key lb = host_loop(seed0, loops);
                    const Tricount g{std::get<0>(lb), std::get<1>(lb), A.rowdim()};
                    if (better(g, best)) { best = g; bseed = std::get<2>(lb); bvar = std::get<3>(lb); }
                }
            } catch (const std::exception &e) { std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m\n"; return 2; }
        }
        // replay of the winner for the text
        std::string text;
        if (bseed == ~0ull) text = basec.text[0];
        else {
            TrilCandidate w = tril_candidate(A, B, T, bseed, bvar, expanded);
            if (w.ops[bvar] != best) { std::cerr << "# \033[1;31mERROR: replay of seed " << bseed << " gives " << w.ops[bvar][0] << '|' << w.ops[bvar][1] << ", search said " << best[0] << '|' << best[1] << "\033[0m\n"; return 3; }
            text = w.text[bvar];
            std::clog << "# Found " << (bvar ? "unoriented" : "oriented") << " [seed " << bseed << "], operations: " << best[0] << '|' << best[1] << '|' << best[2] << std::endl;
        }
        std::cout << text << std::flush;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::clog << std::string(40, '#') << std::endl;
        std::clog << "# \033[1;32m" << best[0] << "\tADD\033[0m\n# \033[1;32m" << best[1] << "\tSCA\033[0m\n# \033[1;32m" << best[2] << (expanded ? "\tAXPY (double size)\033[0m\n" : "\tAXPY\033[0m\n");
        std::clog << std::string(40, '#') << std::endl;
        std::clog << "# " << loops << " restarts on " << (on_gpu ? "GPU" : "host") << " in " << dt << " s";
        if (on_gpu) std::clog << " (kernel " << kms << " ms)";
        std::clog << std::endl;
    } catch (const std::exception &e) { std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m\n"; return 4; }
    return 0;
}
