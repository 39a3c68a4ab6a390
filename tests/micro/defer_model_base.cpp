// scratch: statistics of one config-5 candidate at the kernel's bookkeeping level
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <string>
#include <fstream>
#include <sstream>
#include <cstring>
using namespace std;
static const uint32_t P = 131071;
static uint32_t mulm(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); }
static uint32_t invm(uint32_t a) { int64_t t = 0, nt = 1, r = P, nr = a; while (nr) { int64_t q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; } if (t < 0) t += P; return (uint32_t)t; }
struct Ent { uint32_t col, val, inv; };
struct Tab {
    vector<uint64_t> k; vector<uint32_t> c; uint32_t bits; uint64_t mask;
    Tab(uint32_t b) : k(1ull << b, ~0ull), c(1ull << b, 0), bits(b), mask((1ull << b) - 1) {}
    uint64_t slot(uint64_t key) const { uint64_t s = (key * 0x9E3779B97F4A7C15ull) >> (64 - bits); while (k[s] != ~0ull && k[s] != key) s = (s + 1) & mask; return s; }
    uint32_t get(uint64_t key) const { uint64_t s = slot(key); return k[s] == key ? c[s] : 0; }
    uint32_t &ref(uint64_t key) { uint64_t s = slot(key); if (k[s] != key) { k[s] = key; c[s] = 0; } return c[s]; }
};
#define KEY(a, b, r) (((uint64_t)(a) << 34) | ((uint64_t)(b) << 17) | (uint64_t)(r))
int main(int argc, char **argv) {
    uint64_t seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    ifstream in("/tmp/sim/l32.sms"); string line; getline(in, line);
    uint32_t m, n; { istringstream ls(line); ls >> m >> n; }
    vector<vector<Ent>> rows(m);
    { long i, j, v; while (in >> i >> j >> v) { if (i == 0) break; rows[i - 1].push_back(Ent{(uint32_t)j - 1, (uint32_t)v, invm((uint32_t)v)}); } }
    for (auto &r : rows) sort(r.begin(), r.end(), [](const Ent &a, const Ent &b) { return a.col < b.col; });
    vector<vector<uint32_t>> colrows(n); vector<uint32_t> ucount(n, 0);
    Tab T(24);
    for (uint32_t i = 0; i < m; ++i) {
        auto &r = rows[i];
        for (size_t x = 0; x < r.size(); ++x) { colrows[r[x].col].push_back(i); if (r[x].val == 1 || r[x].val == P - 1) ++ucount[r[x].col];
            for (size_t y = x + 1; y < r.size(); ++y) ++T.ref(KEY(r[x].col, r[y].col, mulm(r[y].val, r[x].inv))); }
    }
    uint32_t maxf = 0; for (auto c : T.c) maxf = max(maxf, c);
    vector<uint64_t> hist(maxf + 2, 0); for (size_t s = 0; s < T.k.size(); ++s) if (T.k[s] != ~0ull) ++hist[T.c[s]];
    { uint64_t acc = 0; printf("initial levels (f: count, cumulative>=f):\n"); for (uint32_t f = maxf; f >= 2; --f) if (hist[f]) { acc += hist[f]; if (f > 40 || f % 4 == 0 || f < 12) printf("  %u: %lu %lu\n", f, hist[f], acc); } }
    // rng
    uint64_t x = seed + 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
    uint32_t rng = 1u + (uint32_t)(x % 2147483646ull);
    uint32_t M = maxf; size_t ncols = n;
    set<uint64_t> level; bool need = true;
    // stats by class
    struct Cls { uint64_t steps = 0, cand = 0, aff = 0, sumL = 0, agg = 0, dc = 0, ret_present = 0, ret_absent = 0, ins2 = 0, ins1 = 0, candboth = 0; } cls[4];
    const int NH = 5; const uint64_t Hcap[NH] = {2048, 8192, 32768, 131072, 1u << 20};
    uint32_t theta[NH]; uint64_t rebuilds[NH] = {0}, hotret[NH] = {0}, coldret[NH] = {0}, hotins[NH] = {0}, coldins[NH] = {0}, hotkeys_at_rebuild[NH] = {0};
    for (int h = 0; h < NH; ++h) theta[h] = maxf + 1;
    uint64_t steps = 0; vector<uint32_t> Mtrace;
    unordered_map<uint64_t, uint32_t> agg; unordered_set<uint32_t> dcs;
    uint64_t livekeys2 = 0;
    for (;;) {
        while (M >= 2 && hist[M] == 0) { --M; need = true; }
        if (M <= 1) break;
        if (need) { level.clear(); for (size_t s = 0; s < T.k.size(); ++s) if (T.k[s] != ~0ull && T.c[s] == M) level.insert(T.k[s]); need = false; }
        for (int h = 0; h < NH; ++h) if (M < theta[h]) {
            uint64_t acc = 0; uint32_t th = M; for (uint32_t f = M; f >= 2; --f) { if (acc + hist[f] > Hcap[h]) break; acc += hist[f]; th = f; }
            theta[h] = th; ++rebuilds[h]; hotkeys_at_rebuild[h] += acc;
        }
        uint64_t Tn = hist[M]; if (Tn != level.size()) { printf("level mismatch %lu %zu at M=%u\n", Tn, level.size(), M); return 1; }
        uint64_t k = 0; if (Tn > 1) { uint64_t y = 950706376ull * (uint64_t)rng; rng = (uint32_t)(y % 2147483647ull); k = rng % Tn; }
        auto it = level.begin(); advance(it, k);
        const uint64_t key = *it; const uint32_t a = key >> 34, b = (key >> 17) & 0x1FFFF, r = key & 0x1FFFF;
        const uint32_t lm = ncols; colrows.emplace_back(); ucount.push_back(0);
        const bool swap = ucount[a] < ucount[b]; const uint32_t l0 = swap ? b : a;
        const auto &cand = colrows[a].size() <= colrows[b].size() ? colrows[a] : colrows[b];
        Cls &C = cls[M >= 256 ? 0 : M >= 64 ? 1 : M >= 16 ? 2 : 3];
        ++C.steps; C.cand += cand.size();
        vector<uint32_t> aff;
        for (uint32_t i : cand) {
            auto &row = rows[i]; int pa = -1, pb = -1;
            for (size_t z = 0; z < row.size(); ++z) { if (row[z].col == a) pa = z; if (row[z].col == b) pb = z; }
            if (pa < 0 || pb < 0) continue; ++C.candboth;
            if (row[pb].val != mulm(r, row[pa].val)) continue;
            aff.push_back(i);
        }
        sort(aff.begin(), aff.end()); aff.erase(unique(aff.begin(), aff.end()), aff.end());
        if (aff.size() != M) { printf("aff mismatch\n"); return 1; }
        C.aff += aff.size();
        agg.clear(); dcs.clear();
        auto dec = [&](uint64_t kk, uint32_t d) {
            uint64_t s = T.slot(kk); if (T.k[s] != kk || T.c[s] < d) { printf("dec fail\n"); exit(1); }
            uint32_t o = T.c[s];
            for (int h = 0; h < NH; ++h) { if (o >= theta[h]) ++hotret[h]; else ++coldret[h]; }
            if (o >= 2) ++C.ret_present; else ++C.ret_absent;
            --hist[o]; if (o == M) level.erase(kk); T.c[s] = o - d; if (o - d) ++hist[o - d];
        };
        for (uint32_t i : aff) {
            auto &row = rows[i]; int pa = -1, pb = -1;
            for (size_t z = 0; z < row.size(); ++z) { if (row[z].col == a) pa = z; if (row[z].col == b) pb = z; }
            const Ent ea = row[pa], eb = row[pb]; C.sumL += row.size();
            for (size_t z = 0; z < row.size(); ++z) { if ((int)z == pa || (int)z == pb) continue; const Ent &e = row[z];
                const uint32_t xx = e.col < a ? mulm(ea.val, e.inv) : mulm(e.val, ea.inv);
                ++agg[((uint64_t)e.col << 17) | xx]; dcs.insert(e.col); }
            if (ea.val == 1 || ea.val == P - 1) --ucount[a]; if (eb.val == 1 || eb.val == P - 1) --ucount[b];
            const Ent co = l0 == a ? ea : eb;
            row.erase(row.begin() + pb); row.erase(row.begin() + pa);
            row.push_back(Ent{lm, co.val, co.inv}); if (co.val == 1 || co.val == P - 1) ++ucount[lm];
        }
        colrows[lm] = aff;
        C.agg += agg.size(); C.dc += dcs.size();
        dec(key, M);
        const uint32_t invr = invm(r);
        for (auto &kv : agg) {
            const uint32_t c = kv.first >> 17, xx = kv.first & 0x1FFFF, d = kv.second;
            const uint32_t y = c > a ? invm(xx) : xx;   // v_a / v_c
            const uint32_t ry = mulm(r, y), x2 = c < b ? ry : mulm(xx, invr);
            dec(c < a ? KEY(c, a, xx) : KEY(a, c, xx), d); dec(c < b ? KEY(c, b, x2) : KEY(b, c, x2), d);
        }
        for (auto &kv : agg) {
            const uint32_t c = kv.first >> 17, xx = kv.first & 0x1FFFF, d = kv.second;
            const uint32_t y = c > a ? invm(xx) : xx;
            const uint64_t kk = KEY(c, lm, l0 == a ? y : mulm(r, y));
            T.ref(kk) += d; ++hist[d]; if (d == M) level.insert(kk);
            if (d >= 2) ++C.ins2; else ++C.ins1;
            for (int h = 0; h < NH; ++h) { if (d >= theta[h]) ++hotins[h]; else if (d >= 2) ++coldins[h]; }
        }
        ncols = lm + 1; ++steps;
        if (steps % 500 == 0) { uint64_t live = 0; for (uint32_t f = 2; f <= maxf; ++f) live += hist[f]; fprintf(stderr, "step %lu M %u T %lu live(>=2) %lu\n", steps, M, (unsigned long)hist[M], live); }
    }
    printf("steps %lu cols %zu\n", steps, ncols);
    const char *nm[4] = {">=256", "64..255", "16..63", "<16"};
    for (int q = 0; q < 4; ++q) { Cls &C = cls[q];
        printf("class %-8s steps %6lu cand %9lu both %9lu aff %8lu sumL %10lu agg %9lu distinct_c %9lu ret present %9lu absent %9lu ins>=2 %8lu ins1 %8lu\n", nm[q], C.steps, C.cand, C.candboth, C.aff, C.sumL, C.agg, C.dc, C.ret_present, C.ret_absent, C.ins2, C.ins1); }
    for (int h = 0; h < NH; ++h) printf("H %8lu: rebuilds %lu (hot keys listed %lu) hot retirements %lu cold %lu hot inserts %lu cold inserts(>=2) %lu\n", Hcap[h], rebuilds[h], hotkeys_at_rebuild[h], hotret[h], coldret[h], hotins[h], coldins[h]);
    return 0;
}
