// ===========================================================================
// plo_fast.hpp -- scalable, exact host engine for ONE candidate of Optimizer()
// over Z_p (reference include/plinopt_optimize.inl:616-631), for matrices far
// beyond what the literal std::map replay (plo::Replay) can walk: the headline
// input 32x32x32_15096_L has 1.26 M non-zeros, 1.25e8 pair instances and 3.15 M
// distinct pair triples; the reference rescans its whole PairMap at every CSE
// step (:244-253) and re-transposes the matrix (:70), i.e. ~1e11 node visits.
//
// Same decisions, same text, different bookkeeping:
//  * pair index without a map: the triples of the INPUT matrix are a sorted,
//    immutable array shared by every candidate (SharedIndex); a candidate only
//    owns their counts.  Every triple created later contains the newest column
//    as its second member, so "sorted by (col_a, col_b, ratio)" is, per first
//    column, exactly creation order: per-column append-only lists.  Lookups are
//    binary searches; no hashing.
//  * max frequency M is non-increasing over a candidate; per-column counters of
//    "triples at frequency M" give the k-th tie in std::map order by one prefix
//    walk over the columns and one segment scan (OneSub :244-265).
//  * +-1 counts per column are maintained incrementally (RemOneCSE :70-77).
//  * Triangle (:427-507) finds its first (iter,next) couple through a value
//    index instead of the cubic scan, then follows the literal code, including
//    the never-reset `found` flag.
// Product code (host replay / `bin/optimizer --gpu 0`); independent of oracle/.
// ===========================================================================
#ifndef PLO_FAST_HPP
#define PLO_FAST_HPP

#include "plo_host.hpp"

#include <unordered_map>

namespace plo {

struct FEntry { uint32_t col, val, inv; };

// Immutable per-matrix data: rows with inverses, transpose lists and the sorted
// distinct pair triples of the input with their initial frequencies.
struct SharedIndex {
    ZpField f;
    uint32_t m = 0, n = 0;
    std::vector<std::vector<FEntry>> rows;
    std::vector<std::vector<uint32_t>> colrows;       // rows holding each column
    std::vector<uint32_t> ucount;                      // +-1 entries per column
    std::vector<uint64_t> keys;                        // per first column a: sorted (b<<32 | r)
    std::vector<uint32_t> cnt;                         // initial frequency of each key
    std::vector<size_t> colptr;                        // n+1 offsets into keys/cnt
    uint64_t pairs0 = 0;

    SharedIndex(const ZpField &ff, const SparseMat<uint32_t> &A) : f(ff) {
        m = (uint32_t)A.rowdim(); n = (uint32_t)A.coldim();
        rows.resize(m); colrows.resize(n); ucount.assign(n, 0);
        for (uint32_t i = 0; i < m; ++i)
            for (auto &e : A.rows[i]) {
                rows[i].push_back(FEntry{(uint32_t)e.first, e.second, f.inv(e.second)});
                colrows[e.first].push_back(i);
                if (absOne(f, e.second)) ++ucount[e.first];
            }
        colptr.assign(n + 1, 0);
        std::vector<uint64_t> tmp;
        for (uint32_t a = 0; a < n; ++a) {             // all pairs whose first column is a (listpairs :30-41)
            tmp.clear();
            for (uint32_t i : colrows[a]) {
                const auto &row = rows[i];
                size_t x = 0; while (row[x].col != a) ++x;
                for (size_t y = x + 1; y < row.size(); ++y)
                    tmp.push_back(((uint64_t)row[y].col << 32) | f.mul(row[y].val, row[x].inv));
            }
            pairs0 += tmp.size();
            std::sort(tmp.begin(), tmp.end());
            for (size_t k = 0; k < tmp.size();) {
                size_t j = k; while (j < tmp.size() && tmp[j] == tmp[k]) ++j;
                keys.push_back(tmp[k]); cnt.push_back((uint32_t)(j - k)); k = j;
            }
            colptr[a + 1] = keys.size();
        }
    }
};

class FastCand {
    struct DEnt { uint32_t b, r, cnt; };
    struct Ref { uint32_t a; uint32_t dyn; size_t idx; };
    const SharedIndex &S; const ZpField &f; const uint32_t p;
    CandRng rng; std::ostream *out; char ouv, tev, rav;
    std::vector<std::vector<FEntry>> rows;
    std::vector<std::vector<uint32_t>> colrows;
    std::vector<uint32_t> ucount, cnt0;
    std::vector<std::vector<DEnt>> dyn;
    std::vector<uint64_t> hist;                       // hist[f] = triples with frequency f
    std::vector<uint32_t> cntM;                       // per first column: triples at frequency M
    std::vector<Ref> active;                          // triples that had frequency >= 2 when listed
    uint32_t M = 0; size_t ncols;
    struct MKey { uint32_t col, val; bool operator==(const MKey &o) const { return col == o.col && val == o.val; } };
    struct MHash { size_t operator()(const MKey &k) const { return ((uint64_t)k.col << 32 | k.val) * 0x9E3779B97F4A7C15ull >> 13; } };
    std::unordered_map<MKey, size_t, MHash> multiples;   // (column, value) -> first variable index
    size_t nbadd = 0, nbmul = 0, steps_ = 0;
public:
    struct Stats { uint64_t decs = 0, fresh_inst = 0, fresh_distinct = 0, rebuilds = 0, rebuild_scan = 0, select_scan = 0, aff_rows = 0, cand_rows = 0, max_level0 = 0, live_nnz_end = 0, cols_end = 0, sum_T = 0, max_T = 0, steps_l2 = 0, sum_T_l2 = 0, steps_le4 = 0, max_fresh = 0, live_keys_max = 0; } st;
private:

    void emit_mul(char c, size_t i, uint32_t e) { if (out) { *out << c << i; } if (!absOne(f, e)) { ++nbmul; if (out) *out << '*' << e; } }
    uint32_t &count_of(const Ref &r) { return r.dyn ? dyn[r.a][r.idx].cnt : cnt0[r.idx]; }
    bool find(uint32_t a, uint32_t b, uint32_t r, Ref &ref) const {
        if (b < S.n) {
            const uint64_t k = ((uint64_t)b << 32) | r;
            auto lo = S.keys.begin() + S.colptr[a], hi = S.keys.begin() + S.colptr[a + 1];
            auto it = std::lower_bound(lo, hi, k);
            if (it == hi || *it != k) return false;
            ref = Ref{a, 0, (size_t)(it - S.keys.begin())}; return true;
        }
        const auto &d = dyn[a];
        size_t lo = 0, hi = d.size();
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (d[mid].b < b || (d[mid].b == b && d[mid].r < r)) lo = mid + 1; else hi = mid; }
        if (lo == d.size() || d[lo].b != b || d[lo].r != r) return false;
        ref = Ref{a, 1, lo}; return true;
    }
    void dec(uint32_t a, uint32_t b, uint32_t r) {       // PairMap[triple]-- (:115-118)
        Ref ref; if (!find(a, b, r, ref)) throw std::logic_error("pair index: triple not found");
        ++st.decs;
        uint32_t &c = count_of(ref);
        if (c == 0) throw std::logic_error("pair index: negative frequency");
        --hist[c]; if (c == M) --cntM[a];
        --c; if (c) ++hist[c];
    }
    void rebuild_level() {                               // per-column counters for the current M
        std::fill(cntM.begin(), cntM.end(), 0u);
        ++st.rebuilds; st.rebuild_scan += active.size();
        size_t w = 0;
        for (size_t k = 0; k < active.size(); ++k) {
            const uint32_t c = count_of(active[k]);
            if (c >= 2) { active[w++] = active[k]; if (c == M) ++cntM[active[k].a]; }
        }
        active.resize(w);
    }
    // the (k)-th triple of frequency M in std::map order
    void select(uint64_t k, uint32_t &a, uint32_t &b, uint32_t &r) {
        uint32_t col = 0;
        for (;; ++col) { if (k < cntM[col]) break; k -= cntM[col]; }
        a = col; st.select_scan += col + (col < S.n ? S.colptr[col + 1] - S.colptr[col] : 0) + dyn[col].size();
        if (col < S.n)
            for (size_t x = S.colptr[col]; x < S.colptr[col + 1]; ++x)
                if (cnt0[x] == M) { if (k == 0) { b = (uint32_t)(S.keys[x] >> 32); r = (uint32_t)S.keys[x]; return; } --k; }
        for (auto &d : dyn[col]) if (d.cnt == M) { if (k == 0) { b = d.b; r = d.r; return; } --k; }
        throw std::logic_error("pair index: tie selection out of range");
    }
    static int pos_of(const std::vector<FEntry> &row, uint32_t c) {
        size_t lo = 0, hi = row.size();
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (row[mid].col < c) lo = mid + 1; else hi = mid; }
        return (lo < row.size() && row[lo].col == c) ? (int)lo : -1;
    }
    void grow_cols(size_t nc) { if (dyn.size() < nc) { dyn.resize(nc); colrows.resize(nc); ucount.resize(nc, 0); cntM.resize(nc, 0); } }

    // RemOneCSE :60-194 on the triple (a,b,r)
    void rem_one_cse(uint32_t a, uint32_t b, uint32_t r) {
        const uint32_t lm = (uint32_t)ncols;
        grow_cols(ncols + 1);
        const bool swap = ucount[a] < ucount[b];                      // :70-88
        const uint32_t l0 = swap ? b : a, l1 = swap ? a : b, rho = swap ? f.inv(r) : r;
        const auto &cand = colrows[a].size() <= colrows[b].size() ? colrows[a] : colrows[b];
        std::vector<uint32_t> aff;
        st.cand_rows += cand.size();
        for (uint32_t i : cand) {
            const auto &row = rows[i];
            int pa = pos_of(row, a), pb = pos_of(row, b);
            if (pa < 0 || pb < 0) continue;
            if (row[pb].val != f.mul(r, row[pa].val)) continue;
            if (!aff.empty() && aff.back() == i) continue;
            aff.push_back(i);
        }
        std::sort(aff.begin(), aff.end()); aff.erase(std::unique(aff.begin(), aff.end()), aff.end());
        st.aff_rows += aff.size();
        std::vector<uint64_t> fresh;                                   // (c<<32 | ratio) of the pairs with the new column
        for (uint32_t i : aff) {
            auto &row = rows[i];
            const int pa = pos_of(row, a), pb = pos_of(row, b);
            const FEntry ea = row[pa], eb = row[pb];
            for (size_t z = 0; z < row.size(); ++z) {
                if ((int)z == pa || (int)z == pb) continue;
                const FEntry &e = row[z];
                if (e.col < a) dec(e.col, a, f.mul(ea.val, e.inv)); else dec(a, e.col, f.mul(e.val, ea.inv));
                if (e.col < b) dec(e.col, b, f.mul(eb.val, e.inv)); else dec(b, e.col, f.mul(e.val, eb.inv));
            }
            dec(a, b, r);
            const FEntry co = (l0 == a) ? ea : eb;
            if (absOne(f, ea.val)) --ucount[a];
            if (absOne(f, eb.val)) --ucount[b];
            row.erase(row.begin() + pb); row.erase(row.begin() + pa);  // pa < pb
            for (const FEntry &e : row) fresh.push_back(((uint64_t)e.col << 32) | f.mul(co.val, e.inv));
            row.push_back(FEntry{lm, co.val, co.inv});
            if (absOne(f, co.val)) ++ucount[lm];
        }
        colrows[lm] = aff;
        std::sort(fresh.begin(), fresh.end());
        st.fresh_inst += fresh.size(); if (fresh.size() > st.max_fresh) st.max_fresh = fresh.size();
        for (size_t k = 0; k < fresh.size();) {                        // PairMap[newrow]++ (:140-142)
            ++st.fresh_distinct;
            size_t j = k; while (j < fresh.size() && fresh[j] == fresh[k]) ++j;
            const uint32_t c = (uint32_t)(fresh[k] >> 32), q = (uint32_t)fresh[k], mult = (uint32_t)(j - k);
            dyn[c].push_back(DEnt{lm, q, mult});
            if (hist.size() <= mult) hist.resize(mult + 1, 0);
            ++hist[mult];
            if (mult == M) ++cntM[c];
            if (mult >= 2) active.push_back(Ref{c, 1, dyn[c].size() - 1});
            k = j;
        }
        // multiplier reuse :153-169 and the two text lines :164-186
        const uint32_t asgs = f.abs(rho); size_t rindex = lm;
        if (!absOne(f, asgs)) {
            auto it = multiples.find(MKey{l1, asgs});
            if (it != multiples.end()) rindex = it->second;
            else {
                if (out) *out << rav << lm << ":=";
                emit_mul(tev, l1, asgs);
                if (out) *out << ";\n";
                multiples.emplace(MKey{l1, asgs}, lm);
            }
        }
        if (out) {
            *out << tev << lm << ":=" << tev << l0 << ((f.isMOne(asgs) || f.sign(rho) < 0) ? '-' : '+');
            if (absOne(f, asgs)) *out << tev << l1; else *out << rav << rindex;
            *out << ";\n";
        }
        ncols = lm + 1;
    }

    // ---- ProgramGen :513-611 on explicit row / column lists
    using TEnt = std::pair<uint32_t, uint32_t>;                       // (row, value) in a column list
    std::vector<std::vector<TEnt>> T;
    void build_T() {
        T.assign(ncols, {});
        for (uint32_t i = 0; i < rows.size(); ++i) for (auto &e : rows[i]) T[e.col].emplace_back(i, e.val);
    }
    void rows_from_T() {                                               // Transpose(M,T)
        for (auto &r : rows) r.clear();
        for (uint32_t j = 0; j < T.size(); ++j) for (auto &e : T[j]) rows[e.first].push_back(FEntry{j, e.second, 0});
        ncols = T.size();
    }
    void factor_out_columns(uint32_t j) {                              // :318-371
        if (T[j].empty()) return;
        std::vector<uint32_t> vals; for (auto &e : T[j]) vals.push_back(f.abs(e.second));
        std::sort(vals.begin(), vals.end());
        for (size_t k = 0; k < vals.size();) {
            size_t q = k; while (q < vals.size() && vals[q] == vals[k]) ++q;
            const uint32_t element = vals[k], freq = (uint32_t)(q - k); k = q;
            if (freq < 2 || absOne(f, element)) continue;
            size_t mvar = T.size(), rindex = mvar;
            auto it = multiples.find(MKey{j, element});
            if (it != multiples.end()) rindex = it->second;
            else {
                if (out) *out << rav << mvar << ":=";
                emit_mul(tev, j, element);
                if (out) *out << ";\n";
                multiples.emplace(MKey{j, element}, mvar);
            }
            if (out) *out << tev << mvar << ":=" << rav << rindex << ";\n";
            T.emplace_back();
            auto &src = T[j]; auto &dst = T[mvar]; size_t w = 0;
            for (size_t z = 0; z < src.size(); ++z) {
                if (f.abs(src[z].second) == element) dst.emplace_back(src[z].first, f.sign(src[z].second) >= 0 ? f.one() : f.mone());
                else src[w++] = src[z];
            }
            src.resize(w);
        }
    }
    // FactorOutRows :375-420 on row i; keeps T in step when `sync`
    void factor_out_rows(uint32_t i, bool sync) {
        auto &row = rows[i];
        if (row.empty()) return;
        std::vector<uint32_t> vals; for (auto &e : row) vals.push_back(f.abs(e.val));
        std::sort(vals.begin(), vals.end());
        for (size_t k = 0; k < vals.size();) {
            size_t q = k; while (q < vals.size() && vals[q] == vals[k]) ++q;
            const uint32_t element = vals[k], freq = (uint32_t)(q - k); k = q;
            if (freq < 2 || absOne(f, element)) continue;
            const uint32_t mvar = (uint32_t)ncols; ++ncols;
            if (out) *out << tev << mvar << ":=";
            bool first = true; size_t w = 0;
            for (size_t z = 0; z < row.size(); ++z) {
                if (f.abs(row[z].val) == element) {
                    if (first) { if (out) { if (f.sign(row[z].val) < 0) *out << '-'; *out << tev << row[z].col; } first = false; }
                    else { ++nbadd; if (out) *out << (f.sign(row[z].val) < 0 ? '-' : '+') << tev << row[z].col; }
                    if (sync) { auto &cl = T[row[z].col]; for (size_t y = 0; y < cl.size(); ++y) if (cl[y].first == i) { cl.erase(cl.begin() + y); break; } }
                } else row[w++] = row[z];
            }
            row.resize(w);
            row.push_back(FEntry{mvar, element, 0});
            if (out) *out << ";\n";
            if (sync) { T.emplace_back(); T[mvar].emplace_back(i, element); }
        }
    }
    bool third_exists(uint32_t i, uint32_t j, uint32_t quot) const {
        const uint32_t nq = f.neg(quot);
        for (auto &e : rows[i]) if (e.col != j && !absOne(f, e.val) && (e.val == quot || e.val == nq)) return true;
        return false;
    }
    void apply_triangle(uint32_t j, size_t it, size_t nx) {            // :453-498
        const TEnt iter = T[j][it], next = T[j][nx];
        const uint32_t i = next.first, quot = f.div(next.second, iter.second);
        const uint32_t mvar = (uint32_t)T.size();
        if (out) { *out << tev << mvar << ":="; if (f.sign(iter.second) < 0 || f.isMOne(iter.second)) *out << '-'; }
        emit_mul(tev, j, f.abs(iter.second));
        if (out) *out << ";\n";
        multiples.emplace(MKey{j, iter.second}, mvar);                // signed value, first match wins (:464)
        T.emplace_back();
        if (iter.first < next.first) { T[mvar].emplace_back(iter.first, f.one()); T[mvar].emplace_back(next.first, quot); }
        else { T[mvar].emplace_back(next.first, quot); T[mvar].emplace_back(iter.first, f.one()); }
        auto &cj = T[j];
        cj.erase(cj.begin() + std::max(it, nx)); cj.erase(cj.begin() + std::min(it, nx));
        for (uint32_t rr : {iter.first, next.first}) {
            auto &row = rows[rr];
            for (size_t z = 0; z < row.size(); ++z) if (row[z].col == j) { row.erase(row.begin() + z); break; }
            row.push_back(FEntry{mvar, rr == iter.first ? f.one() : quot, 0});
        }
        ncols = T.size();
        factor_out_rows(i, true);
    }
    void triangle(uint32_t j) {                                        // :427-507
        bool found = false;
        for (;;) {
            const auto &cj = T[j];
            std::vector<size_t> nu;                                    // positions of the non +-1 entries, row order
            for (size_t z = 0; z < cj.size(); ++z) if (!absOne(f, cj[z].second)) nu.push_back(z);
            if (nu.size() < 2) return;
            size_t hit_it = SIZE_MAX, hit_nx = SIZE_MAX;
            if (found) {                                               // only the first couple is looked at again
                const size_t it = nu[0], nx = nu[1];
                if (third_exists(cj[nx].first, j, f.div(cj[nx].second, cj[it].second))) { hit_it = it; hit_nx = nx; }
            } else {
                // value index: iter value v is accepted by `next` iff some third of row(next) equals +-(next/v)
                std::unordered_map<uint32_t, std::vector<size_t>> acc;
                for (size_t nx : nu) {
                    const uint32_t i = cj[nx].first, vn = cj[nx].second;
                    for (auto &e : rows[i]) {
                        if (e.col == j || absOne(f, e.val)) continue;
                        const uint32_t v = f.div(vn, e.val);
                        auto &l1 = acc[v]; if (l1.empty() || l1.back() != nx) l1.push_back(nx);
                        auto &l2 = acc[f.neg(v)]; if (l2.empty() || l2.back() != nx) l2.push_back(nx);
                    }
                }
                for (size_t it : nu) {
                    auto a = acc.find(cj[it].second);
                    if (a == acc.end()) continue;
                    for (size_t nx : a->second) if (nx != it) { hit_it = it; hit_nx = nx; break; }
                    if (hit_it != SIZE_MAX) break;
                }
            }
            if (hit_it == SIZE_MAX) return;
            found = true;
            apply_triangle(j, hit_it, hit_nx);
        }
    }
    void program_gen() {
        build_T();
        { const size_t nc = ncols; for (uint32_t j = 0; j < nc; ++j) factor_out_columns(j); }
        rows_from_T();
        for (uint32_t i = 0; i < rows.size(); ++i) factor_out_rows(i, false);
        build_T();
        for (uint32_t j = 0; j < T.size(); ++j) triangle(j);
        for (uint32_t i = 0; i < rows.size(); ++i) {                   // :547-604
            const auto &row = rows[i];
            if (row.empty()) { if (out) *out << ouv << i << ":=0;\n"; continue; }
            if (out) *out << ouv << i << ":=";
            for (size_t k = 0; k < row.size(); ++k) {
                const uint32_t v = row[k].val, ais = f.abs(v);
                if (k > 0) ++nbadd;
                auto it = multiples.find(MKey{row[k].col, ais});
                if (it != multiples.end()) {
                    if (out) { if (k == 0) { if (ais != v) *out << '-'; } else *out << (ais == v ? '+' : '-'); *out << rav << it->second; }
                } else {
                    const bool ng = f.sign(v) < 0 || f.isMOne(v);
                    if (out) { if (k == 0) { if (ng) *out << '-'; } else *out << (ng ? '-' : '+'); }
                    emit_mul(tev, row[k].col, ais);
                }
            }
            if (out) *out << ";\n";
        }
    }

public:
    FastCand(const SharedIndex &s, uint64_t seed, std::ostream *os = nullptr, char o = 'o', char t = 't', char r = 'r')
        : S(s), f(s.f), p(s.f.p), rng(seed), out(os), ouv(o), tev(t), rav(r), rows(s.rows), colrows(s.colrows),
          ucount(s.ucount), cnt0(s.cnt), ncols(s.n) {
        dyn.resize(ncols); cntM.assign(ncols, 0);
        uint32_t mx = 0; for (uint32_t c : cnt0) mx = std::max(mx, c);
        hist.assign(mx + 1, 0);
        for (size_t x = 0; x < cnt0.size(); ++x) ++hist[cnt0[x]];
        for (uint32_t a = 0; a < S.n; ++a) for (size_t x = S.colptr[a]; x < S.colptr[a + 1]; ++x) if (cnt0[x] >= 2) active.push_back(Ref{a, 0, x});
        M = mx; st.max_level0 = mx;
        rebuild_level();
    }
    size_t steps() const { return steps_; }
    // Optimizer(): while(OneSub) ; ProgramGen.  (A second OneSub call rebuilds the same table and stops.)
    std::pair<size_t, size_t> optimizer() {
        for (;;) {
            while (M >= 2 && hist[M] == 0) { --M; if (M >= 2 && hist[M]) rebuild_level(); }
            if (M <= 1) break;                                         // :255
            const uint64_t Tn = hist[M];
            st.sum_T += Tn; if (Tn > st.max_T) st.max_T = Tn; if (M == 2) { st.steps_l2++; st.sum_T_l2 += Tn; } if (M <= 4) st.steps_le4++;
            uint64_t k = 0;
            if (Tn > 1) k = rng.next() % Tn;                           // :260-265
            uint32_t a, b, r; select(k, a, b, r);
            ++nbadd; ++steps_;                                         // :292
            rem_one_cse(a, b, r);
        }
        st.cols_end = ncols; for (auto &r : rows) st.live_nnz_end += r.size();
        program_gen();
        return {nbadd, nbmul};
    }
};

} // namespace plo
#endif
