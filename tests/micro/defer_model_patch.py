import re
s = open('sim.cpp').read()
# deferred-model state
s = s.replace("    uint64_t livekeys2 = 0;\n", r'''    uint64_t livekeys2 = 0;
    // ---- deferred model: store (exact at last merge), log (key -> signed delta list), hot (exact)
    unordered_map<uint64_t, uint32_t> store, hot; vector<pair<uint64_t, int32_t>> dlog;
    for (size_t s2 = 0; s2 < T.k.size(); ++s2) if (T.k[s2] != ~0ull && T.c[s2] >= 2) store[T.k[s2]] = T.c[s2];
    uint32_t dtheta = maxf + 1; const uint64_t DH = 4096; vector<uint64_t> dhist(maxf + 2, 0);
    uint64_t nmerges = 0, maxlog = 0, hotprobes = 0, nlog = 0, forced = 0; const uint64_t LOGCAP = 3800000;
    auto merge = [&]() {
        ++nmerges; maxlog = max<uint64_t>(maxlog, dlog.size());
        for (auto &kv : hot) if (kv.second >= 2) dlog.push_back({kv.first, (int32_t)kv.second});
        hot.clear();
        unordered_map<uint64_t, int64_t> sum; for (auto &e : dlog) sum[e.first] += e.second; dlog.clear();
        for (auto &kv : sum) { auto it2 = store.find(kv.first); int64_t base = it2 == store.end() ? 0 : it2->second; int64_t nc = base + kv.second;
            if (nc >= 2) store[kv.first] = (uint32_t)nc; else if (it2 != store.end()) store.erase(it2); }
        fill(dhist.begin(), dhist.end(), 0); for (auto &kv : store) ++dhist[kv.second];
        uint32_t MM = maxf; while (MM >= 2 && dhist[MM] == 0) --MM;
        if (MM < 2) { dtheta = 2; return; }
        uint64_t acc = 0; uint32_t th = MM; for (uint32_t f = MM; f >= 2; --f) { if (acc + dhist[f] > DH && f != MM) break; acc += dhist[f]; th = f; }
        dtheta = th;
        for (auto it2 = store.begin(); it2 != store.end();) { if (it2->second >= th) { hot[it2->first] = it2->second; it2 = store.erase(it2); } else ++it2; }
    };
''')
# at step start: merge if needed, then check tie set equality
s = s.replace("        uint64_t Tn = hist[M]; if (Tn != level.size())", r'''        if (M < dtheta || dlog.size() > LOGCAP) { if (M >= dtheta) ++forced; merge(); }
        { // exactness check: hot keys at count M == level set
            uint64_t cntm = 0; for (auto &kv : hot) if (kv.second == M) { ++cntm; if (!level.count(kv.first)) { printf("hot key not in level\n"); return 1; } }
            if (cntm != level.size()) { printf("step %lu: deferred model tie set %lu vs %zu (M %u theta %u)\n", steps, cntm, level.size(), M, dtheta); return 1; } }
        uint64_t Tn = hist[M]; if (Tn != level.size())''')
# retire hook
s = s.replace("            --hist[o]; if (o == M) level.erase(kk);", r'''            { auto h2 = hot.find(kk); if (h2 != hot.end()) { ++hotprobes; if (h2->second < d) { printf("hot underflow\n"); exit(1); } h2->second -= d; } else { dlog.push_back({kk, -(int32_t)d}); ++nlog; } }
            --hist[o]; if (o == M) level.erase(kk);''')
# insert hook
s = s.replace("            if (d >= 2) ++C.ins2; else ++C.ins1;", r'''            { const bool spilled = ((kk * 0x9E3779B97F4A7C15ull) >> 60) == 0;   // emulate a spill: unaggregated +1s into the hot table
              if (spilled) hot[kk] += d; else if (d >= dtheta) hot[kk] = d; else if (d >= 2) { dlog.push_back({kk, (int32_t)d}); ++nlog; } }
            if (d >= 2) ++C.ins2; else ++C.ins1;''')
s = s.replace('    printf("steps %lu cols %zu\\n", steps, ncols);', r'''    printf("steps %lu cols %zu\n", steps, ncols);
    printf("deferred model: merges %lu (forced by log %lu) max log %lu log entries %lu hot ops %lu final store %zu\n", nmerges, forced, maxlog, nlog, hotprobes, store.size());''')
open('sim2.cpp','w').write(s)
