// tests/host/walk_harness.cpp -- TEST INFRASTRUCTURE.  Compiles the product's serial walker (csrc/walker.h)
// with g++ and drives it on the CPU, so the walk logic (phases, guards, cycle fast-forward) can be checked
// against the golden traces and the oracle without a GPU.  The surrounding prep (components, state bytes) is a
// plain host re-implementation that exists only in this harness.
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include "../../omnirevolve-image-processor_amd/csrc/walker.h"

extern "C" int walk_harness(const uint8_t* skel, int H, int W, int64_t* off_out, int64_t off_cap, int32_t* pts_out, int64_t pts_cap,
                            int64_t* n_paths, int64_t* n_pts, int cap_factor) {
    size_t N = (size_t)H * W;
    std::vector<uint8_t> st(N, 0);
    std::vector<int> root(N, -1);
    int Wb = (W + 1) / 2;
    auto pid = [&](int y, int x) { return (((y >> 1) * Wb + (x >> 1)) << 2) | ((y & 1) << 1) | (x & 1); };
    // flood fill components, root = min block-raster id
    std::vector<size_t> stack, members;
    for (size_t s = 0; s < N; s++) {
        if (!skel[s] || root[s] >= 0) continue;
        members.clear(); stack.push_back(s); root[s] = 0; int best = 1 << 30;
        while (!stack.empty()) {
            size_t i = stack.back(); stack.pop_back(); members.push_back(i);
            int y = (int)(i / W), x = (int)(i % W); best = std::min(best, pid(y, x));
            for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                int yy = y + dy, xx = x + dx; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                size_t j = (size_t)yy * W + xx; if (skel[j] && root[j] < 0) { root[j] = 0; stack.push_back(j); }
            }
        }
        for (size_t i : members) root[i] = best;
    }
    std::vector<std::pair<unsigned, unsigned>> kl;   // (key, lin) raster order then stable sort
    for (size_t i = 0; i < N; i++) if (skel[i]) {
        int y = (int)(i / W), x = (int)(i % W), deg = 0;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { if (!dy && !dx) continue; int yy = y + dy, xx = x + dx; if (yy >= 0 && yy < H && xx >= 0 && xx < W && skel[(size_t)yy * W + xx]) deg++; }
        st[i] = ST_FG | (deg == 1 ? ST_END : 0) | (deg >= 3 ? ST_JUN : 0);
        kl.push_back({(unsigned)root[i], (unsigned)i});
    }
    *n_paths = 0; *n_pts = 0; off_out[0] = 0;
    if (kl.empty()) return 0;
    std::stable_sort(kl.begin(), kl.end(), [](auto& a, auto& b) { return a.first < b.first; });
    unsigned M = (unsigned)kl.size();
    std::vector<unsigned> keys(M), lin(M), cs;
    for (unsigned i = 0; i < M; i++) { keys[i] = kl[i].first; lin[i] = kl[i].second; if (i == 0 || keys[i] != keys[i - 1]) cs.push_back(i); }
    unsigned NC = (unsigned)cs.size(); cs.push_back(M);
    WalkArgs A; memset(&A, 0, sizeof(A));
    A.H = H; A.W = W; A.plane = (int64_t)N; A.st = st.data(); A.keys = keys.data(); A.lin = lin.data(); A.comp_start = cs.data(); A.nc = NC;
    A.total_fg[0] = M; A.comp_order = nullptr;
    const unsigned F = cap_factor > 0 ? (unsigned)cap_factor : 16u;
    std::vector<unsigned> memo((size_t)N * 8, 0), logbuf(((size_t)F * M + 64 * (size_t)NC + 8) * 4, 0);
    std::vector<uint8_t> steplog((size_t)F * M + 256 * (size_t)NC + 8, 0);
    std::vector<WalkInfo> winfo((size_t)2 * M); memset(winfo.data(), 0, winfo.size() * sizeof(WalkInfo));
    int over = 0;
    A.memo = memo.data(); A.logbuf = logbuf.data(); A.steplog = steplog.data(); A.cap_factor = F; A.winfo = winfo.data(); A.overflow = &over;
    for (unsigned c = 0; c < NC; c++) trace_component(A, c);
    if (over) return 2;     // the product retries with a larger factor; the test asks for one explicitly
    const unsigned nslots = 2 * M;
    for (unsigned i = 0; i < nslots; i++) walk_close_tail(A, PlainReader(), i, winfo[i]);      // what k_winfo_lens does on the GPU
    std::vector<unsigned long long> pts_off(nslots + 1, 0); std::vector<unsigned> path_off(nslots + 1, 0);
    for (unsigned i = 0; i < nslots; i++) { pts_off[i + 1] = pts_off[i] + winfo[i].len_kept; path_off[i + 1] = path_off[i] + (winfo[i].len_kept ? 1u : 0u); }
    *n_paths = path_off[nslots]; *n_pts = (int64_t)pts_off[nslots];
    if (*n_paths + 1 > off_cap || *n_pts > pts_cap) return 1;
    A.pts_off = pts_off.data(); A.path_off = path_off.data(); A.pts[0] = pts_out; A.off[0] = off_out;
    std::vector<unsigned> kept; std::vector<unsigned long long> kept_off;
    for (unsigned i = 0; i < nslots; i++) if (winfo[i].len_kept) { kept.push_back(i); kept_off.push_back(pts_off[i]); }
    const unsigned long long total = pts_off[nslots], chunk = 1000;       // the write pass in chunks, as k_write_walks cuts it (odd size: chunks start inside walks)
    for (unsigned long long p0 = 0; p0 < total; p0 += chunk) write_chunk(A, 0, kept.data(), kept_off.data(), (unsigned)kept.size(), p0, std::min(total, p0 + chunk));
    return 0;
}
