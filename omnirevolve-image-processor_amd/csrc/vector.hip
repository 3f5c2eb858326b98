// csrc/vector.hip -- stage 05 (_scale_one, 05:82-96), stage 07 (reorder_one_color, 07:19-95) and
// stage 12 (_build_ops_for_layer, 12:85-187) on gfx950.
#include "vec_common.h"

// ------------------------------------------------------------------------------------------------
// Stage 05: pts.astype(f32) @ S.T + T in float32, truncation to int32.  8 B in + 8 B out per point.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale_pts(const int2* __restrict__ in, int2* __restrict__ out, int64_t n, float sx, float sy, float dx, float dy) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int2 p = in[i];
        float x = __fadd_rn(__fmul_rn((float)p.x, sx), dx), y = __fadd_rn(__fmul_rn((float)p.y, sy), dy);
        out[i] = make_int2((int)x, (int)y);
    }
}

// the log pixels in use (ent_idx[0 .. *n)), scaled
__global__ __launch_bounds__(256) void k_scale_ents(const unsigned* __restrict__ ent_idx, const unsigned* __restrict__ n, const int2* __restrict__ in, int2* __restrict__ out,
                                                     float sx, float sy, float dx, float dy) {
    const unsigned m = *n;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const unsigned e = ent_idx[i]; const int2 p = in[e];
        out[e] = make_int2((int)__fadd_rn(__fmul_rn((float)p.x, sx), dx), (int)__fadd_rn(__fmul_rn((float)p.y, sy), dy));
    }
}

// explicit points of a walk-coded list, on request (orip_get_polys, consumers that read int32 pairs); enqueued on the calling lane's stream
int orip_polys_materialize(orip_ctx* c, DPolys& P) {
    if (!is_coded(P)) return 0;
    VSrc src; ORIP_TRY(vsrc_of(c, P, src));
    HIPC(c, P.pts.ensure((size_t)std::max<int64_t>(P.total, 1) * 8 + 64));
    if (P.n > 0 && P.total > 0) {
        ProfScope ps(c, "k_expand_pts");
        hipLaunchKernelGGL(k_expand_pts<VSrc>, dim3((unsigned)cdiv(P.total, 4096)), dim3(256), 0, LN(c).stream, src, P.n, reinterpret_cast<int2*>(P.pts.p), P.total);
    }
    HIPC(c, hipGetLastError());
    P.pts_ok = true;
    return 0;
}

// `sync`: the public entry points return with the lane's stream drained (the caller may read the slot from another lane next);
// orip_layer_front chains the stages of a layer on one stream and skips the waits in between
int orip_scale_vectors_impl(orip_ctx* c, int layer, float sx, float sy, float dx, float dy, bool sync) {
    if (layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", layer);
    ORIP_LANE(c, layer + 1);
    DPolys& S = c->polys[ORIP_SLOT_CONTOURS][layer]; DPolys& D = c->polys[ORIP_SLOT_SCALED][layer];
    D.n = S.n; D.total = S.total;
    if (is_coded(S) && !S.scaled) {
        // walk-coded contours: the scaled list is the same walks over scaled copies of the two small point tables (own points, log pixels) --
        // _scale_one runs over ~1e6 distinct points of a heavy layer instead of its 2.8e8 list points
        WalkStore& WS = c->wstore[S.vlayer];
        if (S.vepoch != WS.epoch) ORIP_FAIL(c, "the walk records of layer %d behind CONTOURS have been replaced by a newer trace", S.vlayer);
        HIPC(c, WS.own_s.ensure((size_t)std::max<int64_t>(WS.n_own, 1) * 8 + 64));
        HIPC(c, WS.lxy_s.ensure(WS.lxy.cap));
        WS.sx = sx; WS.sy = sy; WS.dx = dx; WS.dy = dy; WS.sepoch++;
        {
            ProfScope ps(c, "k_scale_pts");
            if (WS.n_own) hipLaunchKernelGGL(k_scale_pts, dim3((unsigned)std::min<int64_t>(cdiv(WS.n_own, 256), 8192)), dim3(256), 0, LN(c).stream, WS.own.as<int2>(), WS.own_s.as<int2>(), WS.n_own, sx, sy, dx, dy);
            hipLaunchKernelGGL(k_scale_ents, dim3((unsigned)std::min<int64_t>(cdiv(std::max<int64_t>(WS.ent_cap, 1), 256), 2048)), dim3(256), 0, LN(c).stream, WS.ent_idx.as<unsigned>(), WS.cnt.as<unsigned>(),
                               WS.lxy.as<int2>(), WS.lxy_s.as<int2>(), sx, sy, dx, dy);
        }
        D.virt = true; D.pts_ok = false; D.vident = S.vident; D.vlayer = S.vlayer; D.vepoch = S.vepoch; D.scaled = true; D.vsepoch = WS.sepoch;
        HIPC(c, D.off.ensure((size_t)(S.n + 1) * 8 + 64));
        HIPC(c, hipMemcpyAsync(D.off.p, S.off.p, (size_t)(S.n + 1) * 8, hipMemcpyDeviceToDevice, LN(c).stream));
        if (!S.vident) { HIPC(c, D.vview.ensure((size_t)std::max<int64_t>(S.n, 1) * sizeof(VView) + 64)); HIPC(c, hipMemcpyAsync(D.vview.p, S.vview.p, (size_t)S.n * sizeof(VView), hipMemcpyDeviceToDevice, LN(c).stream)); }
        HIPC(c, hipGetLastError());
        if (sync) HIPC(c, hipStreamSynchronize(LN(c).stream));
        return 0;
    }
    ORIP_TRY(orip_polys_materialize(c, S));
    D.set_explicit();
    HIPC(c, D.off.ensure((size_t)(S.n + 1) * 8 + 64));
    HIPC(c, D.pts.ensure((size_t)std::max<int64_t>(S.total, 1) * 8 + 64));
    if (S.n == 0) { HIPC(c, hipMemsetAsync(D.off.p, 0, 8, LN(c).stream)); return 0; }
    HIPC(c, hipMemcpyAsync(D.off.p, S.off.p, (size_t)(S.n + 1) * 8, hipMemcpyDeviceToDevice, LN(c).stream));
    if (S.total) {
        ProfScope ps(c, "k_scale_pts");
        hipLaunchKernelGGL(k_scale_pts, dim3((unsigned)std::min<int64_t>(cdiv(S.total, 256), 8192)), dim3(256), 0, LN(c).stream,
                           reinterpret_cast<const int2*>(S.pts.p), reinterpret_cast<int2*>(D.pts.p), S.total, sx, sy, dx, dy);
    }
    HIPC(c, hipGetLastError());
    if (sync) HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_scale_vectors(orip_ctx* c, int layer, float sx, float sy, float dx, float dy) {
    orip_enter(c);
    return orip_scale_vectors_impl(c, layer, sx, sy, dx, dy, true);
}

static int hook_prefetch08(orip_ctx* c, void* arg, DPolys& src, const PolyFeat* feat) { return orip_prefetch08(c, arg, src, feat); }
int orip_sort_contours_impl(orip_ctx* c, int layer, bool sync, const orip_params08* prm_for_prefetch) {
    if (layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", layer);
    ORIP_LANE(c, layer + 1);
    LN(c).pf08.valid = false;
    DPolys& S = c->polys[ORIP_SLOT_SCALED][layer]; DPolys& D = c->polys[ORIP_SLOT_SORTED][layer];
    const bool pf = prm_for_prefetch && is_coded(S) && S.vident && S.n > 0 && !getenv("ORIP_NO_PREFETCH08");
    ORIP_TRY(vreorder(c, S, D, 7, pf ? hook_prefetch08 : nullptr, (void*)prm_for_prefetch));
    if (pf && LN(c).pf08.valid) D.pf_tag = LN(c).pf08.tag;
    if (sync) HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_sort_contours(orip_ctx* c, int layer) {
    orip_enter(c);
    return orip_sort_contours_impl(c, layer, true);
}

// ------------------------------------------------------------------------------------------------
// Stage 12.  All positions are integers (line ends, tap centres), so every hypot comparison of the
// reference is evaluated on exact squared distances; ties resolve in the reference's scan order
// (line k start, line k end, ..., then taps).  One 1024-thread block per layer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long d2i(long long ax, long long ay, long long bx, long long by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

__global__ __launch_bounds__(1024) void k_plot_order(const PolyFeat* __restrict__ lf, int nl, const int32_t* __restrict__ taps, int nt, double R,
                                                      uint8_t* __restrict__ alive_l, uint8_t* __restrict__ alive_t, int32_t* __restrict__ ops, int* __restrict__ n_ops_out) {
    __shared__ unsigned long long wb[16];
    __shared__ long long px, py; __shared__ int nops, cursor, found;
    const int tid = threadIdx.x;
    for (int i = tid; i < nl; i += 1024) alive_l[i] = 1;
    for (int i = tid; i < nt; i += 1024) alive_t[i] = 1;
    if (tid == 0) { px = 0; py = 0; nops = 0; }
    __syncthreads();
    auto emit_line = [&](int k, int flip) { int o = nops++; ops[5 * o] = 0; ops[5 * o + 1] = k; ops[5 * o + 2] = flip; ops[5 * o + 3] = 0; ops[5 * o + 4] = 0; };
    auto emit_tap = [&](int t) { int o = nops++; ops[5 * o] = 1; ops[5 * o + 1] = -1; ops[5 * o + 2] = 0; ops[5 * o + 3] = taps[2 * t]; ops[5 * o + 4] = taps[2 * t + 1]; };
    auto block_min = [&](unsigned long long v) -> unsigned long long {
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(v, o, 64); if (t < v) v = t; }
        __syncthreads();
        if ((tid & 63) == 0) wb[tid >> 6] = v;
        __syncthreads();
        unsigned long long b = wb[0];
        for (int w = 1; w < 16; w++) if (wb[w] < b) b = wb[w];
        return b;
    };
    // drain (12:120-127, 174-181): single forward pass over the alive taps, the cursor follows each drained tap
    auto drain = [&]() {
        if (tid == 0) cursor = 0;
        __syncthreads();
        while (true) {
            int cur = cursor; long long x = px, y = py;
            unsigned long long best = ~0ULL;
            for (int t = cur + tid; t < nt; t += 1024) {
                if (!alive_t[t]) continue;
                long long q = d2i(x, y, taps[2 * t], taps[2 * t + 1]);
                if (sqrt((double)q) <= R) { best = (unsigned long long)t; break; }   // smallest index of this lane's stride
            }
            unsigned long long b = block_min(best);
            if (b == ~0ULL) break;
            if (tid == 0) { int t = (int)b; emit_tap(t); alive_t[t] = 0; px = taps[2 * t]; py = taps[2 * t + 1]; cursor = t + 1; }
            __syncthreads();
        }
        __syncthreads();
    };
    if (nl > 0) {
        // first = longest line (first maximum), entered from the end nearer to (0,0) (strict <)
        unsigned long long best = ~0ULL;
        for (int k = tid; k < nl; k += 1024) {
            unsigned long long key = ((unsigned long long)(~__float_as_uint(lf[k].per)) << 32) | (unsigned)k;   // per >= 0: larger per -> smaller key
            if (key < best) best = key;
        }
        unsigned long long b = block_min(best);
        if (tid == 0) {
            int k = (int)(b & 0xffffffffu);
            PolyFeat f = lf[k];
            int flip = d2i(0, 0, f.ex, f.ey) < d2i(0, 0, f.sx, f.sy);
            emit_line(k, flip); alive_l[k] = 0;
            if (flip) { px = f.sx; py = f.sy; } else { px = f.ex; py = f.ey; }
        }
        __syncthreads();
        drain();
    } else if (nt > 0) {
        unsigned long long best = ~0ULL;
        for (int t = tid; t < nt; t += 1024) {
            unsigned long long key = ((unsigned long long)d2i(0, 0, taps[2 * t], taps[2 * t + 1]) << 32) | (unsigned)t;
            if (key < best) best = key;
        }
        unsigned long long b = block_min(best);
        if (tid == 0) { int t = (int)(b & 0xffffffffu); emit_tap(t); alive_t[t] = 0; px = taps[2 * t]; py = taps[2 * t + 1]; }
        __syncthreads();
    }
    const int total = nl + nt;
    while (true) {
        __syncthreads();
        if (nops >= total) break;
        long long x = px, y = py;
        unsigned long long best = ~0ULL;
        for (int k = tid; k < nl; k += 1024) {
            if (!alive_l[k]) continue;
            unsigned long long k1 = ((unsigned long long)d2i(x, y, lf[k].sx, lf[k].sy) << 32) | (unsigned)(2 * k);
            unsigned long long k2 = ((unsigned long long)d2i(x, y, lf[k].ex, lf[k].ey) << 32) | (unsigned)(2 * k + 1);
            if (k1 < best) best = k1;
            if (k2 < best) best = k2;
        }
        for (int t = tid; t < nt; t += 1024) {
            if (!alive_t[t]) continue;
            unsigned long long kt = ((unsigned long long)d2i(x, y, taps[2 * t], taps[2 * t + 1]) << 32) | (unsigned)(2 * nl + t);
            if (kt < best) best = kt;
        }
        unsigned long long b = block_min(best);
        if (b == ~0ULL) break;
        unsigned idx = (unsigned)(b & 0xffffffffu);
        bool is_line = idx < (unsigned)(2 * nl);
        if (tid == 0) {
            if (is_line) {
                int k = idx >> 1, flip = idx & 1;
                emit_line(k, flip); alive_l[k] = 0;
                if (flip) { px = lf[k].sx; py = lf[k].sy; } else { px = lf[k].ex; py = lf[k].ey; }
            } else {
                int t = idx - 2 * nl;
                emit_tap(t); alive_t[t] = 0; px = taps[2 * t]; py = taps[2 * t + 1];
            }
        }
        __syncthreads();
        if (is_line) drain();
    }
    if (tid == 0) *n_ops_out = nops;
    (void)found;
}

// The same greedy as ONE wavefront over LDS-resident end points, taps and alive flags (a layer has a few hundred to a few thousand
// ops): every choice is a 64-wide scan + DPP minimum with no barrier and no global-memory round trip; the ops leave through a
// 64-entry ring.  Same keys, same tie-breaks (line k start, line k end, ..., then taps; exact integer d^2).
__global__ __launch_bounds__(64) void k_plot_order_wave(const PolyFeat* __restrict__ lf, int nl, const int32_t* __restrict__ taps, int nt, double R,
                                                         int32_t* __restrict__ ops, int* __restrict__ n_ops_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    int4* E = reinterpret_cast<int4*>(smem);                       // (sx, sy, ex, ey)
    int2* T = reinterpret_cast<int2*>(E + nl);
    uint8_t* aliveL = reinterpret_cast<uint8_t*>(T + nt); uint8_t* aliveT = aliveL + nl;
    __shared__ int ring[64 * 5];
    const int lane = threadIdx.x;
    for (int i = lane; i < nl; i += 64) { PolyFeat f = lf[i]; E[i] = make_int4(f.sx, f.sy, f.ex, f.ey); aliveL[i] = 1; }
    for (int i = lane; i < nt; i += 64) { T[i] = make_int2(taps[2 * i], taps[2 * i + 1]); aliveT[i] = 1; }
    __syncthreads();
    long long px = 0, py = 0; int nops = 0;
    auto emit = [&](int type, int k, int flip, int x, int y) {
        if (lane == 0) { int* r = ring + 5 * (nops & 63); r[0] = type; r[1] = k; r[2] = flip; r[3] = x; r[4] = y; }
        nops++;
        if ((nops & 63) == 0) { const int base = nops - 64; for (int q = lane; q < 320; q += 64) ops[5 * base + q] = ring[q]; }
    };
    auto drain = [&]() {          // 12:120-127, 174-181: one forward pass over the alive taps, the cursor follows each drained tap
        int cursor = 0;
        while (cursor < nt) {
            int found = -1;
            for (int t0 = cursor; t0 < nt && found < 0; t0 += 64) {
                const int t = t0 + lane; bool ok = false;
                if (t < nt && aliveT[t]) { const int2 p = T[t]; ok = sqrt((double)d2i(px, py, p.x, p.y)) <= R; }
                const unsigned long long m = __ballot(ok);
                if (m) found = t0 + __ffsll((long long)m) - 1;
            }
            if (found < 0) break;
            const int2 p = T[found];
            if (lane == 0) aliveT[found] = 0;
            emit(1, -1, 0, p.x, p.y);
            px = p.x; py = p.y; cursor = found + 1;
        }
    };
    if (nl > 0) {
        // first = longest line (first maximum), entered from the end nearer to (0,0) (strict <)
        unsigned long long best = ~0ULL;
        for (int k = lane; k < nl; k += 64) {
            unsigned long long key = ((unsigned long long)(~__float_as_uint(lf[k].per)) << 32) | (unsigned)k;   // per >= 0: larger per -> smaller key
            if (key < best) best = key;
        }
        best = wave_min_key(best);
        const int k = (int)(best & 0xffffffffu);
        const int4 e = E[k];
        const int flip = d2i(0, 0, e.z, e.w) < d2i(0, 0, e.x, e.y);
        if (lane == 0) aliveL[k] = 0;
        emit(0, k, flip, 0, 0);
        if (flip) { px = e.x; py = e.y; } else { px = e.z; py = e.w; }
        drain();
    } else if (nt > 0) {
        unsigned long long best = ~0ULL;
        for (int t = lane; t < nt; t += 64) {
            const int2 p = T[t];
            unsigned long long key = ((unsigned long long)d2i(0, 0, p.x, p.y) << 32) | (unsigned)t;
            if (key < best) best = key;
        }
        best = wave_min_key(best);
        const int t = (int)(best & 0xffffffffu);
        const int2 p = T[t];
        if (lane == 0) aliveT[t] = 0;
        emit(1, -1, 0, p.x, p.y);
        px = p.x; py = p.y;
    }
    const int total = nl + nt;
    while (nops < total) {
        unsigned long long best = ~0ULL;
        for (int k = lane; k < nl; k += 64) {
            if (!aliveL[k]) continue;
            const int4 e = E[k];
            unsigned long long k1 = ((unsigned long long)d2i(px, py, e.x, e.y) << 32) | (unsigned)(2 * k);
            unsigned long long k2 = ((unsigned long long)d2i(px, py, e.z, e.w) << 32) | (unsigned)(2 * k + 1);
            if (k1 < best) best = k1;
            if (k2 < best) best = k2;
        }
        for (int t = lane; t < nt; t += 64) {
            if (!aliveT[t]) continue;
            const int2 p = T[t];
            unsigned long long kt = ((unsigned long long)d2i(px, py, p.x, p.y) << 32) | (unsigned)(2 * nl + t);
            if (kt < best) best = kt;
        }
        best = wave_min_key(best);
        if (best == ~0ULL) break;
        const unsigned idx = (unsigned)(best & 0xffffffffu);
        if (idx < (unsigned)(2 * nl)) {
            const int k = (int)(idx >> 1), flip = (int)(idx & 1u);
            const int4 e = E[k];
            if (lane == 0) aliveL[k] = 0;
            emit(0, k, flip, 0, 0);
            if (flip) { px = e.x; py = e.y; } else { px = e.z; py = e.w; }
            drain();
        } else {
            const int t = (int)(idx - 2u * (unsigned)nl);
            const int2 p = T[t];
            if (lane == 0) aliveT[t] = 0;
            emit(1, -1, 0, p.x, p.y);
            px = p.x; py = p.y;
        }
    }
    { const int base = nops & ~63; for (int q = lane; q < 5 * (nops - base); q += 64) ops[5 * base + q] = ring[q]; }
    if (lane == 0) *n_ops_out = nops;
}

extern "C" int orip_plot_order(orip_ctx* c, int layer, double R_insert, int64_t* n_ops) {
    orip_enter(c);
    if (layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", layer);
    ORIP_LANE(c, layer + 1);
    DPolys& L = c->polys[ORIP_SLOT_LINES_CROSS][layer]; DTaps& T = c->taps[ORIP_TAPS_CROSS][layer];
    if (c->cross_unordered[layer]) {                 // stage 10's travel reorder (10:253), deferred to this lane by orip_dedup_cross_layer_deferred
        DPolys& tmp = LN(c).tp[5];
        ORIP_TRY(vreorder(c, L, tmp, 10));
        std::swap(tmp.off, L.off); std::swap(tmp.pts, L.pts); L.n = tmp.n; L.total = tmp.total;
        c->cross_unordered[layer] = false;
    }
    int64_t nl = L.n, nt = T.n;
    c->n_ops[layer] = 0; *n_ops = 0;
    if (nl + nt == 0) return 0;
    if (nl + nt > 0x3fffffff) ORIP_FAIL(c, "too many ops");
    HIPC(c, LN(c).vtmp[6].ensure((size_t)std::max<int64_t>(nl, 1) * sizeof(PolyFeat) + (size_t)(nl + nt) + 256));
    PolyFeat* feat = LN(c).vtmp[6].as<PolyFeat>();
    uint8_t* alive_l = (uint8_t*)(feat + std::max<int64_t>(nl, 1)); uint8_t* alive_t = alive_l + nl;
    HIPC(c, c->ops[layer].ensure((size_t)(nl + nt) * 20 + 64));
    HIPC(c, T.xy.ensure(64));
    if (nl) ORIP_TRY(vfeatures(c, L, 2, feat));
    int* d_n = LN(c).flags.as<int>() + 40;
    const size_t lds = (size_t)nl * 17 + (size_t)nt * 9 + 64;
    if (lds <= 150 * 1024 && !getenv("ORIP_PLOT_1WG")) {
        static std::once_flag attr_once;            // several layer threads may arrive here together
        static std::atomic<int> attr_err{0};
        std::call_once(attr_once, [&] { orip_max_lds(k_plot_order_wave, 150 * 1024, attr_err); });
        if (attr_err.load()) ORIP_FAIL(c, "hipFuncSetAttribute(k_plot_order_wave) failed: %s", hipGetErrorString((hipError_t)attr_err.load()));
        ProfScope ps(c, "k_plot_order");
        hipLaunchKernelGGL(k_plot_order_wave, dim3(1), dim3(64), lds, LN(c).stream, feat, (int)nl, T.xy.as<int32_t>(), (int)nt, R_insert, c->ops[layer].as<int32_t>(), d_n);
    } else { ProfScope ps(c, "k_plot_order"); hipLaunchKernelGGL(k_plot_order, dim3(1), dim3(1024), 0, LN(c).stream, feat, (int)nl, T.xy.as<int32_t>(), (int)nt, R_insert, alive_l, alive_t, c->ops[layer].as<int32_t>(), d_n); }
    HIPC(c, hipGetLastError());
    int h = 0;
    ORIP_TRY(vread(c, &h, d_n));
    c->n_ops[layer] = h; *n_ops = h;
    return 0;
}

extern "C" int orip_get_ops(orip_ctx* c, int layer, int32_t* ops5) {
    orip_enter(c);
    if (layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", layer);
    if (!c->n_ops[layer]) return 0;
    HIPC(c, hipMemcpyAsync(ops5, c->ops[layer].p, (size_t)c->n_ops[layer] * 20, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}

