// csrc/vsrc.h -- where the kernels of stages 05 / 07 / 08-front take the points of a polyline list from.
//
//   ESrc : an explicit list (off int64[n+1], pts int32 pairs), e.g. one set with orip_set_polys
//   VSrc : a walk-coded list (orip_ctx.h: DPolys::virt, walker.h: VWalk / VPiece / VView): point k of polyline i is
//          point t = first + k (or first + len - 1 - k when reversed) of walk `wid`, which is its start pixel (t == 0, and again as
//          the closing point), one of its explicit own points (t <= n_own) or the pixel of a log entry of its bounce tail; _scale_one
//          (05:82-96: float32 multiply, add, truncation) is applied on the way out when the list is a scaled one.
// Both give `len(i)` and a cursor `cur(i)` with `at(k)`; kernels are templated on the source, so the arithmetic they do on the points
// -- and with it every rounding the reference's results depend on -- is the same code for both.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "walker.h"

struct ESrc {
    const int64_t* __restrict__ off; const int2* __restrict__ pts;
    struct Cur {
        const int2* p;
        __device__ __forceinline__ int2 at(int64_t k) const { return p[k]; }
    };
    __device__ __forceinline__ int64_t len(int64_t i) const { return off[i + 1] - off[i]; }
    __device__ __forceinline__ Cur cur(int64_t i) const { return Cur{pts + off[i]}; }
};

struct VGeom {                                // what the points of a layer's walks are made of (wave-uniform: lives in SGPRs)
    const VPiece* __restrict__ piece; const int2* __restrict__ own;
    const unsigned* __restrict__ logw;        // state word of log entry e: logw[4 * e] = (pixel index << 3) | direction
    unsigned W; unsigned long long wmagic;    // pixel index -> (x, y): y = (lin * wmagic) >> 40, exact for lin < 2^26, W <= 8192
    int scaled; float sx, sy, dx, dy;
};
struct VSrc {
    const int64_t* __restrict__ off;
    const VView* __restrict__ view;           // nullptr: polyline i = walk i, whole, forward
    const VWalk* __restrict__ walk;
    VGeom g;
    struct Cur {
        VGeom g; VWalk w; unsigned first, len, rev;
        __device__ __forceinline__ int2 at(int64_t k) const {
            unsigned t = rev ? first + len - 1u - (unsigned)k : first + (unsigned)k;
            if ((w.flags & 1u) && t == w.len - 1u) t = 0u;
            int2 p;
            if (t <= w.n_own) p = g.own[w.own_off + t];
            else {
                const unsigned u = t - w.n_own - 1u;
                unsigned j = w.n_piece - 1u;
                while (j > 0u && g.piece[w.piece_off + j].u0 > u) j--;
                const VPiece q = g.piece[w.piece_off + j];
                unsigned d = u - q.u0; if (q.lam) d %= q.lam;
                const unsigned lin = g.logw[4ull * (q.ent + d)] >> 3;
                const unsigned y = (unsigned)(((unsigned long long)lin * g.wmagic) >> 40);
                p = make_int2((int)(lin - y * g.W), (int)y);
            }
            if (g.scaled) { p.x = (int)__fadd_rn(__fmul_rn((float)p.x, g.sx), g.dx); p.y = (int)__fadd_rn(__fmul_rn((float)p.y, g.sy), g.dy); }
            return p;
        }
    };
    __device__ __forceinline__ int64_t len(int64_t i) const { return off[i + 1] - off[i]; }
    __device__ __forceinline__ Cur cur(int64_t i) const {
        Cur c; c.g = g;
        if (view) { const VView v = view[i]; c.w = walk[v.wid]; c.first = v.first; c.len = v.len; c.rev = v.rev; }
        else { c.w = walk[i]; c.first = 0u; c.len = c.w.len; c.rev = 0u; }
        return c;
    }
};

// the int32 x,y view the serial helpers of vec_serial.h index (xy[2 * i], xy[2 * i + 1]) over a cursor: a small polyline copied to registers / scratch
template <class Cur, int CAP>
struct LocalPts {
    int32_t xy[2 * CAP];
    __device__ __forceinline__ void load(const Cur& c, int n) { for (int i = 0; i < n; i++) { const int2 p = c.at(i); xy[2 * i] = p.x; xy[2 * i + 1] = p.y; } }
};
