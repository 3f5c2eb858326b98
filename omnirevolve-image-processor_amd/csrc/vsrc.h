// csrc/vsrc.h -- where the kernels of stages 05 / 07 / 08-front take the points of a polyline list from.
//
//   ESrc : an explicit list (off int64[n+1], pts int32 pairs), e.g. one set with orip_set_polys
//   VSrc : a walk-coded list (orip_ctx.h: DPolys::virt, walker.h: VWalk / VPiece / VView): point k of polyline i is
//          point t = first + k (or first + len - 1 - k when reversed) of walk `wid`, which is its start pixel (t == 0, and again as
//          the closing point), one of its explicit own points (t <= n_own) or the pixel of a log entry of its bounce tail.  A scaled
//          list (05:82-96) reads the scaled copies of those two small tables: stage 05 scales ~1e6 distinct points, not 2.8e8.
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
        __device__ __forceinline__ int2 at(int64_t k) { return p[k]; }
    };
    __device__ __forceinline__ int64_t len(int64_t i) const { return off[i + 1] - off[i]; }
    __device__ __forceinline__ Cur cur(int64_t i) const { return Cur{pts + off[i]}; }
};

struct VGeom {                                // what the points of a layer's walks are made of
    const VPiece* __restrict__ piece;
    const int2* __restrict__ own;             // own points of the kept walks (start + own steps), already scaled for a scaled list
    const int2* __restrict__ lxy;             // pixel of log entry e (index as in VPiece::ent), already scaled for a scaled list
};
struct VSrc {
    const int64_t* __restrict__ off;
    const VView* __restrict__ view;           // nullptr: polyline i = walk i, whole, forward
    const VWalk* __restrict__ walk;
    VGeom g;
    // A point costs the index arithmetic below and ONE 8-byte load: consecutive k of a thread group read consecutive own points or
    // consecutive log entries (modulo the cycle), i.e. coalesced reads of a table that lives in L2.  The tail piece last used is kept in
    // registers; d % lam is a multiply-high with the piece's precomputed floor(2^32 / lam) and one correction.
    struct Cur {
        VGeom g; VWalk w; unsigned first, len, rev;
        unsigned pc_u0, pc_n, pc_ent, pc_lam, pc_magic;      // current piece: tail points [pc_u0, pc_u0 + pc_n)
        __device__ __forceinline__ int2 at(int64_t k) {
            unsigned t = rev ? first + len - 1u - (unsigned)k : first + (unsigned)k;
            if ((w.flags & 1u) && t == w.len - 1u) t = 0u;
            if (t <= w.n_own) return g.own[w.own_off + t];
            const unsigned u = t - w.n_own - 1u;
            if (u - pc_u0 >= pc_n) {                           // (unsigned: also u < pc_u0) another piece: the last one that starts at or before u
                unsigned j = 0u, hi = w.n_piece - 1u;                   // (a walk may hold thousands of pieces, and a fresh cursor starts here)
                while (j < hi) { const unsigned mid = (j + hi + 1u) >> 1; if (g.piece[w.piece_off + mid].u0 <= u) j = mid; else hi = mid - 1u; }
                const VPiece q = g.piece[w.piece_off + j];
                pc_u0 = q.u0; pc_ent = q.ent; pc_lam = q.lam; pc_magic = q.magic;
                pc_n = (j + 1u < w.n_piece ? g.piece[w.piece_off + j + 1u].u0 : 0xffffffffu) - q.u0;
            }
            unsigned d = u - pc_u0;
            if (pc_lam) { const unsigned qq = __umulhi(d, pc_magic); d -= qq * pc_lam; d = d >= pc_lam ? d - pc_lam : d; }
            return g.lxy[pc_ent + d];
        }
    };
    __device__ __forceinline__ int64_t len(int64_t i) const { return off[i + 1] - off[i]; }
    __device__ __forceinline__ Cur cur(int64_t i) const {
        Cur c; c.g = g; c.pc_u0 = 0u; c.pc_n = 0u; c.pc_ent = 0u; c.pc_lam = 0u; c.pc_magic = 0u;
        if (view) { const VView v = view[i]; c.w = walk[v.wid]; c.first = v.first; c.len = v.len; c.rev = v.rev; }
        else { c.w = walk[i]; c.first = 0u; c.len = c.w.len; c.rev = 0u; }
        return c;
    }
};

// the int32 x,y view the serial helpers of vec_serial.h index (xy[2 * i], xy[2 * i + 1]) over a cursor: a small polyline copied to registers / scratch
template <class Cur, int CAP>
struct LocalPts {
    int32_t xy[2 * CAP];
    __device__ __forceinline__ void load(Cur& c, int n) { for (int i = 0; i < n; i++) { const int2 p = c.at(i); xy[2 * i] = p.x; xy[2 * i + 1] = p.y; } }
};
