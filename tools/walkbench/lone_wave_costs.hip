#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define PAT(name, body) \
__global__ __launch_bounds__(64) void name(unsigned long long* out, int n) { \
    __shared__ int lds[256]; lds[threadIdx.x] = threadIdx.x; __syncthreads(); \
    unsigned base = (unsigned)(uintptr_t)lds; int vv = threadIdx.x, v1 = 0, v2 = base + 4 * (threadIdx.x & 7); \
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 0, s4 = 3, s5 = 0, s6 = 0, s7 = 0; \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < n; i++) { \
        asm volatile(REP64(body) "9:\n\t" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), "+v"(vv), "+v"(v1), "+v"(v2) :: "vcc", "scc", "memory"); } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s0 + s3 + s5 + s6 + s7; } out[2 + threadIdx.x] = vv + v1; }

PAT(p_sadd_dep, "s_add_i32 %0, %0, 1\n\t")
PAT(p_sadd_ind, "s_add_i32 %0, %0, 1\n\ts_add_i32 %3, %3, 1\n\ts_add_i32 %5, %5, 1\n\ts_add_i32 %6, %6, 1\n\t")
PAT(p_cmp_br, "s_cmp_eq_u32 %1, %2\n\ts_cbranch_scc1 9f\n\t")
PAT(p_and_br, "s_and_b32 %3, %0, 0\n\ts_cbranch_scc1 9f\n\t")
PAT(p_readlane_add, "v_readlane_b32 %3, %8, %4\n\ts_add_i32 %5, %5, %3\n\t")
PAT(p_readlane3, "v_readlane_b32 %3, %8, %4\n\tv_readlane_b32 %6, %9, %4\n\tv_readlane_b32 %7, %10, %4\n\t")
PAT(p_vcmp_sand, "v_cmp_gt_i32 vcc, 0, %8\n\ts_and_b32 %3, vcc_lo, %4\n\t")
PAT(p_valu_ind, "v_add_u32 %9, %0, %8\n\t")
PAT(p_valu_dep, "v_add_u32 %9, %9, %8\n\t")
PAT(p_writelane, "v_writelane_b32 %9, %0, m0\n\t")
PAT(p_taken, "s_branch 1f\n\t1:\n\t")
PAT(p_lds_rd, "ds_read_i8 %9, %10\n\ts_waitcnt lgkmcnt(0)\n\t")
PAT(p_lds_wr_rd, "ds_write_b8 %10, %8\n\tds_read_i8 %9, %10\n\ts_waitcnt lgkmcnt(0)\n\t")
PAT(p_lds_rd_add, "ds_read_i8 %9, %10\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %10, %10, %9\n\t")
PAT(p_salu_valu_alt, "s_add_i32 %0, %0, 1\n\tv_add_u32 %9, %0, %8\n\t")
PAT(p_ff1_readlane, "s_ff1_i32_b32 %3, %4\n\tv_readlane_b32 %5, %8, %3\n\ts_add_i32 %4, %4, %5\n\t")
PAT(p_cselect, "s_and_b32 %3, %0, %1\n\ts_cselect_b32 %5, %1, %2\n\ts_cselect_b32 %6, -1, 0\n\t")

// cross-lane and wide LDS patterns (the chains of csrc/vector08.hip k_cumlen_*)
#define PATX(name, body) \
__global__ __launch_bounds__(64) void name(unsigned long long* out, int n) { \
    __shared__ __align__(16) int lds[1024]; for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i; __syncthreads(); \
    unsigned base = (unsigned)(uintptr_t)lds; int vv = threadIdx.x, v1 = 0, v2 = base; \
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 0, s4 = 3, s5 = 0, s6 = 0, s7 = 0; \
    if (n < 0) asm volatile("s_mov_b64 exec, 1"); \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < (n < 0 ? -n : n); i++) { \
        asm volatile(REP64(body) "9:\n\t" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), "+v"(vv), "+v"(v1), "+v"(v2) :: "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); } \
    asm volatile("s_waitcnt lgkmcnt(0)"); \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_mov_b64 exec, -1"); \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s0 + s3 + s5 + s6 + s7; } out[2 + threadIdx.x] = vv + v1; }
PATX(x_dpp_row_dep, "v_add_f32_dpp %9, %9, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t")
PATX(x_dpp_wave_dep, "v_add_f32_dpp %9, %9, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t")
PATX(x_dpp_row_ind, "v_add_f32_dpp %9, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
PATX(x_vadd_dep4, "v_add_f32 %9, %9, %8\n\tv_add_f32 %9, %9, %8\n\tv_add_f32 %9, %9, %8\n\tv_add_f32 %9, %9, %8\n\t")
PATX(x_lds_r128, "ds_read_b128 v[100:103], %10\n\t")
PATX(x_lds_w128, "ds_write_b128 %10, v[100:103]\n\t")
PATX(x_lds_r32, "ds_read_b32 v100, %10\n\t")
PATX(x_lds_w32, "ds_write_b32 %10, v100\n\t")
PATX(x_chain_lds, "ds_read_b128 v[100:103], %10\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %9, %9, v100\n\tv_add_f32 %9, %9, v101\n\tv_add_f32 %9, %9, v102\n\tv_add_f32 %9, %9, v103\n\tds_write_b128 %10, v[104:107] offset:2048\n\t")
PATX(x_chain_lds_nowait, "ds_read_b128 v[100:103], %10\n\tv_add_f32 %9, %9, v100\n\tv_add_f32 %9, %9, v101\n\tv_add_f32 %9, %9, v102\n\tv_add_f32 %9, %9, v103\n\tds_write_b128 %10, v[104:107] offset:2048\n\t")
PATX(x_readlane_vadd, "v_readlane_b32 %3, %8, %4\n\tv_add_f32 %9, %9, %3\n\t")
PATX(x_bpermute_dep, "ds_bpermute_b32 %9, %10, %9\n\ts_waitcnt lgkmcnt(0)\n\t")

#define RUNX(name, per, nn) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, d, nn); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-22s %s %8.2f cycles per pattern (%d instr) -> %.2f per instr\n", #name, nn < 0 ? "[1 lane] " : "[64 lanes]", (double)h[0] / (2000.0 * 64), per, (double)h[0] / (2000.0 * 64 * per)); }
#define RUN(name, per) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, d, 2000); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-18s %8.2f cycles per pattern (%d instr) -> %.2f per instr\n", #name, (double)h[0] / (2000.0 * 64), per, (double)h[0] / (2000.0 * 64 * per)); }
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 4096); unsigned long long h[2];
    for (int r = 0; r < 2; r++) {
    RUN(p_sadd_dep, 1) RUN(p_sadd_ind, 4) RUN(p_cmp_br, 2) RUN(p_and_br, 2) RUN(p_readlane_add, 2) RUN(p_readlane3, 3) RUN(p_vcmp_sand, 2) RUN(p_valu_ind, 1) RUN(p_valu_dep, 1)
    RUN(p_writelane, 1) RUN(p_taken, 1) RUN(p_lds_rd, 2) RUN(p_lds_wr_rd, 3) RUN(p_lds_rd_add, 3) RUN(p_salu_valu_alt, 2) RUN(p_ff1_readlane, 3) RUN(p_cselect, 3)
    RUNX(x_dpp_row_dep, 2, 2000) RUNX(x_dpp_wave_dep, 2, 2000) RUNX(x_dpp_row_ind, 1, 2000) RUNX(x_vadd_dep4, 4, 2000) RUNX(x_readlane_vadd, 2, 2000) RUNX(x_bpermute_dep, 2, 2000)
    RUNX(x_lds_r128, 1, 2000) RUNX(x_lds_r128, 1, -2000) RUNX(x_lds_w128, 1, 2000) RUNX(x_lds_w128, 1, -2000) RUNX(x_lds_r32, 1, 2000) RUNX(x_lds_r32, 1, -2000) RUNX(x_lds_w32, 1, 2000) RUNX(x_lds_w32, 1, -2000)
    RUNX(x_chain_lds, 7, 2000) RUNX(x_chain_lds, 7, -2000) RUNX(x_chain_lds_nowait, 6, 2000) RUNX(x_chain_lds_nowait, 6, -2000)
    }
    return 0;
}
