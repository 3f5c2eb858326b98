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

#define RUN(name, per) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, d, 2000); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-18s %8.2f cycles per pattern (%d instr) -> %.2f per instr\n", #name, (double)h[0] / (2000.0 * 64), per, (double)h[0] / (2000.0 * 64 * per)); }
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 4096); unsigned long long h[2];
    for (int r = 0; r < 2; r++) {
    RUN(p_sadd_dep, 1) RUN(p_sadd_ind, 4) RUN(p_cmp_br, 2) RUN(p_and_br, 2) RUN(p_readlane_add, 2) RUN(p_readlane3, 3) RUN(p_vcmp_sand, 2) RUN(p_valu_ind, 1) RUN(p_valu_dep, 1)
    RUN(p_writelane, 1) RUN(p_taken, 1) RUN(p_lds_rd, 2) RUN(p_lds_wr_rd, 3) RUN(p_lds_rd_add, 3) RUN(p_salu_valu_alt, 2) RUN(p_ff1_readlane, 3) RUN(p_cselect, 3)
    }
    return 0;
}
