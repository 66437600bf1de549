// VALU issue-cost microbenchmark on gfx950: each kernel runs a long dependent-free stream of one instruction kind
// in 8 waves per SIMD (2048 threads per CU x 256 CUs) and reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITERS 2000
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n) {
    float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long m = 0;
    // KIND 12..15: the v_fma stream under a partial EXEC mask (does the SIMD skip passes whose lanes are all disabled?)
    if (KIND == 12) asm volatile("s_mov_b32 exec_lo, -1\n s_mov_b32 exec_hi, 0");
    if (KIND == 13) asm volatile("s_mov_b32 exec_lo, 0xffff\n s_mov_b32 exec_hi, 0");
    if (KIND == 14) asm volatile("s_mov_b32 exec_lo, 0xffff\n s_mov_b32 exec_hi, 0xffff");
    if (KIND == 15) asm volatile("s_mov_b32 exec_lo, 0xff\n s_mov_b32 exec_hi, 0");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0 || KIND >= 12) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
            if (KIND == 1) { asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
            if (KIND == 2) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 3) { asm volatile("v_cmp_lt_f32 %8, %0, %1\n v_cmp_lt_f32 %8, %1, %2\n v_cmp_lt_f32 %8, %2, %3\n v_cmp_lt_f32 %8, %3, %4\n v_cmp_lt_f32 %8, %4, %5\n v_cmp_lt_f32 %8, %5, %6\n v_cmp_lt_f32 %8, %6, %7\n v_cmp_lt_f32 %8, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(m)); }
            if (KIND == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(0x5555555555555555ull)); }
            if (KIND == 5) { asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 6) { asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 7) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4), "v"(*(double*)&a6)); }
            if (KIND == 8) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 9) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 10) { asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((int)((threadIdx.x ^ 5) & 63) << 2)); }
            if (KIND == 16) { asm volatile("v_cmp_lt_u64 %4, %0, %1\n v_cmp_lt_u64 %4, %1, %2\n v_cmp_lt_u64 %4, %2, %3\n v_cmp_lt_u64 %4, %3, %0\n v_cmp_lt_u64 %4, %0, %2\n v_cmp_lt_u64 %4, %1, %3\n v_cmp_lt_u64 %4, %2, %0\n v_cmp_lt_u64 %4, %3, %1" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6), "=s"(m)); }
            if (KIND == 17) { asm volatile("v_cmp_lt_u32 %8, %0, %1\n v_cmp_lt_u32 %8, %1, %2\n v_cmp_lt_u32 %8, %2, %3\n v_cmp_lt_u32 %8, %3, %4\n v_cmp_lt_u32 %8, %4, %5\n v_cmp_lt_u32 %8, %5, %6\n v_cmp_lt_u32 %8, %6, %7\n v_cmp_lt_u32 %8, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(m)); }
            if (KIND == 18) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (KIND == 11) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc"); }
        }
    }
    if (KIND >= 12) asm volatile("s_mov_b64 exec, -1");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m;
}
template <int KIND> double run(float* d, const char* name) {
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, ITERS); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)blocks * 4 /*waves*/ * ITERS * REP / (256.0 * 4);
    const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
    printf("%-22s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, cyc);
    return cyc;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_fma_f32"); run<1>(d, "v_mul_f32"); run<2>(d, "v_exp_f32"); run<3>(d, "v_cmp_lt_f32 -> sgpr"); run<11>(d, "v_cmp_lt_f32 -> vcc");
    run<4>(d, "v_cndmask_b32 (sgpr)"); run<5>(d, "v_add_f32_dpp"); run<6>(d, "v_permlane32_swap"); run<7>(d, "v_pk_fma_f32"); run<8>(d, "v_rcp_f32");
    run<9>(d, "v_mov_b32"); run<10>(d, "ds_bpermute_b32");
    run<16>(d, "v_cmp_lt_u64 -> sgpr"); run<17>(d, "v_cmp_lt_u32 -> sgpr"); run<18>(d, "v_mov_b32_dpp");
    run<12>(d, "v_fma exec=lanes 0-31"); run<13>(d, "v_fma exec=lanes 0-15"); run<14>(d, "v_fma exec=rows 0,2"); run<15>(d, "v_fma exec=lanes 0-7");
    return 0;
}
