// Do VALU and fp32 MFMA instructions of the waves of one SIMD overlap on gfx950?  (replay loop of the backward blend)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NM>   // NV independent fma chains steps and NM mfma per group; 8 groups per iteration
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0;
    float x = a * threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(g + j) & 7] = __builtin_fmaf(v[(g + j) & 7], a, b);
            if (NM >= 1) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, d0, 0, 0, 0);
            if (NM >= 2) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, d1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
}

template <int NV, int NM>
float run(int wgs, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NM>), dim3(wgs), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 4 * 256 * 4096);
    const int iters = 2000;
    for (int wpc = 1; wpc <= 4; ++wpc) {   // workgroups per CU = waves per SIMD
        const int wgs = 256 * wpc;
        float t_v16 = run<16, 0>(wgs, iters, out), t_m1 = run<0, 1>(wgs, iters, out), t_m2 = run<0, 2>(wgs, iters, out);
        float t_b1 = run<16, 1>(wgs, iters, out), t_b2 = run<16, 2>(wgs, iters, out), t_v8 = run<8, 0>(wgs, iters, out), t_b82 = run<8, 2>(wgs, iters, out);
        // cycles per group per SIMD at 2.4 GHz nominal: time * 2.4e6 / (iters * 8) [per wave] ; waves per SIMD = wpc
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (iters * 8.0); };
        printf("waves/SIMD %d: cycles per group (all waves of a SIMD together): 16 fma %.0f | 1 mfma %.0f | 2 mfma %.0f | 16 fma + 1 mfma %.0f | 16 fma + 2 mfma %.0f | 8 fma %.0f | 8 fma + 2 mfma %.0f\n",
               wpc, cyc(t_v16), cyc(t_m1), cyc(t_m2), cyc(t_b1), cyc(t_b2), cyc(t_v8), cyc(t_b82));
    }
    return 0;
}
