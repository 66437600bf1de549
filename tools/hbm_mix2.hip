// Which property of k_preprocess_backward's memory pattern costs its bandwidth (round 5)?  One thread per "splat":
//   reads 12 float4 (coalesced over the workgroup's block, non-temporal), writes 12 float4 of "SH gradient" (coalesced,
//   non-temporal) + 56 bytes of small gradients.  Variants switch ONE property at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4* p) { v4 x = __builtin_nontemporal_load((const v4*)p); return make_float4(x.x, x.y, x.z, x.w); }
__device__ __forceinline__ void stnt(float4* p, float4 v) { v4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, (v4*)p); }
// MODE bits: 1 = small outputs as strided 4-byte stores ([N,3] x 3, [N,4] as float4, [N] x 1) instead of coalesced float4
//            2 = the 192-byte output goes through an LDS staging area of 13 float4 per thread with a barrier (53 KB: 3 WG per CU)
//            4 = 24 bytes of the input as six strided 4-byte loads
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ in, const float* __restrict__ in3, float4* __restrict__ out, float* __restrict__ small, int n) {
    __shared__ float4 s[(MODE & 2) ? 256 * 13 : 1];
    const int t = threadIdx.x;
    const size_t blk = blockIdx.x;
    float4 v[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) v[r] = ldnt(in + blk * 3072 + r * 256 + t);
    float3 a = make_float3(0.f, 0.f, 0.f), b = a;
    if (MODE & 4) {
        const size_t i = blk * 256 + t;
        a = make_float3(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]);
        b = make_float3(in3[3 * (size_t)n + 3 * i], in3[3 * (size_t)n + 3 * i + 1], in3[3 * (size_t)n + 3 * i + 2]);
    }
    float4 acc = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, 0.f);
#pragma unroll
    for (int r = 0; r < 12; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
    if (MODE & 8) {   // the 192-byte row written by its own thread: twelve 16-byte stores at a 192-byte stride (no LDS)
        float4* row = out + (blk * 256 + t) * 12;
#pragma unroll
        for (int r = 0; r < 12; ++r) { if (MODE & 16) stnt(row + r, make_float4(acc.x + r, acc.y, acc.z, acc.w)); else row[r] = make_float4(acc.x + r, acc.y, acc.z, acc.w); }
    } else if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < 12; ++r) s[t * 13 + r] = make_float4(acc.x + r, acc.y, acc.z, acc.w);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 12; ++r) { const int i = r * 256 + t; const int sp = i / 12; stnt(out + blk * 3072 + i, s[sp * 13 + (i - sp * 12)]); }
    } else {
#pragma unroll
        for (int r = 0; r < 12; ++r) stnt(out + blk * 3072 + r * 256 + t, make_float4(acc.x + r, acc.y, acc.z, acc.w));
    }
    const size_t i = blk * 256 + t;
    if (MODE & 1) {
        float* m3 = small; float* m2 = small + 3 * (size_t)n; float* sc = small + 6 * (size_t)n; float4* rot = (float4*)(small + 9 * (size_t)n); float* op = small + 13 * (size_t)n;
        m3[3 * i] = acc.x; m3[3 * i + 1] = acc.y; m3[3 * i + 2] = acc.z;
        m2[3 * i] = acc.y; m2[3 * i + 1] = acc.x; m2[3 * i + 2] = 0.f;
        sc[3 * i] = acc.z; sc[3 * i + 1] = acc.y; sc[3 * i + 2] = acc.x;
        rot[i] = acc; op[i] = acc.w;
    } else {   // the same 56 bytes per thread as 3.5 coalesced float4 (4 here: 64 bytes)
        float4* o = (float4*)small;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[blk * 1024 + r * 256 + t] = acc;
    }
}
template <int MODE> void run(const float4* in, const float* in3, float4* out, float* small, int n, const char* name) {
    const int blocks = n / 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(in, in3, out, small, n);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); k<MODE><<<blocks, 256>>>(in, in3, out, small, n); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double bytes = (double)n * (192.0 + 192.0 + ((MODE & 1) ? 56.0 : 64.0) + ((MODE & 4) ? 24.0 : 0.0));
    printf("%-60s %7.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
}
int main() {
    const int n = 1 << 20;
    float4 *in, *out; float *small, *in3;
    hipMalloc(&in, (size_t)n * 192); hipMalloc(&out, (size_t)n * 192); hipMalloc(&small, (size_t)n * 64); hipMalloc(&in3, (size_t)n * 24);
    hipMemset(in, 0, (size_t)n * 192); hipMemset(in3, 0, (size_t)n * 24);
    run<0>(in, in3, out, small, n, "coalesced everything");
    run<1>(in, in3, out, small, n, "+ small outputs strided (12-byte stride x 9, float4, 4 B)");
    run<2>(in, in3, out, small, n, "+ 192-byte output through 53 KB of LDS + barrier");
    run<4>(in, in3, out, small, n, "+ 24 input bytes as six strided loads");
    run<3>(in, in3, out, small, n, "strided outputs AND LDS staging");
    run<7>(in, in3, out, small, n, "all three (k_preprocess_backward's shape)");
    run<8>(in, in3, out, small, n, "192-byte rows stored by their own thread (no LDS), temporal");
    run<24>(in, in3, out, small, n, "192-byte rows stored by their own thread (no LDS), non-temporal");
    run<13>(in, in3, out, small, n, "rows by own thread + strided small outputs + strided inputs");
    return 0;
}
