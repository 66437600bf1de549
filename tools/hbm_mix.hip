// What a streaming kernel with the byte mix of the per-splat kernels can reach on MI355X (round 5): N threads, each reads R
// float4 (coalesced: consecutive lanes, consecutive 16-byte pieces of a contiguous per-workgroup block) and writes W float4 the
// same way.  Reported: GB/s of bytes moved.  k_preprocess moves 240 B in / 120 B out per splat, k_preprocess_backward
// ~200 in / 248 out; a copy is R = W.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    const size_t base_in = (size_t)blockIdx.x * 256 * R, base_out = (size_t)blockIdx.x * 256 * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4* p = in + base_in + r * 256 + threadIdx.x;
        if (NT) { typedef float v4 __attribute__((ext_vector_type(4))); v4 x = __builtin_nontemporal_load((const v4*)p); v[r] = make_float4(x.x, x.y, x.z, x.w); }
        else v[r] = *p;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
#pragma unroll
    for (int w = 0; w < W; ++w) out[base_out + w * 256 + threadIdx.x] = make_float4(acc.x + w, acc.y, acc.z, acc.w);
    if (W == 0 && acc.x == 12345.678f) out[0] = acc;
}
template <int R, int W, bool NT> void run(const float4* in, float4* out, int n, const char* name) {
    const int blocks = n / 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<R, W, NT><<<blocks, 256>>>(in, out, n);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); k<R, W, NT><<<blocks, 256>>>(in, out, n); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double bytes = (double)n * 16.0 * (R + W);
    printf("%-34s %7.1f us  %6.2f TB/s  (%d B in, %d B out per thread, %d threads)\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, 16 * R, 16 * W, n);
}
int main() {
    const int n = 1 << 20;
    float4 *in, *out;
    hipMalloc(&in, (size_t)n * 16 * 16); hipMalloc(&out, (size_t)n * 16 * 16);
    hipMemset(in, 0, (size_t)n * 16 * 16); hipMemset(out, 0, (size_t)n * 16 * 16);
    run<16, 0, false>(in, out, n, "read 256 B");
    run<16, 0, true>(in, out, n, "read 256 B non-temporal");
    run<0, 16, false>(in, out, n, "write 256 B");
    run<8, 8, false>(in, out, n, "copy 128 B");
    run<15, 8, false>(in, out, n, "240 in / 128 out (k_preprocess)");
    run<15, 8, true>(in, out, n, "240 in nt / 128 out");
    run<12, 16, false>(in, out, n, "192 in / 256 out (k_preprocess_bwd)");
    run<12, 16, true>(in, out, n, "192 in nt / 256 out");
    return 0;
}
