// dev tool: issue cost of v_mfma_f64_16x16x4_f64 / v_fma_f64 / ds_read_b64 / s_barrier on gfx950 in clock64 ticks (one workgroup per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double *out, long long *cyc, int reps)
{
    __shared__ double lds[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = 1.0 / (1 + e);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double a = lds[lane], b = lds[lane + 64];
    d4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }                  // dependent chain
    long long t1 = clock64();
    for (int r = 0; r < reps; ++r) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t2 = clock64();
    double f0 = a, f1 = b, f2 = a + 1, f3 = b + 1;
    for (int r = 0; r < reps; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b); }
    }
    long long t3 = clock64();
    double g = a;
    for (int r = 0; r < reps; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) g = __builtin_fma(g, a, b);
    }                                                        // dependent fma
    long long t4 = clock64();
    int idx = lane;
    for (int r = 0; r < reps; ++r) { idx = (int)lds[idx & 4095] + lane; }                                                // dependent LDS read (+cvt)
    long long t5 = clock64();
    for (int r = 0; r < reps; ++r) __syncthreads();
    long long t6 = clock64();
    double h = a;
    for (int r = 0; r < reps; ++r) { h = __builtin_amdgcn_rsq(h + 1.5); }
    long long t7 = clock64();
    double r0 = a + 2, r1 = b + 2, r2 = a + 3, r3 = b + 3, q0 = a, q1 = b, q2 = a, q3 = b;
    for (int r = 0; r < reps; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { r0 = __builtin_amdgcn_rsq(r0); r1 = __builtin_amdgcn_rsq(r1); r2 = __builtin_amdgcn_rsq(r2); r3 = __builtin_amdgcn_rsq(r3); }
    }
    long long t8 = clock64();
    for (int r = 0; r < reps; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { r0 = __builtin_amdgcn_rsq(r0); q0 = __builtin_fma(q0, a, b); q1 = __builtin_fma(q1, a, b); q2 = __builtin_fma(q2, a, b); q3 = __builtin_fma(q3, a, b); }
    }
    long long t9 = clock64();
    h += r0 + r1 + r2 + r3 + q0 + q1 + q2 + q3;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[7] = t8 - t7; cyc[8] = t9 - t8; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; cyc[6] = t7 - t6; }
    out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + f0 + f1 + f2 + f3 + g + idx + h;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 256 * 256 * 8); hipMalloc(&c, 128);
    const int reps = 1000;
    for (int it = 0; it < 2; ++it) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, o, c, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
        long long tot = 0; for (int i = 0; i < 7; ++i) tot += h[i];
        printf("4 indep rsq %.1f | dep rsq + 4 indep fma %.1f || ", h[7] / (double)reps, h[8] / (double)reps); printf("ticks per: dep mfma %.1f | 4 indep mfma %.1f | 4 indep fma %.1f | dep fma %.1f | dep lds read %.1f | barrier %.1f | dep rsq+add %.1f ; total ticks %lld in %.3f ms -> %.2f ticks/ns\n",
               h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps, h[3] / (double)reps, h[4] / (double)reps, h[5] / (double)reps, h[6] / (double)reps, tot, ms, tot / (ms * 1e6));
    }
    return 0;
}
