// Dev microbenchmark (GPU box): operand layout and issue cost of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products, one per
// 16-lane group).  hipcc -O3 --offload-arch=gfx950 -o mfma4_layout mfma4_layout.hip && ./mfma4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void k_layout(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int DEP>
__global__ void k_rate(double *out, int iters, long long *ticks)
{
    const int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (DEP == 1) {            // the result feeds the next product's B operand
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0 + b, 0.0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, 0.0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, 0.0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, 0.0, 0, 0, 0);
        } else if (DEP == 2) {     // accumulator chain
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        } else {                   // four independent accumulators
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = c0 + c1 + c2 + c3;
    if (l == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
int main()
{
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    srand(1);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX; hb[i] = rand() / (double)RAND_MAX; }
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // hypotheses: the lane is three 2-bit fields f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4; each operand maps (row, col, block) to a
    // permutation of them.  All 6^3 combinations are tried.
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    auto lane_of = [&](const int *pm, int x, int y, int z) { int f[3]; f[pm[0]] = x; f[pm[1]] = y; f[pm[2]] = z; return f[0] + 4 * f[1] + 16 * f[2]; };
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        double err = 0;
        for (int g = 0; g < 4; ++g) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += ha[lane_of(perm[pa], i, k, g)] * hb[lane_of(perm[pb], k, j, g)];
            err = fmax(err, fabs(s - hd[lane_of(perm[pd], i, j, g)]));
        }
        if (err < 1e-12)
            printf("MATCH: A[i][k] of block g: fields (i,k,g) -> bit pairs (%d,%d,%d); B[k][j]: (k,j,g) -> (%d,%d,%d); D[i][j]: (i,j,g) -> (%d,%d,%d)  (bit pair 0 = lane&3, 1 = (lane>>2)&3, 2 = lane>>4)\n",
                   perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2]);
    }
    double *dout; long long *dt, ht;
    hipMalloc(&dout, 8 * 64 * 4096); hipMalloc(&dt, 8);
    const int iters = 10000;
    for (int waves = 1; waves <= 2; ++waves) {
        // one workgroup of 64*waves*4 threads = `waves` wavefronts per SIMD on one CU
        hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(256 * waves), 0, 0, dout, iters, dt); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        printf("waves/SIMD %d: dependent through B operand: %.1f ticks per MFMA\n", waves, ht / (4.0 * iters));
        hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(256 * waves), 0, 0, dout, iters, dt); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        printf("waves/SIMD %d: accumulator chain:            %.1f ticks per MFMA\n", waves, ht / (4.0 * iters));
        hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(256 * waves), 0, 0, dout, iters, dt); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        printf("waves/SIMD %d: four independent accumulators: %.1f ticks per MFMA\n", waves, ht / (4.0 * iters));
    }
    return 0;
}
