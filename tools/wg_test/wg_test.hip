// standalone check of lqmpc_wg_linalg.h: blocked MFMA Cholesky / solve / inverse vs numpy (dev tool)
__device__ long long g_zt[8];
#define LQMPC_ZTZ_T(k) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_zt[k] = clock64(); } while (0)
#include "../../lq_mpc_amd/csrc/lqmpc_wg_linalg.h"
#include <cstdio>
using namespace lqmpc;
using namespace lqmpc::wg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)
__device__ long long g_cyc[12];
__global__ void __launch_bounds__(256) k_test(const double *Kin, const double *bin, double *Lout, double *xout, double *Wout, double *sout, int nb, int msmall, int *okout, double *W2out)
{
    extern __shared__ double lds_raw[];
    ldsd *lds = (ldsd *)lds_raw;
    const int nblk = nb * (nb + 1) / 2;
    unsigned base = (unsigned)(size_t)lds; asm volatile("" : "+s"(base)); lds = (ldsd *)(size_t)base;   // opaque: the helpers must take K as an argument, as in the product kernel (no dynamic-LDS table look-ups)
    ldsd *K = lds, *Linv = K + nblk * BLK, *b = Linv + nb * BLK, *T = b + nb * BS, *S = T + BLK, *sv = S + BLK;
    ldsi *flag = (ldsi *)(sv + BS);
    ldsd *Sc = sv + BS + 2;
    const long long inst = blockIdx.x;
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) K[e] = Kin[inst * nblk * BLK + e];
    for (int e = threadIdx.x; e < nb * BS; e += 256) b[e] = bin[inst * nb * BS + e];
    // small system: leading msmall x msmall of block (0,0), identity outside; rhs = b[0..m)
    for (int e = threadIdx.x; e < BLK; e += 256) { const int r = e / LD, c = e % LD; S[e] = (r < msmall && c < msmall) ? Kin[inst * nblk * BLK + e] : (r == c ? 1.0 : 0.0); }
    __syncthreads();
    long long t0 = clock64();
    bool oks = true;
    if (threadIdx.x < 64) oks = small_spd_solve(S, T, b, sv, msmall);
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x < BS) sout[inst * BS + threadIdx.x] = sv[threadIdx.x];
    long long t2 = clock64();
    const bool ok = chol_blocked(K, Linv, nb, flag);
    long long t3 = clock64();
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) Lout[inst * nblk * BLK + e] = K[e];
    long long t4 = clock64();
    solve_blocked(K, Linv, nb, b);
    long long t5 = clock64();
    tri_invert_blocked(K, Linv, nb);
    long long t6 = clock64();
    ztz_blocked(K, nb);
    long long t7 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { g_cyc[0] = t1 - t0; g_cyc[1] = t3 - t2; g_cyc[2] = t5 - t4; g_cyc[3] = t6 - t5; g_cyc[4] = t7 - t6; g_cyc[5] = g_zt[0] - t6; g_cyc[6] = g_zt[1] - g_zt[0]; g_cyc[7] = g_zt[3] - g_zt[1]; }
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) Wout[inst * nblk * BLK + e] = K[e];
    // the fused route: Cholesky with the factor's inverse riding along, then Z'Z
    __syncthreads();
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) K[e] = Kin[inst * nblk * BLK + e];
    __syncthreads();
    long long t8 = clock64();
    const bool ok2 = chol_inverse_blocked(K, Linv, nb, flag, Sc);
    long long t9 = clock64();
    ztz_blocked(K, nb);
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) g_cyc[8] = t9 - t8;
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) W2out[inst * nblk * BLK + e] = ok2 ? K[e] : 0.0;
    for (int e = threadIdx.x; e < nb * BS; e += 256) xout[inst * nb * BS + e] = b[e];
    if (threadIdx.x == 0) okout[inst] = (ok && oks) ? 1 : 0;
}
__global__ void k_rsq(const double *x, double *seed, double *one, double *full, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    seed[i] = y;
    double e = __builtin_fma(-v * y, y, 1.0);
    y = __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
    one[i] = y;
    full[i] = frsqrt(v);
}
extern "C" int run_rsq(const double *x, double *seed, double *one, double *full, int n)
{
    double *dx, *ds, *d1, *df;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&ds, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&df, n * 8));
    CK(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rsq, dim3((n + 255) / 256), dim3(256), 0, 0, dx, ds, d1, df, n);
    CK(hipMemcpy(seed, ds, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(one, d1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(full, df, n * 8, hipMemcpyDeviceToHost));
    CK(hipFree(dx)); CK(hipFree(ds)); CK(hipFree(d1)); CK(hipFree(df));
    return 0;
}
extern "C" int run_test(const double *Kin, const double *bin, double *Lout, double *xout, double *Wout, double *sout, int nb, int msmall, int ninst, int *okout, float *ms, long long *cyc, double *W2out)
{
    const int nblk = nb * (nb + 1) / 2;
    size_t kb = (size_t)ninst * nblk * BLK * 8, vb = (size_t)ninst * nb * BS * 8;
    double *dK, *db, *dL, *dx, *dW, *ds, *dW2; int *dok;
    CK(hipMalloc(&dK, kb)); CK(hipMalloc(&db, vb)); CK(hipMalloc(&dL, kb)); CK(hipMalloc(&dW, kb)); CK(hipMalloc(&dW2, kb)); CK(hipMalloc(&dx, vb)); CK(hipMalloc(&ds, ninst * BS * 8)); CK(hipMalloc(&dok, ninst * 4));
    CK(hipMemcpy(dK, Kin, kb, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bin, vb, hipMemcpyHostToDevice));
    size_t lds = (size_t)(nblk * BLK + nb * BLK + nb * BS + 2 * BLK + BS + 2 + nb * BLK) * 8 + 64;
    CK(hipFuncSetAttribute((const void *)k_test, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_test, dim3(ninst), dim3(256), lds, 0, dK, db, dL, dx, dW, ds, nb, msmall, dok, dW2);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_test, dim3(ninst), dim3(256), lds, 0, dK, db, dL, dx, dW, ds, nb, msmall, dok, dW2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(ms, e0, e1));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(Lout, dL, kb, hipMemcpyDeviceToHost)); CK(hipMemcpy(Wout, dW, kb, hipMemcpyDeviceToHost)); CK(hipMemcpy(W2out, dW2, kb, hipMemcpyDeviceToHost)); CK(hipMemcpy(xout, dx, vb, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sout, ds, ninst * BS * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(okout, dok, ninst * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), 12 * sizeof(long long)));
    CK(hipFree(dK)); CK(hipFree(db)); CK(hipFree(dL)); CK(hipFree(dW)); CK(hipFree(dW2)); CK(hipFree(dx)); CK(hipFree(ds)); CK(hipFree(dok));
    return 0;
}
