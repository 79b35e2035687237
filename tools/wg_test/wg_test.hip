// standalone check of lqmpc_wg_linalg.h: blocked MFMA Cholesky + solve vs numpy (dev tool)
#include "../../lq_mpc_amd/csrc/lqmpc_wg_linalg.h"
#include <cstdio>
using namespace lqmpc;
using namespace lqmpc::wg;
__global__ void __launch_bounds__(256) k_test(const double *Kin, const double *bin, double *Lout, double *xout, int nb, int *okout)
{
    extern __shared__ double lds[];
    const int nblk = nb * (nb + 1) / 2;
    double *K = lds, *Linv = K + nblk * BLK, *b = Linv + nb * BLK;
    int *flag = (int *)(b + nb * BS);
    const long long inst = blockIdx.x;
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) K[e] = Kin[inst * nblk * BLK + e];
    for (int e = threadIdx.x; e < nb * BS; e += 256) b[e] = bin[inst * nb * BS + e];
    __syncthreads();
    const bool ok = chol_blocked(K, Linv, nb, flag);
    solve_blocked(K, Linv, nb, b);
    for (int e = threadIdx.x; e < nblk * BLK; e += 256) Lout[inst * nblk * BLK + e] = K[e];
    for (int e = threadIdx.x; e < nb * BS; e += 256) xout[inst * nb * BS + e] = b[e];
    if (threadIdx.x == 0) okout[inst] = ok ? 1 : 0;
}
extern "C" int run_test(const double *Kin, const double *bin, double *Lout, double *xout, int nb, int ninst, int *okout, float *ms)
{
    const int nblk = nb * (nb + 1) / 2;
    size_t kb = (size_t)ninst * nblk * BLK * 8, vb = (size_t)ninst * nb * BS * 8;
    double *dK, *db, *dL, *dx; int *dok;
    hipMalloc(&dK, kb); hipMalloc(&db, vb); hipMalloc(&dL, kb); hipMalloc(&dx, vb); hipMalloc(&dok, ninst * 4);
    hipMemcpy(dK, Kin, kb, hipMemcpyHostToDevice); hipMemcpy(db, bin, vb, hipMemcpyHostToDevice);
    size_t lds = (size_t)(nblk * BLK + nb * BLK + nb * BS) * 8 + 64;
    hipFuncSetAttribute((const void *)k_test, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_test, dim3(ninst), dim3(256), lds, 0, dK, db, dL, dx, nb, dok);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_test, dim3(ninst), dim3(256), lds, 0, dK, db, dL, dx, nb, dok);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(ms, e0, e1);
    hipError_t err = hipDeviceSynchronize();
    hipMemcpy(Lout, dL, kb, hipMemcpyDeviceToHost); hipMemcpy(xout, dx, vb, hipMemcpyDeviceToHost); hipMemcpy(okout, dok, ninst * 4, hipMemcpyDeviceToHost);
    hipFree(dK); hipFree(db); hipFree(dL); hipFree(dx); hipFree(dok);
    return (int)err;
}
