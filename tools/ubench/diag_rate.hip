// dev tool: ticks of diag_factor_invert / small_spd_solve of lqmpc_wg_linalg.h (wave 0 of a 4-wave workgroup, one workgroup per CU)
#include "../../lq_mpc_amd/csrc/lqmpc_wg_linalg.h"
#include <cstdio>
using namespace lqmpc::wg;
__global__ void __launch_bounds__(256) k(double *out, long long *cyc)
{
    extern __shared__ double raw[];
    ldsd *lds = (ldsd *)raw;
    for (int e = threadIdx.x; e < 20 * BLK; e += 256) { const int r = (e % BLK) / LD, c = (e % BLK) % LD; lds[e] = (r == c) ? 4.0 + 0.01 * (e / BLK) : 0.05 / (1 + r + c); }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool ok = true;
    long long t0 = clock64();
    if (wave == 0) for (int m = 0; m < 8; ++m) ok = diag_factor_invert(lds + m * BLK, lds + (8 + m) * BLK) && ok;
    long long t1 = clock64();
    __syncthreads();
    if (wave == 0) for (int m = 0; m < 8; ++m) ok = small_spd_solve(lds + 16 * BLK, lds + 17 * BLK, lds + 18 * BLK, lds + 19 * BLK, 8 + m) && ok;
    long long t2 = clock64();
    double acc[16], x = lds[threadIdx.x], y = lds[threadIdx.x + 7];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = lds[threadIdx.x + j];
    long long t3 = clock64();
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) fmac_rowb(acc[j], x, y, 3);                 // 256 x (s_nop 1 + v_fmac_f64_dpp), independent
    }
    long long t4 = clock64();
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) fmac_rowb4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3], x, x, x, x, y, 3);   // 256 fmacs, one s_nop per 4
    }
    long long t5 = clock64();
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_fma(rowb(x, j), y, acc[j]);     // 256 x (DPP mov + fma)
    }
    long long t6 = clock64();
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_fma(rdlane(x, j), y, acc[j]);     // 256 x (2 readlane + fma with an SGPR operand)
    }
    long long t7 = clock64();
    x = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] + acc[8] + acc[9] + acc[10] + acc[11] + acc[12] + acc[13] + acc[14] + acc[15];
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t4 - t3; cyc[3] = t5 - t4; cyc[4] = t6 - t5; cyc[5] = t7 - t6; }
    lds[threadIdx.x + 8 * BLK] += x;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x + 8 * BLK] + ok;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 64);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 20 * BLK * 8, 0, o, c);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("ticks per call: diag_factor_invert %.0f | small_spd_solve %.0f ; per op: fmac_rowb %.1f | fmac_rowb4 %.1f | rowb+fma %.1f | rdlane+fma %.1f\n", h[0] / 8.0, h[1] / 8.0, h[2] / 256.0, h[3] / 256.0, h[4] / 256.0, h[5] / 256.0);
    }
    return 0;
}
