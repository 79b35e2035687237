// dev tool: ticks per 16x16x16 f64 block product through block_mm / block_sum of lqmpc_wg_linalg.h (4 waves, one workgroup per CU)
#include "../../lq_mpc_amd/csrc/lqmpc_wg_linalg.h"
#include <cstdio>
using namespace lqmpc::wg;
__global__ void __launch_bounds__(256) k(double *out, long long *cyc, int len)
{
    extern __shared__ double raw[];
    ldsd *lds = (ldsd *)raw;
    for (int e = threadIdx.x; e < 40 * BLK; e += 256) lds[e] = 1.0 / (1 + e);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d4_t c = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int m = 0; m < len; ++m) c = block_mm<false, true>(lds + (wave * 8 + (m & 7)) * BLK, lds + (32 + (m & 7)) * BLK, c, true);
    long long t1 = clock64();
    c = block_sum<false, true>(0, len, [&](int m) { return lds + (wave * 8 + (m & 7)) * BLK; }, [&](int m) { return lds + (32 + (m & 7)) * BLK; }, c, true);
    long long t2 = clock64();
    c = block_sum<true, true>(0, len, [&](int m) { return lds + (wave * 8 + (m & 7)) * BLK; }, [&](int m) { return lds + (32 + (m & 7)) * BLK; }, c, false);
    long long t3 = clock64();
    if (wave == 0) c = block_sum<true, true>(0, len, [&](int m) { return lds + (wave * 8 + (m & 7)) * BLK; }, [&](int m) { return lds + (32 + (m & 7)) * BLK; }, c, false);
    long long t4 = clock64();
    for (int m = 0; m < len; ++m) { tile_store(lds + (wave * 8 + (m & 7)) * BLK, c); c = tile_load(lds + (wave * 8 + ((m + 1) & 7)) * BLK); }
    long long t5 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
    tile_store(lds, c);
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 256 * 256 * 8); hipMalloc(&c, 64);
    for (int len : {1, 2, 4, 16, 64}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 40 * BLK * 8, 0, o, c, len);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("len %d total ticks: block_mm loop %lld | sum NT %lld | sum TT %lld | TT one wave %lld | store+load %lld ; per product:", len, h[0], h[1], h[2], h[3], h[4]); printf(" block_mm loop %.1f | block_sum NT %.1f | block_sum TT %.1f | TT, one wave only %.1f | tile store+load %.1f\n",
               h[0] / (double)len, h[1] / (double)len, h[2] / (double)len, h[3] / (double)len, h[4] / (double)len);
    }
    return 0;
}
