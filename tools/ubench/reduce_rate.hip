// dev tool: ticks of workgroup reductions (256 threads): shfl-based vs DPP-based wave reduction, __syncthreads_or vs ballot + one barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double ror(double x, int n)
{
    long long v = __double_as_longlong(x);
    switch (n) {
    case 1: v = __builtin_amdgcn_update_dpp(0ll, v, 0x121, 0xF, 0xF, true); break;
    case 2: v = __builtin_amdgcn_update_dpp(0ll, v, 0x122, 0xF, 0xF, true); break;
    case 4: v = __builtin_amdgcn_update_dpp(0ll, v, 0x124, 0xF, 0xF, true); break;
    case 8: v = __builtin_amdgcn_update_dpp(0ll, v, 0x128, 0xF, 0xF, true); break;
    }
    return __longlong_as_double(v);
}
__device__ __forceinline__ double rdl(double x, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src));
}
__global__ void __launch_bounds__(256) k(double *out, long long *cyc, int reps)
{
    __shared__ double red[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = 1.0 / (1 + threadIdx.x), acc = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        double y = x + acc;
        for (int o = 32; o > 0; o >>= 1) y = fmax(y, __shfl_xor(y, o));
        __syncthreads();
        if (lane == 0) red[wave] = y;
        __syncthreads();
        acc += fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
    long long t1 = clock64();
    int par = 0;
    for (int r = 0; r < reps; ++r) {
        double y = x + acc;
        y = fmax(y, ror(y, 8)); y = fmax(y, ror(y, 4)); y = fmax(y, ror(y, 2)); y = fmax(y, ror(y, 1));
        y = fmax(fmax(rdl(y, 0), rdl(y, 16)), fmax(rdl(y, 32), rdl(y, 48)));
        double *s = red + 4 * par;
        if (lane == 0) s[wave] = y;
        __syncthreads();
        par ^= 1;
        acc += fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
    }
    long long t2 = clock64();
    int cnt = 0;
    for (int r = 0; r < reps; ++r) cnt += __syncthreads_or(x + acc + cnt < 0.0);
    long long t3 = clock64();
    int *fl = (int *)(red + 16);
    for (int r = 0; r < reps; ++r) {
        const bool w = __ballot(x + acc + cnt < 0.0) != 0;
        int *f = fl + 4 * par;
        if (lane == 0) f[wave] = w;
        __syncthreads();
        par ^= 1;
        cnt += (f[0] | f[1] | f[2] | f[3]);
    }
    long long t4 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
    out[blockIdx.x * 256 + threadIdx.x] = acc + cnt;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 512 * 256 * 8); hipMalloc(&c, 64);
    const int reps = 200;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, o, c, reps);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("ticks per call: block_max shfl + 2 barriers %.0f | DPP + readlane + 1 barrier %.0f | __syncthreads_or %.0f | ballot + 1 barrier %.0f\n",
               h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps, h[3] / (double)reps);
    }
    return 0;
}
