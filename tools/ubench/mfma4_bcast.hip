// Dev microbenchmark (GPU box): the A-operand broadcast of v_mfma_f64_4x4x4_4b_f64 (CBSZ / ABID): which block's A does block g multiply with?
// hipcc -O3 --offload-arch=gfx950 -o mfma4_bcast mfma4_bcast.hip && ./mfma4_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
template <int CBSZ, int ABID>
__global__ void k(const double *a, const double *b, double *d) { const int l = threadIdx.x; d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0); }
static double ha[64], hb[64], hd[64], *da, *db, *dd;
template <int CBSZ, int ABID> static void run()
{
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    printf("cbsz %d abid %d: block g multiplies its B with the A of block", CBSZ, ABID);
    for (int g = 0; g < 4; ++g) {
        int found = -1;
        for (int s = 0; s < 4; ++s) {
            double err = 0;
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
                double acc = 0;
                for (int kk = 0; kk < 4; ++kk) acc += ha[16 * kk + 4 * s + i] * hb[16 * kk + 4 * g + j];      // entry [x][y] of block g in lane 16x + 4g + y: D = A'B
                err = fmax(err, fabs(acc - hd[16 * i + 4 * g + j]));
            }
            if (err < 1e-12) found = s;
        }
        printf(" %d", found);
    }
    printf("\n");
}
int main()
{
    srand(2);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX; hb[i] = rand() / (double)RAND_MAX; }
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    run<0, 0>(); run<1, 0>(); run<1, 1>(); run<2, 0>(); run<2, 1>(); run<2, 2>(); run<2, 3>();
    return 0;
}
