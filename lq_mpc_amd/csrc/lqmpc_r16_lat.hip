// Latency build of the 16-lane-row kernel (lqmpc_r16_body.h) for the shapes whose throughput build (lqmpc_r16.hip)
// is the two-waves-per-SIMD, low-register one: fully unrolled, one wave per SIMD, the whole 512-register budget.
// A batch that does not fill the chip past one wave per SIMD (<= 4096 instances) gains nothing from the second
// wave and runs 10-30 % faster on this build (one-shot 1024 instances of C3: 56 us vs 67 us); the host picks by batch
// size (launch_r16).  Same algorithm and iteration; the two builds differ in unrolling and in where the constants live.
#include <cstdio>
#include "lqmpc_r16_body.h"

namespace lqmpc {

template <int NX, int NU, int N, int MODE>
__global__ void __launch_bounds__(64, 1) lqmpc_r16_lat_kernel(KParams p)
{
    using C = R16<NX, NU, N, 16, false>;
    __shared__ double lds_raw[C::IPW * C::INST];
    r16_body<NX, NU, N, MODE, 16, 1>(p, lds_raw, (long long)blockIdx.x * C::IPW, p.Bsz);
}

template <int NX, int NU, int N>
static void launch_lat_one(const KParams &p, hipStream_t stream)
{
    const dim3 grid((unsigned)((p.Bsz + 3) / 4));
    if (p.mode == MODE_SOLVE) hipLaunchKernelGGL((lqmpc_r16_lat_kernel<NX, NU, N, MODE_SOLVE>), grid, dim3(64), 0, stream, p);
    else if (p.mode == MODE_MAXVN) hipLaunchKernelGGL((lqmpc_r16_lat_kernel<NX, NU, N, MODE_MAXVN>), grid, dim3(64), 0, stream, p);
    else if (p.mode == MODE_SWEEP) hipLaunchKernelGGL((lqmpc_r16_lat_kernel<NX, NU, N, MODE_SWEEP>), grid, dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((lqmpc_r16_lat_kernel<NX, NU, N, MODE_ROLLOUT>), grid, dim3(64), 0, stream, p);
}

bool launch_r16_lat(const KParams &p, hipStream_t stream)
{
    if (p.nx == 4 && p.nu == 2 && p.N == 10) { launch_lat_one<4, 2, 10>(p, stream); return true; }
    if (p.nx == 2 && p.nu == 1 && p.N == 20) { launch_lat_one<2, 1, 20>(p, stream); return true; }
    return false;
}

}  // namespace lqmpc
