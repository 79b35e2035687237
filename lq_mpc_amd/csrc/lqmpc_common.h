// lqmpc_common.h -- kernel parameter block and fp64 device helpers shared by the HIP kernels.
// gfx950 (MI355X) only.
#pragma once
// LQMPC_JIT: compiled at run time by hiprtc (lqmpc_jit.hip) -- the HIP device API is built in there, and the device headers
// (this one, lqmpc_wg_linalg.h, lqmpc_r16_setup.h, lqmpc_r16_body.h) use no standard-library header, so that the run-time
// compile depends on nothing outside the library.
#ifndef LQMPC_JIT
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace lqmpc {

// ---- compile-time loops without <utility>: f(ic<I0>{}) ... f(ic<I1 - 1>{}) ----
template <int I> struct ic { static constexpr int value = I; };
template <int I0, int I1, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I0 < I1) { f(ic<I0>{}); sfor<I0 + 1, I1>(f); }
}

enum Mode : int { MODE_SOLVE = 0, MODE_ROLLOUT = 1, MODE_MAXVN = 2, MODE_PROBE = 3, MODE_SWEEP = 4 /* max V_N, then the rollout */ };

// Offsets (in doubles) into the small "shared" device block that holds the data common to the
// whole batch: Q, R, P, lb, ub, x_ref (nx,N), u_ref (nu,N), A_true, B_true, x0s (nx,K).
struct SharedOff {
    int Q, R, P, lb, ub, xref, uref, At, Bt, x0s;
    int Kg;                               // the first gain of the N-stage problem on the shared plant (nu x nx): the order's roll (order_roll)
};

struct KParams {
    int nx, nu, N, n;
    int T, K, mode;
    int true_per_instance;
    int has_ref;
    int has_lin;                          // references or an off-centre box: the linear term has a constant part (host: has_ref || any lb + ub != 0)
    int order_roll;                       // MODE_PROBE: key = the last step of a clipped roll of the shared plant under so.Kg whose input saturates
    int max_iter, polish, presolve, warm_start;
    double eps, tau, z0_scale;
    long long Bsz;
    long long ws_stride;   // generic kernel: instances per workspace entry row
    const double *A, *B, *x0, *At, *Bt;   // per-instance, instance-minor (device)
    const double *sh;                     // shared block (device)
    SharedOff so;
    double *ws;                           // generic kernel workspace (device)
    double *u0, *VN, *JT, *X, *U, *MV;    // outputs (device; X, U may be null)
    int *status, *iters;                  // may be null
    const int *perm;                      // processing order (slot -> instance), null = natural order
    double *key;                          // MODE_PROBE output per instance: (difficulty bucket, position inside the bucket), two ints
    int *hist;                            // MODE_PROBE: instances per difficulty bucket (ORDER_CELLS counters, zero on entry)
    int *hist_next;                       // MODE_PROBE: the counters of the NEXT call, zeroed by this launch (the two sets alternate: no fill launch per call)
    double *stage;                        // MODE_PROBE output: instance-major [A|B|x0] records (null: none)
    const double *rec;                    // input records staged by the probe (null: read A, B, x0 directly)
    long long nwide;                      // tiered rollout: the first nwide slots of the order get a wavefront each
    int *fail_list, *fail_count;          // lqmpc_r16_kernel: instances it hands back (status 3), and their number
    const int *count_dev;                 // packed kernel as the fallback pass: number of slots to process, on the device
    int r16_maxit;                        // active-set iteration cap of the 16-lane-row layout before it hands an instance back
    int r16_build;                        // options.r16_build: -1 auto, 0 throughput build, 1 latency build
};

constexpr int ORDER_BUCKETS = 512;           // difficulty buckets of the ordering: 16 per binade of the key over [2^-2, 2^30)
constexpr int ORDER_COPIES = 8;              // counters per bucket (wavefront w uses copy w % 8): spreads the atomics on a popular bucket
constexpr int ORDER_CELLS = ORDER_BUCKETS * ORDER_COPIES;
constexpr int ORDER_PAD = 16;                // ints between two counters: one 64-byte line each (atomics on one line serialise)

// ---- fp64 reciprocal / reciprocal square root: hardware seed + Newton steps ----
// v_rcp_f64 / v_rsq_f64 give a seed good to ~2^-26 or better; two Newton steps reach ~1 ulp
// without the div_scale/div_fmas/div_fixup sequence of an IEEE divide.  Arguments here are
// always finite, positive and far from the denormal range (slacks, pivots).
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// one Newton step on the hardware seed (~2^-26 -> ~2^-52): for the pivots of the Gauss-Jordan eliminations, where the reciprocal
// only scales a row and its last bit does not matter
__device__ __forceinline__ double frcp1(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double frsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    // y <- y + y*e*(1/2 + 3/8 e),  e = 1 - x y^2   (cubic step), then one quadratic step
    double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
    e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    return y;
}

}  // namespace lqmpc
