// lqmpc_probe.h -- the difficulty probe of the ordered rollouts (options.order): one launch computes a key per instance, reserves
// its position inside its difficulty bucket and stages the instance-major [A | B | x0] records the sorted walk reads.
// Device code only (no standard-library header): included by lqmpc_spec.hip for the prebuilt shapes and compiled at run time for the
// others (lqmpc_jit.hip).
#pragma once
#include "lqmpc_common.h"

namespace lqmpc {

// ---------------- difficulty probe (options.order) ----------------
// One instance per lane.  Two keys: for rollouts on a shared plant a clipped roll of that plant (order_roll, below); otherwise the
// largest stage gradient of the FREE response over the horizon, in
// units of what one input can counter:  max_r max_k |B_k' Q A^(r+1) x0| / ((B'QB + R)_kk h_k).
// It needs neither condensing nor a factorisation (240 FMAs per instance at C3) and orders the batch
// almost as well as the exact overshoot of the unconstrained minimiser (20.5 % vs 19.8 % of wave-steps
// left with a constrained instance on C3; natural order 48.7 %).  A heuristic: it only decides which
// instances share a wavefront, never a result.  The same pass stages the instance-major [A | B | x0]
// records the sorted walk reads.
template <int NX, int NU, int N>
__device__ __forceinline__ void probe_body(const KParams &p)
{
    const long long Bsz = p.Bsz;
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    // housekeeping that used to be a fill launch per call: the counters the NEXT call's probe will count into (the two sets
    // alternate; the last reader of that set, the previous call's scatter, is long done) and this call's hand-back count
    if (p.hist_next) {
        const long long total = (long long)gridDim.x * 64;
        for (long long e = b; e < (long long)ORDER_CELLS * ORDER_PAD; e += total) p.hist_next[e] = 0;
    }
    if (p.fail_count && b == 0) { p.fail_count[0] = 0; p.fail_count[1] = 0; }
    if (b >= Bsz) return;
    constexpr int REC = NX * NX + NX * NU + NX;
    const double *sh = p.sh;
    double A[NX][NX], Bm[NX][NU], x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) A[i][j] = p.A[(long long)(i * NX + j) * Bsz + b];
#pragma unroll
        for (int k = 0; k < NU; ++k) Bm[i][k] = p.B[(long long)(i * NU + k) * Bsz + b];
        x[i] = p.x0[(long long)i * Bsz + b];
    }
    if (p.stage) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) p.stage[b * REC + i * NX + j] = A[i][j];
#pragma unroll
            for (int k = 0; k < NU; ++k) p.stage[b * REC + NX * NX + i * NU + k] = Bm[i][k];
            p.stage[b * REC + NX * NX + NX * NU + i] = x[i];
        }
    }
    int raw;
    if (p.order_roll) {
        // Rollouts on a shared plant (zero references, centred box): the instance's own closed loop, approximately -- the PLANT rolled
        // forward from x0 under the first gain of the N-stage problem on it (host: so.Kg), inputs clipped.  Key = the last step whose
        // input saturates (the MPC steps up to there are the constrained ones), 16ths of a margin behind it: 1 - 1/m at that step,
        // or m at x0 where no step saturates (m = max_k |u_k| / h_k).  Wave-steps with a constrained instance at C3, four instances
        // per wavefront in this order: 15.1 % (the free-response key below: 18.3 %, the true count of constrained steps: 14.8 %,
        // natural order: 29.3 %; hard mix 67.3 / 78.6 / 66.2 / 91.5 %: tests/dev/order_keys.py).  ~1000 FMAs per instance on
        // scalar operands.
        double At[NX][NX], Bt[NX][NU], Kg[NU][NX], hk[NU], nhk[NU], hinv[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) At[i][j] = sh[p.so.At + i * NX + j];
#pragma unroll
            for (int k = 0; k < NU; ++k) Bt[i][k] = sh[p.so.Bt + i * NU + k];
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
#pragma unroll
            for (int j = 0; j < NX; ++j) Kg[k][j] = sh[p.so.Kg + k * NX + j];
            hk[k] = 0.5 * (sh[p.so.ub + k] - sh[p.so.lb + k]);
            nhk[k] = -hk[k];
            hinv[k] = 1.0 / hk[k];
        }
        const int Ts = p.T < 31 ? p.T : 31;              // (16 (Ts) + 15 < ORDER_BUCKETS)
        int last = 0;
        double marg = 0.0;
#pragma unroll 1
        for (int t = 0; t < Ts; ++t) {
            double u[NU], m = 0.0;
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                double uk = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) uk = __builtin_fma(-Kg[k][j], x[j], uk);
                m = fmax(m, fabs(uk) * hinv[k]);
                u[k] = fmin(fmax(uk, nhk[k]), hk[k]);
            }
            const bool over = m > 1.0;
            last = over ? t + 1 : last;
            marg = (over || t == 0) ? m : marg;
            double xn[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) acc = __builtin_fma(At[i][j], x[j], acc);
#pragma unroll
                for (int k = 0; k < NU; ++k) acc = __builtin_fma(Bt[i][k], u[k], acc);
                xn[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (__ballot(t - last < 3) == 0ull) break;   // no lane of the wavefront saturated in the last three steps: they have settled
        }
        double xs = 0.0;                                 // (a NaN anywhere has reached every component by now: first in the order)
#pragma unroll
        for (int i = 0; i < NX; ++i) xs += fabs(x[i]);
        const double g = last > 0 ? 1.0 - 1.0 / marg : fmin(marg, 0.999);
        raw = (xs < 1e300 && marg == marg) ? 16 * last + (int)(16.0 * g) : ORDER_BUCKETS - 1;
    } else {
        double QB[NX][NU], dinv[NU];        // Q B and 1 / ((B'QB + R)_kk h_k)
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) t = __builtin_fma(sh[p.so.Q + i * NX + j], Bm[j][k], t);
                QB[i][k] = t;
            }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            double t = sh[p.so.R + k * NU + k];
#pragma unroll
            for (int i = 0; i < NX; ++i) t = __builtin_fma(Bm[i][k], QB[i][k], t);
            dinv[k] = 1.0 / (t * 0.5 * (sh[p.so.ub + k] - sh[p.so.lb + k]));
        }
        double key = 0.0;
#pragma unroll 1
        for (int r = 0; r < N; ++r) {
            double xn[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) t = __builtin_fma(A[i][j], x[j], t);
                xn[i] = t;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                double g = 0.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) g = __builtin_fma(QB[i][k], x[i], g);
                key = fmax(key, fabs(g) * dinv[k]);
            }
        }
        const double kk = (key == key) ? key : 1e300;
        // The order only has to group similar instances, hardest first: a bucket sort on the logarithm of the key (exponent and
        // four mantissa bits of the fp64: 16 buckets per binade, clamped to [2^-2, 2^30): an instance whose key is below 1/4 never meets its bounds; finer buckets cost more atomics
        // than they save in the rollout.  Each wavefront reserves its
        // positions inside a bucket with one atomic per distinct bucket it holds; lqmpc_order_scatter_kernel turns
        // (bucket, position) into the slot of the instance.
        raw = (int)((unsigned)__double2hiint(kk) >> 16) - ((1023 - 2) << 4);
    }
    const int bucket = raw < 0 ? 0 : (raw > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : raw);
    const int lane = threadIdx.x;
    int my_leader = lane, rank = 0, cnt = 0;           // the lanes of my bucket: first of them, my rank among them, their number
    unsigned long long todo = __ballot(1);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __shfl(bucket, leader);
        const unsigned long long same = __ballot(bucket == lb);
        if (bucket == lb) {
            my_leader = leader;
            rank = __popcll(same & ((1ull << lane) - 1ull));
            cnt = __popcll(same);
        }
        todo &= ~same;
    }
    const int cell = bucket * ORDER_COPIES + (int)(blockIdx.x % ORDER_COPIES);
    int base = 0;
    if (lane == my_leader) base = atomicAdd(&p.hist[cell * ORDER_PAD], cnt);   // all the wave's reservations in flight at once
    base = __shfl(base, my_leader);
    ((int2 *)p.key)[b] = make_int2(cell, base + rank);
}

template <int NX, int NU, int N>
__global__ void __launch_bounds__(64) lqmpc_probe_kernel(KParams p) { probe_body<NX, NU, N>(p); }

}  // namespace lqmpc
