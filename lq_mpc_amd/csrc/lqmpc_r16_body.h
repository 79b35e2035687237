// lqmpc_r16_body.h -- device code of the 16-lane-row rollout (see lqmpc_r16.hip for the description);
// shared by the stand-alone kernel there and by the tiered launch in lqmpc_spec.hip.
#pragma once
#include "lqmpc_wg_linalg.h"
#include "lqmpc_r16_setup.h"

namespace lqmpc {

using wg::ldsd;
using wg::ldsi;
using wg::rowb;
using wg::fmac_rowb;
using wg::fmac_rowb_self;
using wg::fmac_rowb4;
using wg::fmac_rowb_self4;
using wg::fmac_rowb_self2;
using wg::fmac_rowb_self3;
using wg::fmac_rowb_lanes4;
using wg::fmac_rowb_lanes4x2;
using wg::dpp_settle;

#ifdef LQMPC_R16_PROF
__device__ long long g_r16_prof[32];
#define RPROF(k) do { const long long now_ = clock64(); if (threadIdx.x == 0 && blockIdx.x == PROFBLK) g_r16_prof[k] += now_ - prof_t; if (k < 7) prof_ph[k] += now_ - prof_t; prof_t = clock64(); } while (0)
#define RPROF_START long long prof_t = clock64()
// whole-grid accounting (every wavefront adds): [8] cycles of iteration-free wave-steps, [9] cycles of wave-steps with active-set
// iterations, [10] / [11] their numbers, [12] wave-iterations, [13] set-up cycles, [14] total cycles, [15] wavefronts
#define RPROF_ADD(k, v) do { if (threadIdx.x == 0) atomicAdd((unsigned long long *)&g_r16_prof[k], (unsigned long long)(v)); } while (0)
#ifndef PROFBLK
#define PROFBLK 0
#endif
#else
#define RPROF(k) do { } while (0)
#define RPROF_START do { } while (0)
#define RPROF_ADD(k, v) do { } while (0)
#endif

typedef unsigned long long mask_t;        // one bit per row of an instance (n <= 64)

// LPI = lanes per instance: 16 (four instances per wavefront, row broadcasts by DPP row_newbcast) or 64 (one instance
// per wavefront for 32 < n <= 48, row broadcasts by v_readlane: the value becomes a scalar operand)
template <int NX, int NU, int N, int LPI = 16, bool PACKED = false>
struct R16 {
    static constexpr int n = N * NU;
    static constexpr int RB = (n + LPI - 1) / LPI;
    static constexpr int IPW = 64 / LPI;              // instances per wavefront
    static constexpr int CS = 4 * (((n + 1) / 2 + 3) / 4);   // most unknowns of a gathered system: min(|A|, |F|) <= n / 2, in groups of four
    static constexpr int LDW = n + 1;                // row stride of P and W: odd, so that a column read is conflict-free
    // P and W: full rows (every access is row base + constant) or, for the builds that need the LDS for a second wave per
    // SIMD, the packed lower triangle (an address select per access)
    static constexpr int PK = PACKED ? n * (n + 1) / 2 : n * LDW;
    static constexpr int VEC = LPI * RB;
    // LDS per instance, in doubles: P | W | r | x | y | list (CS ints)
    static constexpr int oP = 0, oW = PK, oR = 2 * PK, oX = oR + VEC, oY = oX + VEC, oL = oY + VEC;
    static constexpr int oC = oL + CS / 2;            // Q | R | A | B | P_T: the open-loop value function's constants (read from LDS
    static constexpr int CN0 = 3 * NX * NX + NU * NU + NX * NU;  // when built for two waves per SIMD)
    static constexpr int CN1 = (NX > NU ? NX : NU) * (2 * NX + 2 * NU);   // ... or the closed loop's per-lane rows of [A_true B_true | Q | R]
    static constexpr int CN = CN0 > CN1 ? CN0 : CN1;
    static constexpr int SETUP = n * NX;                                 // the hand-over of G aliases the P / W regions (lqmpc_r16_setup.h)
    static constexpr int oG = 0;
    static constexpr int END = oC + CN + (CN & 1);
    static constexpr int oD = (END > SETUP) ? END : SETUP;        // a dummy slot BEHIND both: predicated LDS stores go there instead of toggling exec
    static constexpr int INST = oD + 2;
    static constexpr int MAXIT = 12;
    static_assert(LPI == 16 || LPI == 64, "a DPP row or the whole wavefront");
    static_assert((n + 1) / 2 <= CS && n <= 64, "the smaller side must fit the gathered system");
};


// ---- the per-instance primitives in the two mappings ----
// ballot over the lanes of my instance
template <int LPI>
__device__ __forceinline__ mask_t iballot(bool c, int q)
{
    if constexpr (LPI == 16) return (__ballot(c) >> (16 * q)) & 0xFFFFull;
    else return __ballot(c);
}
// value of lane k of my instance (k: compile-time constant after unrolling)
template <int LPI>
__device__ __forceinline__ double ibcast(double x, int k)
{
    if constexpr (LPI == 16) return rowb(x, k);
    else return wg::rdlane(x, k);
}
// acc += (lane k's x) * y
template <int LPI>
__device__ __forceinline__ void ifmac(double &acc, double x, double y, int k)
{
    if constexpr (LPI == 16) fmac_rowb(acc, x, y, k);
    else acc = __builtin_fma(wg::rdlane(x, k), y, acc);
}
// acc += (lane k's acc) * y
template <int LPI>
__device__ __forceinline__ void ifmac_self(double &acc, double y, int k)
{
    if constexpr (LPI == 16) fmac_rowb_self(acc, y, k);
    else acc = __builtin_fma(wg::rdlane(acc, k), y, acc);
}
// four of each with one lane and one multiplier
template <int LPI>
__device__ __forceinline__ void ifmac4(double *acc, const double *x, double y, int k)
{
    if constexpr (LPI == 16) fmac_rowb4(acc[0], acc[1], acc[2], acc[3], x[0], x[1], x[2], x[3], y, k);
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fma(wg::rdlane(x[e], k), y, acc[e]);
    }
}
template <int LPI>
__device__ __forceinline__ void ifmac_self4(double *acc, double y, int k)
{
    if constexpr (LPI == 16) fmac_rowb_self4(acc[0], acc[1], acc[2], acc[3], y, k);
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fma(wg::rdlane(acc[e], k), y, acc[e]);
    }
}
// The tail of pivot K's own group of four columns (columns K+1 .. 4 (K/4) + 3) together with the right-hand side: one to four
// self-updates behind a single s_nop instead of one each
template <int LPI, int K, int CS>
__device__ __forceinline__ void ifmac_self_tail(double (&S)[CS], double &rhs, double g)
{
    constexpr int r = K % 4;
    if constexpr (LPI == 16) {
        if constexpr (r == 0) fmac_rowb_self4(S[K + 1], S[K + 2], S[K + 3], rhs, g, K);
        else if constexpr (r == 1) fmac_rowb_self3(S[K + 1], S[K + 2], rhs, g, K);
        else if constexpr (r == 2) fmac_rowb_self2(S[K + 1], rhs, g, K);
        else fmac_rowb_self(rhs, g, K);
    } else {
#pragma unroll
        for (int j = K + 1; j < 4 * (K / 4) + 4; ++j) S[j] = __builtin_fma(wg::rdlane(S[j], K), g, S[j]);
        rhs = __builtin_fma(wg::rdlane(rhs, K), g, rhs);
    }
}
template <int LPI>
__device__ __forceinline__ void isettle(double &x)
{
    if constexpr (LPI == 16) dpp_settle(x);
}
// 32-bit value of lane c of my 16-lane row (compiler builtin: it places the wait states itself)
__device__ __forceinline__ int rowb_i(int x, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: return __builtin_amdgcn_update_dpp(0, x, 0x150 + C, 0xF, 0xF, true);
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
    return x;
}
// reductions over the 16 lanes of a row by rotations (row_ror 8, 4, 2, 1): every lane ends up with the result
__device__ __forceinline__ unsigned row_or(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xF, 0xF, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xF, 0xF, false);
    return x;
}
__device__ __forceinline__ unsigned row_umax(unsigned x)
{
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xF, 0xF, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xF, 0xF, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xF, 0xF, false));
    return x;
}

// compile-time loop: f(ic<0>{}) ... f(ic<CNT - 1>{})
template <int CNT, typename F>
__device__ __forceinline__ void static_for(F &&f) { sfor<0, CNT>(f); }

// OCC = 2: built for two waves per SIMD (256 registers): the horizon loops of the condensing stay rolled and the stage weights
// and the plant live in LDS instead of registers (they are the loop-invariant values the allocator would otherwise reload
// from scratch in every step).
template <int NX, int NU, int N, int MODE, int LPI = 16, int OCC = 1>
__device__ __forceinline__ void r16_body(const KParams &p, double *lds_raw, long long slot0, long long slot_end)
{
    constexpr bool PACKED = (OCC == 2);
    using C = R16<NX, NU, N, LPI, PACKED>;
    constexpr int CS = C::CS;
    constexpr int n = C::n, RB = C::RB, LDW = C::LDW;
    constexpr int REC = NX * NX + NX * NU + NX;
    const int lane = threadIdx.x, q = lane / LPI, i = lane % LPI;
    ldsd *L = (ldsd *)lds_raw + q * C::INST;
    ldsd *Pp = L + C::oP, *Wp = L + C::oW, *rL = L + C::oR, *xL = L + C::oX, *yL = L + C::oY;
    ldsi *list = (ldsi *)(L + C::oL);
    const long long Bsz = p.Bsz;
    const long long b_raw = slot0 + q;
    const bool valid = b_raw < slot_end;
    const long long slot = valid ? b_raw : slot_end - 1;
    const long long b = p.perm ? (long long)p.perm[slot] : slot;
    const double *sh = p.sh;
    const mask_t nmask = (n == 64) ? ~0ull : ((1ull << n) - 1ull);

    int rw[RB], tri[RB];
    bool vrow[RB];
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        rw[s] = i + LPI * s;
        tri[s] = rw[s] * (rw[s] + 1) / 2;
        vrow[s] = rw[s] < n;
    }

    // element (r, j) of P or W for a row r of mine (tr = r (r + 1) / 2) and any column j; (a, b) for two run-time indices
    auto ad = [&](int r, int tr, int j) -> int { if constexpr (PACKED) return (j <= r) ? tr + j : j * (j + 1) / 2 + r; else return r * LDW + j; };
    auto ad2 = [&](int a, int b2) -> int {
        if constexpr (PACKED) { const int hi = a > b2 ? a : b2, lo = a > b2 ? b2 : a; return hi * (hi + 1) / 2 + lo; }
        else return a * LDW + b2;
    };

    // ---------------- condensing (utils_class.py:62-75 in matrix form) ----------------
    // Row i = bi*NU + ui has a = N-1-bi stages to go; with ma_i = column ui of M_a = A^a B,
    //   H(i, j) = sum_{s>=0} T(i + s NU, j + s NU),   T(i, j) = ma_i . (P_T ma_j) + ma_{i+NU} . ((Q - P_T) ma_{j+NU}),
    //   Fq(i, :) = 2 [ (P_T ma_i)' A^N + sum_{d>=1} (Q ma_{i+d NU})' A^(N-d) ]
    // (the displacement structure of Gamma'Qbar Gamma): O(n^2 NX) work and no accumulator matrix carried through
    // a loop.  The row images ma, P_T ma, Q ma and the powers of A sit in LDS (the W region and the vectors
    // behind it, free until the inverse is stored).
    double G[RB][NX], vr[RB];
    // P is read by the primal side of an iteration only (|A| > n / 2).  One instance per wavefront (LPI = 64, one wavefront in a
    // hundred at C4): built in the set-up when the cold-start active set predicts a primal-side iteration at step 0, otherwise the
    // first time an iteration needs it (C4 4.92 -> 4.58 ms).  Four instances per wavefront: in every set-up -- the mere presence of
    // the on-demand call in qp() costs every primal-side iteration of these kernels ~25 % (C3 hard mix 1.27 -> 1.56 ms, default mix
    // 0.338 -> 0.345 ms, measured with the call inside the iteration loop, outside it, and never taken).
    constexpr bool LAZY_P = (LPI == 64);
    bool P_ready = !LAZY_P;
    // the set-up's guess of the active set at x0 (lqmpc_r16_setup.h, ROLL): the face the first, cold-started QP begins from; the
    // stage gains are kept in registers through the sweep, so short horizons only, and where the first QP is the one at x0
#ifdef LQMPC_NO_ROLL                         // (dev builds: tools/prof_build.sh)
    constexpr bool ROLL = false;
#else
    constexpr bool ROLL = LPI == 16 && N * ((NX + 3) / 4) <= 12 && MODE != MODE_MAXVN;
#endif
    unsigned cold32[2] = {0u, 0u};
    bool cold_armed = false;                 // set by the caller of qp() for the QP at x0
#ifdef LQMPC_R16_PROF
    const long long prof_t0 = clock64();
    int prof_wit = 0, prof_slow = 0, prof_fast = 0;
    long long prof_slowt = 0;
    long long prof_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    {
        // Set-up on the matrix core (lqmpc_r16_setup.h): Riccati recursion, W by rank-NU tile updates, G from the same sweep, P from
        // its Toeplitz form -- v_mfma_f64_4x4x4_4b_f64 products only.  MFMA block g = (lane >> 2) & 3 works for the instance of lanes
        // 16g .. 16g+15 (LPI = 16), so the set-up takes its inputs and its LDS by g and hands G back through LDS.
        RPROF_START;
        long long bg = b;
        ldsd *Lg = L;
        if constexpr (LPI == 16) {
            const int gq = (lane >> 2) & 3;
            const long long sraw = slot0 + gq, sl = sraw < slot_end ? sraw : slot_end - 1;
            bg = p.perm ? (long long)p.perm[sl] : sl;
            Lg = (ldsd *)lds_raw + gq * C::INST;
        }
        r16_setup_mfma<NX, NU, N, LPI, PACKED, RB, ROLL>(setup_args(p), bg, Lg, L, C::oW, C::oG, C::oD, G, roll_args(p), cold32[0], cold32[1]);
        if constexpr (!LAZY_P) r16_build_P<NX, NU, N, LPI, PACKED>(setup_args(p), bg, Lg, C::oP, C::oD);
        RPROF(5);
        // constant part of the unconstrained minimiser: v_r = -W (2 gref + P centre) = -2 W gref - centre (references / off-centre boxes only)
        const bool has_lin = p.has_lin != 0;
#pragma unroll
        for (int s = 0; s < RB; ++s) vr[s] = 0.0;
        if (has_lin) {
            if (p.has_ref) {
                //   d_r = -xref_r,  lam_r = Q_r d_r + A' lam_{r+1},  gref_r = B' lam_r - R uref_r   (columns r <-> x_{r+1}, u_r)
                double qr[RB];
#pragma unroll
                for (int s = 0; s < RB; ++s) qr[s] = 0.0;
                double lam[NX], A2[NX][NX], B2[NX][NU];
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    lam[a] = 0.0;
#pragma unroll
                    for (int c = 0; c < NX; ++c) A2[a][c] = p.rec ? p.rec[b * REC + a * NX + c] : p.A[(long long)(a * NX + c) * Bsz + b];
#pragma unroll
                    for (int k = 0; k < NU; ++k) B2[a][k] = p.rec ? p.rec[b * REC + NX * NX + a * NU + k] : p.B[(long long)(a * NU + k) * Bsz + b];
                }
#pragma unroll 1
                for (int r = N - 1; r >= 0; --r) {
                    const int oQ = (r < N - 1) ? p.so.Q : p.so.P;
                    double l2[NX];
#pragma unroll
                    for (int a = 0; a < NX; ++a) {
                        double t = 0.0;
#pragma unroll
                        for (int c = 0; c < NX; ++c) t = __builtin_fma(sh[oQ + a * NX + c], -sh[p.so.xref + c * N + r], t);
#pragma unroll
                        for (int c = 0; c < NX; ++c) t = __builtin_fma(A2[c][a], lam[c], t);
                        l2[a] = t;
                    }
#pragma unroll
                    for (int a = 0; a < NX; ++a) lam[a] = l2[a];
#pragma unroll
                    for (int s = 0; s < RB; ++s) {
                        const int ui = rw[s] % NU;
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < NU; ++k) {
                            double tk = 0.0;
#pragma unroll
                            for (int a = 0; a < NX; ++a) tk = __builtin_fma(B2[a][k], lam[a], tk);
#pragma unroll
                            for (int j = 0; j < NU; ++j) tk = __builtin_fma(-sh[p.so.R + k * NU + j], sh[p.so.uref + j * N + r], tk);
                            t = (ui == k) ? tk : t;
                        }
                        qr[s] = (vrow[s] && rw[s] / NU == r) ? 2.0 * t : qr[s];
                    }
                }
#pragma unroll
                for (int s = 0; s < RB; ++s) xL[rw[s]] = qr[s];
                __syncthreads();
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    double t = 0.0;
                    const int mr = vrow[s] ? rw[s] : 0, mt = vrow[s] ? tri[s] : 0;
#pragma unroll
                    for (int j = 0; j < n; ++j) t = __builtin_fma(Wp[ad(mr, mt, j)], xL[j], t);
                    vr[s] = -t;
                }
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const int k = rw[s] % NU;
                vr[s] = vrow[s] ? vr[s] - 0.5 * (sh[p.so.ub + k] + sh[p.so.lb + k]) : 0.0;
            }
        }
        if constexpr (LAZY_P && MODE != MODE_MAXVN) {
            // more than half of the rows of the unconstrained minimiser at x0 outside the box: the first iteration takes the primal side
            mask_t out = 0;
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                double acc = vr[s];
#pragma unroll
                for (int a = 0; a < NX; ++a) acc = __builtin_fma(G[s][a], p.rec ? p.rec[b * REC + NX * NX + NX * NU + a] : p.x0[(long long)a * Bsz + b], acc);
                const int k = rw[s] % NU;
                out |= iballot<LPI>(vrow[s] && !(fabs(acc) <= 0.5 * (sh[p.so.ub + k] - sh[p.so.lb + k])), q) << (LPI * s);
            }
            if (__ballot(2 * __popcll(out) > n) != 0ull) {
                r16_build_P<NX, NU, N, LPI, PACKED>(setup_args(p), bg, Lg, C::oP, C::oD);
                P_ready = true;
            }
        }
        RPROF(6);
    }
    RPROF_START;
    RPROF_ADD(13, clock64() - prof_t0);

    // ---------------- the requested operation ----------------
    // an opaque copy of the instance index for everything below: the addresses derived from it are recomputed here instead of
    // being kept (spilled) across the whole set-up
    long long bq = b;
    asm volatile("" : "+v"(bq));
    // (the box of my rows is fetched here, not at the top: nothing of the set-up above needs it and it would only be spilled)
    double h[RB], ctr[RB];
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        const int k = rw[s] % NU;
        h[s] = vrow[s] ? 0.5 * (sh[p.so.ub + k] - sh[p.so.lb + k]) : 1.0;
        ctr[s] = vrow[s] ? 0.5 * (sh[p.so.ub + k] + sh[p.so.lb + k]) : 0.0;
    }
    // stage weights of the open-loop value function once (the closed loop keeps its rows of them per lane, see below)
    double Qm[NX][NX], Rm[NU][NU];
    ldsd *cQ = L + C::oC, *cR = cQ + NX * NX;
    if constexpr (MODE == MODE_ROLLOUT) {
    } else if constexpr (OCC == 2) {
        if (i == 0) {
#pragma unroll
            for (int a = 0; a < NX; ++a) {
#pragma unroll
                for (int c = 0; c < NX; ++c) {
                    cQ[a * NX + c] = sh[p.so.Q + a * NX + c];
                }
            }
#pragma unroll
            for (int k = 0; k < NU; ++k)
#pragma unroll
                for (int j = 0; j < NU; ++j) cR[k * NU + j] = sh[p.so.R + k * NU + j];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int a = 0; a < NX; ++a) {
#pragma unroll
            for (int c = 0; c < NX; ++c) Qm[a][c] = sh[p.so.Q + a * NX + c];
        }
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int j = 0; j < NU; ++j) Rm[k][j] = sh[p.so.R + k * NU + j];
    }
    auto Qv = [&](int a, int c) -> double { if constexpr (OCC == 2) return cQ[a * NX + c]; else return Qm[a][c]; };
    auto Rv = [&](int k, int j) -> double { if constexpr (OCC == 2) return cR[k * NU + j]; else return Rm[k][j]; };
    const bool writer = valid && i == 0;
    mask_t pL = 0, pU = 0;                  // active set of the previous step (row-uniform bit masks)
    int iters = 0, status = 0;
    // ---- one box QP at state x: v <- the optimum (my rows); updates the warm-start face, iters, status ----
    auto qp = [&](const double (&x)[NX], double (&v)[RB]) {
        double vu[RB];
        mask_t cl = 0, cu = 0;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            double acc = vr[s];
#pragma unroll
            for (int a = 0; a < NX; ++a) acc = __builtin_fma(G[s][a], x[a], acc);
            vu[s] = acc; v[s] = acc;
            cl |= iballot<LPI>(vrow[s] && acc < -h[s], q) << (LPI * s);
            cu |= iballot<LPI>(vrow[s] && acc > h[s], q) << (LPI * s);
            bad = bad || (vrow[s] && !(fabs(acc) < 1e300));
        }
        const bool rowbad = iballot<LPI>(bad, q) != 0;
        bool busy = ((cl | cu) != 0) && !rowbad;              // row-uniform
        mask_t mL = 0, mU = 0;
        if (busy) {
            if (p.warm_start && (pL | pU) != 0) {
                // the previous face shifted by one stage; the last stage keeps its flags
                const mask_t top = nmask & ~(nmask >> NU);
                mL = (pL >> NU) | (pL & top);
                mU = (pU >> NU) | (pU & top);
            } else if (ROLL && cold_armed && (cold32[0] | cold32[1]) != 0u) {
                mL = cold32[0]; mU = cold32[1];
            } else {
                mL = cl; mU = cu;
            }
        }
        cold_armed = false;
        bool failed = false;
        // The iterations.  Where P is built on demand (LAZY_P) the loop leaves with need_P, the call happens here, outside it, and the
        // loop resumes at the same count.
        int it = 0;
        bool need_P = false;
#pragma unroll 1
        do {
        if (LAZY_P && need_P) {
            long long bg = bq;
            ldsd *Lg = L;
            if constexpr (LPI == 16) {
                const int gq = (lane >> 2) & 3;
                const long long sraw = slot0 + gq, sl = sraw < slot_end ? sraw : slot_end - 1;
                bg = p.perm ? (long long)p.perm[sl] : sl;
                Lg = (ldsd *)lds_raw + gq * C::INST;
            }
            r16_build_P<NX, NU, N, LPI, PACKED>(setup_args(p), bg, Lg, C::oP, C::oD);
            P_ready = true;
            need_P = false;
        }
        if (__ballot(busy) != 0ull) {
#pragma unroll 1
            for (; it < p.r16_maxit; ++it) {
#ifdef LQMPC_R16_PROF
                const long long prof_it0 = clock64();
#endif
                const mask_t mA = mL | mU;
                const int m = __popcll(mA);
                // row-uniform: the smaller side -- with a bias to the dual side where the wavefront holds four instances: one of them on
                // the primal side takes all four through the general path (twice the price of a dual-only iteration), so a face of up
                // to n/2 + 2 rows is still solved from the dual side (C3 rollout 0.327 -> 0.323 ms, hard mix 1.215 -> 1.195 ms;
                // at n = 10 the larger gathered system costs more than it saves)
                constexpr int DUAL_MAX = (LPI == 16 && n >= 16) ? (n / 2 + 2 < CS ? n / 2 + 2 : CS) : n / 2;
                const bool dual = m <= DUAL_MAX;
                const mask_t mC = busy ? (dual ? mA : (~mA & nmask)) : 0ull;
                const int c = __popcll(mC);
                int cw;                                          // wave maximum: uniform loop bound (c is row-uniform: one lane per row)
                if constexpr (LPI == 16)
                    cw = max(max(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 16)),
                             max(__builtin_amdgcn_readlane(c, 32), __builtin_amdgcn_readlane(c, 48)));
                else cw = __builtin_amdgcn_readfirstlane(c);
                const bool any_primal = __ballot(busy && !dual) != 0ull;
                if (LAZY_P && any_primal && !P_ready) { need_P = true; break; }     // wave-uniform: build P outside the loop, come back
                if constexpr (LPI == 16) {
                    if (!any_primal) {
                        // ---- every busy instance of the wavefront is on the dual side: W_AA lam = r_A, v_F = v_unc,F - W_FA lam ----
                        // Nothing but the compacted list and right-hand side goes through LDS.  Lane k < c holds unknown k (row
                        // la_k = list[k]); the column indices of the gathered system and of W_FA reach the other lanes as DPP row
                        // broadcasts of la_k, the multipliers as the DPP operand of the product, the released bounds as a rotate-OR.
                        const unsigned mC32 = (unsigned)mC, mL32 = (unsigned)mL, mA32 = (unsigned)mA;
#pragma unroll
                        for (int s = 0; s < RB; ++s) {                           // (predicated stores go to the dummy slot)
                            const unsigned bit = 1u << rw[s];
                            const bool on = vrow[s] && (mC32 & bit);
                            const int rk = __popc(mC32 & (bit - 1u));
                            const double sg = (mL32 & bit) ? -1.0 : 1.0;
                            list[on ? rk : (C::oD - C::oL) * 2] = rw[s];
                            rL[on ? rk : C::oD - C::oR] = __builtin_fma(-sg, h[s], vu[s]);
                        }
                        __syncthreads();
                        const bool mine = i < c;
                        const int la = mine ? list[i] : 0;
                        double rhs = mine ? rL[i] : 0.0;
                        const int tla = PACKED ? la * (la + 1) / 2 : la * LDW;
                        auto widx = [&](int ra, int ta, int lb, int tlb) -> int {   // element (ra, lb) of W; ta / tlb: tri (packed) or row base
                            if constexpr (PACKED) return max(ta, tlb) + min(ra, lb);
                            else return ta + lb;
                        };
                        double S[CS];
                        static_for<CS / 4>([&](auto bgc) {
                            constexpr int bg = decltype(bgc)::value;
#pragma unroll
                            for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) S[bb] = (bb == i) ? 1.0 : 0.0;
                            if (4 * bg < cw) {                                   // uniform
                                static_for<4>([&](auto bc) {
                                    constexpr int bb = 4 * bg + decltype(bc)::value;
                                    const int lb = rowb_i(la, bb), tlb = PACKED ? rowb_i(tla, bb) : 0;
                                    const double val = Wp[widx(la, tla, lb, tlb)];
                                    S[bb] = mine ? val : S[bb];                  // lanes without an unknown keep a row of the identity
                                });
                            }
                        });
                        bool ok = true;
                        static_for<CS>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            if (k < cw) {
                                isettle<LPI>(S[k]);
                                const double d = ibcast<LPI>(S[k], k);
                                ok = ok && (d > 0.0);
                                const double inv = frcp1(d);
                                const double g = (i == k) ? (inv - 1.0) : -S[k] * inv;
#pragma unroll
                                for (int jg = 0; jg < CS / 4; ++jg)
                                    if (4 * jg > k && 4 * jg < cw) ifmac_self4<LPI>(&S[4 * jg], g, k);
                                ifmac_self_tail<LPI, k>(S, rhs, g);
                            }
                        });
                        const bool rowfail = iballot<LPI>(!ok, q) != 0;
                        // t = W_FA lam on my rows: column la_k of W times the multiplier of lane k
                        double tt[RB];
#pragma unroll
                        for (int s = 0; s < RB; ++s) tt[s] = 0.0;
                        const int trow[2] = {PACKED ? tri[0] : rw[0] * LDW, RB > 1 ? (PACKED ? tri[RB - 1] : rw[RB - 1] * LDW) : 0};
                        static_for<CS / 4>([&](auto kgc) {
                            constexpr int kg = decltype(kgc)::value;
                            if (4 * kg < cw) {                                   // uniform
                                double wv[4][RB];
                                static_for<4>([&](auto kc) {
                                    constexpr int kk = decltype(kc)::value;
                                    const int lk = rowb_i(la, 4 * kg + kk), tlk = PACKED ? rowb_i(tla, 4 * kg + kk) : 0;
#pragma unroll
                                    for (int s = 0; s < RB; ++s) wv[kk][s] = Wp[vrow[s] ? widx(rw[s], trow[s], lk, tlk) : 0];
                                });
                                if constexpr (RB == 2) fmac_rowb_lanes4x2(tt[0], tt[1], rhs, wv[0][0], wv[0][1], wv[1][0], wv[1][1], wv[2][0], wv[2][1], wv[3][0], wv[3][1], 4 * kg);
                                else if constexpr (RB == 1) fmac_rowb_lanes4(tt[0], rhs, wv[0][0], wv[1][0], wv[2][0], wv[3][0], 4 * kg);
                                else {
                                    static_for<4>([&](auto kc) {
                                        constexpr int kk = decltype(kc)::value;
#pragma unroll
                                        for (int s = 0; s < RB; ++s) fmac_rowb(tt[s], rhs, wv[kk][s], 4 * kg + kk);
                                    });
                                }
                            }
                        });
                        // the multipliers decide in their own lanes which bounds stay: rows la_k whose multiplier keeps its sign
                        const unsigned lhi = (unsigned)__double2hiint(rhs) & 0x7fffffffu;
                        const double tol = 1e-10 * __hiloint2double((int)row_umax(mine ? lhi : 0u), 0);
                        const bool isL = (mL32 >> la) & 1u;
                        const unsigned keepL = (mine && isL && rhs <= tol) ? (1u << la) : 0u;
                        const unsigned keepU = (mine && !isL && rhs >= -tol) ? (1u << la) : 0u;
                        unsigned nL32 = row_or(keepL), nU32 = row_or(keepU);
                        bool nf = false;
#pragma unroll
                        for (int s = 0; s < RB; ++s) {
                            const unsigned bit = 1u << rw[s];
                            const bool act = vrow[s] && (mA32 & bit);
                            const double sg = (mL32 & bit) ? -1.0 : 1.0;
                            const double nv = act ? sg * h[s] : vu[s] - tt[s];
                            v[s] = busy ? nv : v[s];
                            const bool fr = vrow[s] && !act;
                            nL32 |= (unsigned)(iballot<LPI>(fr && v[s] < -h[s] * (1.0 + 1e-12), q) << (LPI * s));
                            nU32 |= (unsigned)(iballot<LPI>(fr && v[s] > h[s] * (1.0 + 1e-12), q) << (LPI * s));
                            nf = nf || (vrow[s] && !(fabs(v[s]) < 1e300));
                        }
                        const bool rownf = iballot<LPI>(nf || (mine && !(fabs(rhs) < 1e300)), q) != 0;
#ifdef LQMPC_R16_PROF
                        prof_wit += 1; prof_fast += 1; prof_ph[7] += clock64() - prof_it0;
#endif
                        if (busy) {
                            iters += 1;
                            if (rowfail || rownf) { failed = true; busy = false; }
                            else if ((mask_t)nL32 == mL && (mask_t)nU32 == mU) busy = false;
                            else { mL = nL32; mU = nU32; }
                        }
                        __syncthreads();
                        if (__ballot(busy) == 0ull) break;
                        continue;
                    } else {
                        // ---- mixed wavefront: some busy instance is on the primal side (|A| > n/2: P_FF dlt = (P r)_F,
                        // v_F = v_unc,F + dlt, gradient on A = P_AF dlt - (P r)_A with r = v_unc - s h on A, 0 on F) ----
                        // Same skeleton as the dual-only path, the side chosen per instance: the matrix is P instead of W, the
                        // right-hand side and the gradient need z = P r (one row product, r through LDS by row), and the
                        // solution goes back to its rows through LDS once.
                        const unsigned mC32 = (unsigned)mC, mL32 = (unsigned)mL, mA32 = (unsigned)mA, mU32 = (unsigned)mU;
                        double rr[RB], z[RB];
#pragma unroll
                        for (int s = 0; s < RB; ++s) {
                            const unsigned bit = 1u << rw[s];
                            const bool act = vrow[s] && (mA32 & bit);
                            const double sg = (mL32 & bit) ? -1.0 : 1.0;
                            rr[s] = act ? __builtin_fma(-sg, h[s], vu[s]) : 0.0;
                            rL[rw[s]] = rr[s];
                            z[s] = 0.0;
                        }
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < n; ++j) {
                            const double rj = rL[j];
#pragma unroll
                            for (int s = 0; s < RB; ++s) z[s] = __builtin_fma(Pp[vrow[s] ? ad(rw[s], tri[s], j) : 0], rj, z[s]);
                        }
#pragma unroll
                        for (int s = 0; s < RB; ++s) {                           // compacted list and right-hand side of the chosen side
                            const unsigned bit = 1u << rw[s];
                            const bool on = vrow[s] && (mC32 & bit);
                            const int rk = __popc(mC32 & (bit - 1u));
                            list[on ? rk : (C::oD - C::oL) * 2] = rw[s];
                            yL[on ? rk : C::oD - C::oY] = dual ? rr[s] : z[s];
                        }
                        __syncthreads();
                        const bool mine = i < c;
                        const int la = mine ? list[i] : 0;
                        double rhs = mine ? yL[i] : 0.0;
                        const int tla = PACKED ? la * (la + 1) / 2 : la * LDW;
                        const ldsd *Mx = dual ? Wp : Pp;
                        auto midx = [&](int ra, int ta, int lb, int tlb) -> int {
                            if constexpr (PACKED) return max(ta, tlb) + min(ra, lb);
                            else return ta + lb;
                        };
                        double S[CS];
                        static_for<CS / 4>([&](auto bgc) {
                            constexpr int bg = decltype(bgc)::value;
#pragma unroll
                            for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) S[bb] = (bb == i) ? 1.0 : 0.0;
                            if (4 * bg < cw) {
                                static_for<4>([&](auto bc) {
                                    constexpr int bb = 4 * bg + decltype(bc)::value;
                                    const int lb = rowb_i(la, bb), tlb = PACKED ? rowb_i(tla, bb) : 0;
                                    const double val = Mx[midx(la, tla, lb, tlb)];
                                    S[bb] = mine ? val : S[bb];
                                });
                            }
                        });
                        bool ok = true;
                        static_for<CS>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            if (k < cw) {
                                isettle<LPI>(S[k]);
                                const double d = ibcast<LPI>(S[k], k);
                                ok = ok && (d > 0.0);
                                const double inv = frcp1(d);
                                const double g = (i == k) ? (inv - 1.0) : -S[k] * inv;
#pragma unroll
                                for (int jg = 0; jg < CS / 4; ++jg)
                                    if (4 * jg > k && 4 * jg < cw) ifmac_self4<LPI>(&S[4 * jg], g, k);
                                ifmac_self_tail<LPI, k>(S, rhs, g);
                            }
                        });
                        const bool rowfail = iballot<LPI>(!ok, q) != 0;
                        if (mine) xL[la] = rhs;                                  // the solution back to its rows (primal side reads it)
                        double tt[RB];
#pragma unroll
                        for (int s = 0; s < RB; ++s) tt[s] = 0.0;
                        const int trow[2] = {PACKED ? tri[0] : rw[0] * LDW, RB > 1 ? (PACKED ? tri[RB - 1] : rw[RB - 1] * LDW) : 0};
                        static_for<CS / 4>([&](auto kgc) {
                            constexpr int kg = decltype(kgc)::value;
                            if (4 * kg < cw) {
                                double wv[4][RB];
                                static_for<4>([&](auto kc) {
                                    constexpr int kk = decltype(kc)::value;
                                    const int lk = rowb_i(la, 4 * kg + kk), tlk = PACKED ? rowb_i(tla, 4 * kg + kk) : 0;
#pragma unroll
                                    for (int s = 0; s < RB; ++s) wv[kk][s] = Mx[vrow[s] ? midx(rw[s], trow[s], lk, tlk) : 0];
                                });
                                if constexpr (RB == 2) fmac_rowb_lanes4x2(tt[0], tt[1], rhs, wv[0][0], wv[0][1], wv[1][0], wv[1][1], wv[2][0], wv[2][1], wv[3][0], wv[3][1], 4 * kg);
                                else if constexpr (RB == 1) fmac_rowb_lanes4(tt[0], rhs, wv[0][0], wv[1][0], wv[2][0], wv[3][0], 4 * kg);
                                else {
                                    static_for<4>([&](auto kc) {
                                        constexpr int kk = decltype(kc)::value;
#pragma unroll
                                        for (int s = 0; s < RB; ++s) fmac_rowb(tt[s], rhs, wv[kk][s], 4 * kg + kk);
                                    });
                                }
                            }
                        });
                        __syncthreads();
                        // new iterate, gradient on the active rows, tolerances (dual: |lam| in the solution lanes; primal: |g| by row)
                        double gl[RB];
                        unsigned ghi = 0u;
#pragma unroll
                        for (int s = 0; s < RB; ++s) {
                            const unsigned bit = 1u << rw[s];
                            const bool act = vrow[s] && (mA32 & bit);
                            const double sg = (mL32 & bit) ? -1.0 : 1.0;
                            const double dl = xL[rw[s]];                         // primal: my row's dlt (rows of F)
                            const double nv = act ? sg * h[s] : (dual ? vu[s] - tt[s] : vu[s] + dl);
                            v[s] = busy ? nv : v[s];
                            gl[s] = (act && !dual) ? tt[s] - z[s] : 0.0;
                            ghi = max(ghi, (unsigned)__double2hiint(gl[s]) & 0x7fffffffu);
                        }
                        const unsigned lhi = (unsigned)__double2hiint(rhs) & 0x7fffffffu;
                        const double tol = 1e-10 * __hiloint2double((int)row_umax(dual ? (mine ? lhi : 0u) : ghi), 0);
                        const bool isL = (mL32 >> la) & 1u;
                        const unsigned keepL = (dual && mine && isL && rhs <= tol) ? (1u << la) : 0u;
                        const unsigned keepU = (dual && mine && !isL && rhs >= -tol) ? (1u << la) : 0u;
                        unsigned nL32 = row_or(keepL), nU32 = row_or(keepU);
                        bool nf = false;
#pragma unroll
                        for (int s = 0; s < RB; ++s) {
                            const unsigned bit = 1u << rw[s];
                            const bool act = vrow[s] && (mA32 & bit);
                            const bool fr = vrow[s] && !act;
                            const bool lo = (fr && v[s] < -h[s] * (1.0 + 1e-12)) || (act && !dual && (mL32 & bit) && gl[s] >= -tol);
                            const bool up = (fr && v[s] > h[s] * (1.0 + 1e-12)) || (act && !dual && (mU32 & bit) && gl[s] <= tol);
                            nL32 |= (unsigned)(iballot<LPI>(lo, q) << (LPI * s));
                            nU32 |= (unsigned)(iballot<LPI>(up, q) << (LPI * s));
                            nf = nf || (vrow[s] && !(fabs(v[s]) < 1e300));
                        }
                        const bool rownf = iballot<LPI>(nf || (mine && !(fabs(rhs) < 1e300)), q) != 0;
#ifdef LQMPC_R16_PROF
                        prof_wit += 1; prof_slow += 1; prof_slowt += clock64() - prof_it0;
#endif
                        if (busy) {
                            iters += 1;
                            if (rowfail || rownf) { failed = true; busy = false; }
                            else if ((mask_t)nL32 == mL && (mask_t)nU32 == mU) busy = false;
                            else { mL = nL32; mU = nU32; }
                        }
                        __syncthreads();
                        if (__ballot(busy) == 0ull) break;
                        continue;
                    }
                } else if (!any_primal) {
                    // ---- one instance per wavefront, dual side: the same skeleton with the broadcasts as v_readlane -- everything about
                    // the instance is wave-uniform, so the column indices, the multipliers and the new masks live in scalar registers ----
                    {
                        const mask_t bit = 1ull << rw[0];
                        const bool on = vrow[0] && (mC & bit);
                        const int rk = __popcll(mC & (bit - 1ull));
                        const double sg = (mL & bit) ? -1.0 : 1.0;
                        list[on ? rk : (C::oD - C::oL) * 2] = rw[0];
                        rL[on ? rk : C::oD - C::oR] = __builtin_fma(-sg, h[0], vu[0]);
                    }
                    __syncthreads();
                    const bool mine = i < c;
                    const int la = mine ? list[i] : 0;
                    double rhs = mine ? rL[i] : 0.0;
                    const int tla = PACKED ? la * (la + 1) / 2 : la * LDW;
                    auto widx = [&](int ra, int ta, int lb, int tlb) -> int {
                        if constexpr (PACKED) return max(ta, tlb) + min(ra, lb);
                        else return ta + lb;
                    };
                    double S[CS];
                    static_for<CS / 4>([&](auto bgc) {
                        constexpr int bg = decltype(bgc)::value;
#pragma unroll
                        for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) S[bb] = (bb == i) ? 1.0 : 0.0;
                        if (4 * bg < cw) {
                            static_for<4>([&](auto bc) {
                                constexpr int bb = 4 * bg + decltype(bc)::value;
                                const int lb = __builtin_amdgcn_readlane(la, bb), tlb = PACKED ? __builtin_amdgcn_readlane(tla, bb) : 0;
                                const double val = Wp[widx(la, tla, lb, tlb)];
                                S[bb] = mine ? val : S[bb];
                            });
                        }
                    });
                    bool ok = true;
                    static_for<CS>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if (k < cw) {
                            const double d = ibcast<LPI>(S[k], k);
                            ok = ok && (d > 0.0);
                            const double inv = frcp1(d);
                            const double g = (i == k) ? (inv - 1.0) : -S[k] * inv;
#pragma unroll
                            for (int jg = 0; jg < CS / 4; ++jg)
                                if (4 * jg > k && 4 * jg < cw) ifmac_self4<LPI>(&S[4 * jg], g, k);
                            ifmac_self_tail<LPI, k>(S, rhs, g);
                        }
                    });
                    const bool rowfail = iballot<LPI>(!ok, q) != 0;
                    // t = W_FA lam on my row; the largest multiplier and the bounds that stay, in scalar registers
                    double tt = 0.0;
                    const int trow0 = PACKED ? tri[0] : rw[0] * LDW;
                    unsigned smax = 0u;
                    const unsigned lhi = (unsigned)__double2hiint(rhs) & 0x7fffffffu;
                    static_for<CS>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if (k < cw) {
                            const int lk = __builtin_amdgcn_readlane(la, k), tlk = PACKED ? __builtin_amdgcn_readlane(tla, k) : 0;
                            const double wv = Wp[vrow[0] ? widx(rw[0], trow0, lk, tlk) : 0];
                            tt = __builtin_fma(wg::rdlane(rhs, k), wv, tt);
                            smax = max(smax, (unsigned)__builtin_amdgcn_readlane((int)lhi, k));
                        }
                    });
                    const double tol = 1e-10 * __hiloint2double((int)smax, 0);
                    const bool isL = (mL >> la) & 1ull;
                    const mask_t kL = __ballot(mine && isL && rhs <= tol), kU = __ballot(mine && !isL && rhs >= -tol);   // by unknown k
                    mask_t nL = 0, nU = 0;
                    static_for<CS>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if (k < cw) {
                            const int lk = __builtin_amdgcn_readlane(la, k);
                            nL |= ((kL >> k) & 1ull) << lk;
                            nU |= ((kU >> k) & 1ull) << lk;
                        }
                    });
                    {
                        const mask_t bit = 1ull << rw[0];
                        const bool act = vrow[0] && (mA & bit);
                        const double sg = (mL & bit) ? -1.0 : 1.0;
                        const double nv = act ? sg * h[0] : vu[0] - tt;
                        v[0] = busy ? nv : v[0];
                        const bool fr = vrow[0] && !act;
                        nL |= __ballot(fr && v[0] < -h[0] * (1.0 + 1e-12));
                        nU |= __ballot(fr && v[0] > h[0] * (1.0 + 1e-12));
                    }
                    const bool rownf = __ballot((vrow[0] && !(fabs(v[0]) < 1e300)) || (mine && !(fabs(rhs) < 1e300))) != 0ull;
#ifdef LQMPC_R16_PROF
                    prof_wit += 1; prof_fast += 1; prof_ph[7] += clock64() - prof_it0;
#endif
                    if (busy) {
                        iters += 1;
                        if (rowfail || rownf) { failed = true; busy = false; }
                        else if (nL == mL && nU == mU) busy = false;
                        else { mL = nL; mU = nU; }
                    }
                    __syncthreads();
                    if (__ballot(busy) == 0ull) break;
                    continue;
                }
                // publish v_unc, r = v_unc - s h on the active rows (0 elsewhere), the list of the chosen side
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    const mask_t bit = 1ull << rw[s];
                    const double sg = (mL & bit) ? -1.0 : ((mU & bit) ? 1.0 : 0.0);
                    rL[rw[s]] = (sg != 0.0) ? vu[s] - sg * h[s] : 0.0;
                    xL[rw[s]] = 0.0;
                    if (vrow[s] && (mC & bit)) list[__popcll(mC & (bit - 1ull))] = rw[s];
                }
                __syncthreads();
                const ldsd *Mx = dual ? Wp : Pp;
                const int la = (i < c) ? list[i] : 0, tla = la * (la + 1) / 2;
                double S[CS], rhs;
                if constexpr (LPI == 16) {
                    // the column indices reach the lanes as DPP broadcasts of la: no dependent LDS reads, every load of a group in flight
                    // at once.  Only lanes without an unknown need the identity: columns beyond c meet zeros only (see the dual-only path).
                    const int tla2 = PACKED ? tla : la * LDW;
                    static_for<CS / 4>([&](auto bgc) {
                        constexpr int bg = decltype(bgc)::value;
#pragma unroll
                        for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) S[bb] = (bb == i) ? 1.0 : 0.0;
                        if (4 * bg < cw) {
                            static_for<4>([&](auto bc) {
                                constexpr int bb = 4 * bg + decltype(bc)::value;
                                const int lb = rowb_i(la, bb), tlb = PACKED ? rowb_i(tla2, bb) : 0;
                                int idx;
                                if constexpr (PACKED) idx = max(tla2, tlb) + min(la, lb); else idx = tla2 + lb;
                                const double val = Mx[idx];
                                S[bb] = (i < c) ? val : S[bb];
                            });
                        }
                    });
                } else {
#pragma unroll
                    for (int bg = 0; bg < CS / 4; ++bg) {            // columns in groups of four: one uniform test per group
#pragma unroll
                        for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) S[bb] = (bb == i) ? 1.0 : 0.0;
                        if (4 * bg < cw) {
#pragma unroll
                            for (int bb = 4 * bg; bb < 4 * bg + 4; ++bb) {
                                const int lb = (bb < c) ? list[bb] : 0;
                                const double val = Mx[ad2(la, lb)];
                                if (i < c && bb < c) S[bb] = val;
                            }
                        }
                    }
                }
                rhs = (i < c && dual) ? rL[la] : 0.0;
                if (any_primal) {
                    double tp = 0.0;
#pragma unroll
                    for (int j = 0; j < n; ++j) tp = __builtin_fma(Pp[ad(la, tla, j)], rL[j], tp);
                    if (i < c && !dual) rhs = tp;
                }
                // Gauss-Jordan on [S | rhs]: afterwards S = I and rhs = the solution
                bool ok = true;
                static_for<CS>([&](auto kc) {                    // (compile-time pivot index: the DPP control is an immediate)
                    constexpr int k = decltype(kc)::value;
                    if (k < cw) {                                // uniform
                        isettle<LPI>(S[k]);
                        const double d = ibcast<LPI>(S[k], k);
                        ok = ok && (d > 0.0);
                        const double inv = frcp1(d);
                        const double g = (i == k) ? (inv - 1.0) : -S[k] * inv;
#pragma unroll
                        for (int jg = 0; jg < CS / 4; ++jg)      // columns in groups of four: one uniform test per group (first half static)
                            if (4 * jg > k && 4 * jg < cw) ifmac_self4<LPI>(&S[4 * jg], g, k);
                        ifmac_self_tail<LPI, k>(S, rhs, g);
                    }
                });
                const bool rowfail = iballot<LPI>(!ok, q) != 0;
                if (i < c) xL[la] = rhs;
                __syncthreads();
                double xs_[RB];
#pragma unroll
                for (int s = 0; s < RB; ++s) {                   // y = x (dual side) or x - r (primal side): what the matrix row multiplies
                    xs_[s] = xL[rw[s]];
                    yL[rw[s]] = dual ? xs_[s] : xs_[s] - rL[rw[s]];
                }
                __syncthreads();
                // t = (M y)_row with y = x (dual, M = W) or x - r (primal, M = P); x is zero off the chosen side
                double tol = 0.0, gl[RB];
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    double tt = 0.0;
                    const int mr = vrow[s] ? rw[s] : 0, mt = vrow[s] ? tri[s] : 0;
#pragma unroll
                    for (int j = 0; j < n; ++j) tt = __builtin_fma(Mx[ad(mr, mt, j)], yL[j], tt);
                    const mask_t bit = 1ull << rw[s];
                    const bool act = vrow[s] && (mA & bit);
                    const double sg = (mL & bit) ? -1.0 : 1.0;
                    const double xs = xs_[s];
                    // free rows: the new value; active rows: the bound, and the gradient there (rows that are done keep theirs)
                    const double nv = act ? sg * h[s] : (dual ? vu[s] - tt : vu[s] + xs);
                    v[s] = busy ? nv : v[s];
                    gl[s] = act ? (dual ? -xs : tt) : 0.0;
                    tol = fmax(tol, fabs(gl[s]));
                }
                tol = fmax(tol, __shfl_xor(tol, 1)); tol = fmax(tol, __shfl_xor(tol, 2));
                tol = fmax(tol, __shfl_xor(tol, 4)); tol = fmax(tol, __shfl_xor(tol, 8));
                if (LPI == 64) { tol = fmax(tol, __shfl_xor(tol, 16)); tol = fmax(tol, __shfl_xor(tol, 32)); }
                tol *= 1e-10;
                mask_t nL = 0, nU = 0;
                bool nf = false;
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    const mask_t bit = 1ull << rw[s];
                    const bool act = vrow[s] && (mA & bit);
                    const bool lo = act ? ((mL & bit) && gl[s] >= -tol) : (vrow[s] && v[s] < -h[s] * (1.0 + 1e-12));
                    const bool up = act ? ((mU & bit) && gl[s] <= tol) : (vrow[s] && v[s] > h[s] * (1.0 + 1e-12));
                    nL |= iballot<LPI>(lo, q) << (LPI * s);
                    nU |= iballot<LPI>(up, q) << (LPI * s);
                    nf = nf || (vrow[s] && !(fabs(v[s]) < 1e300));
                }
                const bool rownf = iballot<LPI>(nf, q) != 0;
#ifdef LQMPC_R16_PROF
                prof_wit += 1; prof_slow += 1; prof_slowt += clock64() - prof_it0;
#endif
                if (busy) {
                    iters += 1;
                    if (rowfail || rownf) { failed = true; busy = false; }
                    else if (nL == mL && nU == mU) busy = false;
                    else { mL = nL; mU = nU; }
                }
                __syncthreads();
                if (__ballot(busy) == 0ull) break;
            }
        }
        } while (LAZY_P && need_P);
        if (busy) failed = true;
        if (failed || rowbad) {
            status = 3; pL = 0; pU = 0;
#pragma unroll
            for (int s = 0; s < RB; ++s) v[s] = fmin(fmax(vu[s], -h[s]), h[s]);
        } else {
            pL = mL; pU = mU;
        }
    };
    // u_k of stage st after a solve: clipped v of row st*NU + k plus the centre, from the lane that holds the row
    auto stage_input = [&](const double (&v)[RB], int st, double (&u)[NU]) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int row = st * NU + k;                         // compile-time after unrolling
            const double uk = fmin(fmax(v[row / LPI], -h[row / LPI]), h[row / LPI]) + ctr[row / LPI];
            u[k] = ibcast<LPI>(uk, row % LPI);
        }
    };
    if (MODE != MODE_ROLLOUT) {
        // open loop (utils_class.py:48-91): V_N by rolling the MODEL forward with the optimal inputs
        double Am[NX][NX], Bmm[NX][NU], Pm[NX][NX];
        ldsd *cA = cR + NU * NU, *cB = cA + NX * NX, *cP = cB + NX * NU;
        if constexpr (OCC == 2) {
            if (i == 0) {
#pragma unroll
                for (int a = 0; a < NX; ++a) {
#pragma unroll
                    for (int c = 0; c < NX; ++c) {
                        cA[a * NX + c] = p.rec ? p.rec[bq * REC + a * NX + c] : p.A[(long long)(a * NX + c) * Bsz + bq];
                        cP[a * NX + c] = sh[p.so.P + a * NX + c];
                    }
#pragma unroll
                    for (int k = 0; k < NU; ++k) cB[a * NU + k] = p.rec ? p.rec[bq * REC + NX * NX + a * NU + k] : p.B[(long long)(a * NU + k) * Bsz + bq];
                }
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int a = 0; a < NX; ++a) {
#pragma unroll
                for (int c = 0; c < NX; ++c) {
                    Am[a][c] = p.rec ? p.rec[bq * REC + a * NX + c] : p.A[(long long)(a * NX + c) * Bsz + bq];
                    Pm[a][c] = sh[p.so.P + a * NX + c];
                }
#pragma unroll
                for (int k = 0; k < NU; ++k) Bmm[a][k] = p.rec ? p.rec[bq * REC + NX * NX + a * NU + k] : p.B[(long long)(a * NU + k) * Bsz + bq];
            }
        }
        auto Av = [&](int a, int c) -> double { if constexpr (OCC == 2) return cA[a * NX + c]; else return Am[a][c]; };
        auto Bv = [&](int a, int k) -> double { if constexpr (OCC == 2) return cB[a * NU + k]; else return Bmm[a][k]; };
        auto Pv = [&](int a, int c) -> double { if constexpr (OCC == 2) return cP[a * NX + c]; else return Pm[a][c]; };
        auto value_fn = [&](const double (&x0v)[NX], const double (&v)[RB]) -> double {
            double xs[NX], c = 0.0;
#pragma unroll
            for (int a = 0; a < NX; ++a) xs[a] = x0v[a];
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) c = __builtin_fma(xs[a] * Qv(a, cc), xs[cc], c);
            // the optimal inputs by row through LDS, then a ROLLED loop over the stages (unrolled with one DPP broadcast per input
            // the ten stages of C3 kept ~100 registers in scratch in the one-shot / max-V_N / sweep builds)
#pragma unroll
            for (int s = 0; s < RB; ++s) xL[rw[s]] = vrow[s] ? fmin(fmax(v[s], -h[s]), h[s]) + ctr[s] : 0.0;
            __syncthreads();
#pragma unroll 1
            for (int st = 0; st < N; ++st) {
                double u[NU], xn[NX];
#pragma unroll
                for (int k = 0; k < NU; ++k) u[k] = xL[st * NU + k];
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    double acc = 0.0;
#pragma unroll
                    for (int cc = 0; cc < NX; ++cc) acc = __builtin_fma(Av(a, cc), xs[cc], acc);
#pragma unroll
                    for (int k = 0; k < NU; ++k) acc = __builtin_fma(Bv(a, k), u[k], acc);
                    xn[a] = acc;
                }
#pragma unroll
                for (int a = 0; a < NX; ++a) { xs[a] = xn[a]; if (p.has_ref) xn[a] -= sh[p.so.xref + a * N + st]; }
#pragma unroll
                for (int a = 0; a < NX; ++a)
#pragma unroll
                    for (int cc = 0; cc < NX; ++cc) c = __builtin_fma(xn[a] * ((st == N - 1) ? Pv(a, cc) : Qv(a, cc)), xn[cc], c);
                if (p.has_ref) {
#pragma unroll
                    for (int k = 0; k < NU; ++k) u[k] -= sh[p.so.uref + k * N + st];
                }
#pragma unroll
                for (int k = 0; k < NU; ++k)
#pragma unroll
                    for (int j = 0; j < NU; ++j) c = __builtin_fma(u[k] * Rv(k, j), u[j], c);
            }
            __syncthreads();
            return c;
        };
        if (MODE == MODE_SOLVE) {
            double x[NX], v[RB], u[NU];
#pragma unroll
            for (int a = 0; a < NX; ++a) x[a] = p.rec ? p.rec[bq * REC + NX * NX + NX * NU + a] : p.x0[(long long)a * Bsz + bq];
            cold_armed = ROLL;
            qp(x, v);
            const double vn = value_fn(x, v);
            stage_input(v, 0, u);
            if (writer) {
                p.VN[b] = vn;
#pragma unroll
                for (int k = 0; k < NU; ++k) p.u0[(long long)k * Bsz + bq] = u[k];
            }
        } else {
            double best = -1e308;
            for (int kk = 0; kk < p.K; ++kk) {
                double x[NX], v[RB];
#pragma unroll
                for (int a = 0; a < NX; ++a) x[a] = sh[p.so.x0s + a * p.K + kk];
                pL = 0; pU = 0;
                qp(x, v);
                const double vn = value_fn(x, v);
                best = (vn > best || vn != vn) ? vn : best;
            }
            if (writer) p.MV[b] = best;
        }
    }
    if (MODE == MODE_ROLLOUT || MODE == MODE_SWEEP) {
        pL = 0; pU = 0;
        // closed loop (utils_class.py:266-283).  The plant and the stage cost are spread over the lanes of the instance: lane
        // a < NX owns row a of [A_true B_true] and of Q (its state is broadcast to the others each step), lane k < NU owns row k
        // of R; the partial costs are summed over the lanes once, after the last step.
        double x[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) x[a] = p.rec ? p.rec[bq * REC + NX * NX + NX * NU + a] : p.x0[(long long)a * Bsz + bq];
        int ia = i < NX ? i : 0, ik = i < NU ? i : 0;
        asm volatile("" : "+v"(ia), "+v"(ik));          // (opaque: or these loads are issued at the top and their results spilled across the set-up)
        double Qr_[NX], Ar_[NX], Br_[NU], Rr_[NU];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
            Qr_[c] = sh[p.so.Q + ia * NX + c];
            Ar_[c] = p.true_per_instance ? p.At[(long long)(ia * NX + c) * Bsz + bq] : sh[p.so.At + ia * NX + c];
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            Br_[k] = p.true_per_instance ? p.Bt[(long long)(ia * NU + k) * Bsz + bq] : sh[p.so.Bt + ia * NU + k];
            Rr_[k] = sh[p.so.R + ik * NU + k];
        }
        // my rows: in registers up to 24 of them (C3 0.325 -> 0.318 ms, C4 4.30 -> 4.14 ms: the free steps of a rollout read nothing from
        // LDS); the larger state dimensions of the one-instance-per-wavefront build keep them in the LDS constants region (the open-loop
        // part is done with it) and read them per step
        constexpr bool ROWS_IN_LDS = OCC == 2 && LPI == 64 && 2 * NX + 2 * NU > 12;
        constexpr int RW = 2 * NX + 2 * NU;
        static_assert((NX > NU ? NX : NU) * RW <= C::CN + (C::CN & 1), "the per-lane rows fit the constants region");
        ldsd *myc = L + C::oC + ((i < NX || i < NU) ? i : 0) * RW;
        if constexpr (ROWS_IN_LDS) {
            __syncthreads();
            if (i < NX || i < NU) {
#pragma unroll
                for (int c = 0; c < NX; ++c) { myc[c] = Ar_[c]; myc[NX + NU + c] = Qr_[c]; }
#pragma unroll
                for (int k = 0; k < NU; ++k) { myc[NX + k] = Br_[k]; myc[2 * NX + NU + k] = Rr_[k]; }
            }
            __syncthreads();
        }
        auto Ar = [&](int c) -> double { if constexpr (ROWS_IN_LDS) return myc[c]; else return Ar_[c]; };
        auto Br = [&](int k) -> double { if constexpr (ROWS_IN_LDS) return myc[NX + k]; else return Br_[k]; };
        auto Qr = [&](int c) -> double { if constexpr (ROWS_IN_LDS) return myc[NX + NU + c]; else return Qr_[c]; };
        auto Rr = [&](int k) -> double { if constexpr (ROWS_IN_LDS) return myc[2 * NX + NU + k]; else return Rr_[k]; };
        double costx, costu = 0.0;
        {
            double xm = x[0], qx = 0.0;
#pragma unroll
            for (int a = 1; a < NX; ++a) xm = (i == a) ? x[a] : xm;
#pragma unroll
            for (int c = 0; c < NX; ++c) qx = __builtin_fma(Qr(c), x[c], qx);
            costx = xm * qx;
        }
        if (p.X && writer) {
#pragma unroll
            for (int a = 0; a < NX; ++a) p.X[((long long)a * (p.T + 1)) * Bsz + bq] = x[a];
        }
#ifdef LQMPC_R16_PROF
        long long prof_ts = 0, prof_acc[5] = {0, 0, 0, 0, 0};
#endif
        for (int t = 0; t < p.T; ++t) {
            double v[RB], u[NU], xn[NX];
#ifdef LQMPC_R16_PROF
            if (t == 1) { RPROF_ADD(27, clock64() - prof_ts); RPROF_ADD(28, prof_wit); RPROF_ADD(29, prof_slow); }
            if (t > 0) {
                const long long d_ = clock64() - prof_ts;
                if (prof_wit) { prof_acc[1] += d_; prof_acc[3] += 1; prof_acc[4] += prof_wit; } else { prof_acc[0] += d_; prof_acc[2] += 1; }
            }
            prof_ts = clock64();
            prof_wit = 0;
#endif
            // most steps of most wavefronts: the unconstrained minimiser of every instance of the wavefront is inside its box --
            // one compare per row slot and one wave-wide test, none of the mask bookkeeping of qp()
            bool inside = false;
            {
                bool out = false;
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    double acc = vr[s];
#pragma unroll
                    for (int a = 0; a < NX; ++a) acc = __builtin_fma(G[s][a], x[a], acc);
                    v[s] = acc;
                    out = out || (vrow[s] && !(fabs(acc) <= h[s]));       // (not <=: a NaN counts as outside and takes the general path)
                }
                inside = __ballot(out) == 0ull;
            }
            cold_armed = ROLL && t == 0;
            if (inside) { pL = 0; pU = 0; }
            else qp(x, v);
            const double um = fmin(fmax(v[0], -h[0]), h[0]) + ctr[0];     // the input of my first row: u_i in lane i < NU
            stage_input(v, 0, u);
            double xm = 0.0, qx = 0.0, ru = 0.0;
#pragma unroll
            for (int c = 0; c < NX; ++c) xm = __builtin_fma(Ar(c), x[c], xm);
#pragma unroll
            for (int k = 0; k < NU; ++k) xm = __builtin_fma(Br(k), u[k], xm);
#pragma unroll
            for (int a = 0; a < NX; ++a) xn[a] = ibcast<LPI>(xm, a);
#pragma unroll
            for (int a = 0; a < NX; ++a) x[a] = xn[a];
#pragma unroll
            for (int c = 0; c < NX; ++c) qx = __builtin_fma(Qr(c), xn[c], qx);
            costx = __builtin_fma(xm, qx, costx);
#pragma unroll
            for (int j = 0; j < NU; ++j) ru = __builtin_fma(Rr(j), u[j], ru);
            costu = __builtin_fma(um, ru, costu);
            if (writer) {
                if (p.X) {
#pragma unroll
                    for (int a = 0; a < NX; ++a) p.X[((long long)a * (p.T + 1) + t + 1) * Bsz + bq] = xn[a];
                }
                if (p.U) {
#pragma unroll
                    for (int k = 0; k < NU; ++k) p.U[((long long)k * p.T + t) * Bsz + bq] = u[k];
                }
            }
        }
        double cost = 0.0;
#pragma unroll
        for (int a = 0; a < NX; ++a) cost += ibcast<LPI>(costx, a);
#pragma unroll
        for (int k = 0; k < NU; ++k) cost += ibcast<LPI>(costu, k);
        RPROF(7);
#ifdef LQMPC_R16_PROF
        for (int k_ = 0; k_ < 5; ++k_) RPROF_ADD(8 + k_, prof_acc[k_]);
#endif
        RPROF_ADD(14, clock64() - prof_t0); RPROF_ADD(15, 1); RPROF_ADD(16, prof_slow); RPROF_ADD(30, prof_slow > 0 ? 1 : 0);
#ifdef LQMPC_R16_PROF
        for (int k_ = 0; k_ < 8; ++k_) RPROF_ADD(17 + k_, prof_ph[k_]);
        RPROF_ADD(25, prof_slowt); RPROF_ADD(26, prof_fast);
#endif
        if (writer) p.JT[b] = cost;
    }
    if (writer) {
        if (p.status) p.status[b] = status;
        if (p.iters) p.iters[b] = iters;
        if (status == 3 && p.fail_list) p.fail_list[atomicAdd(p.fail_count, 1)] = (int)b;
    }
}

}  // namespace lqmpc
