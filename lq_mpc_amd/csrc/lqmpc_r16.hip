// lqmpc_r16.hip -- closed-loop rollouts for n = N*nu <= 32: one instance per 16-lane row of a wavefront (four
// per wave), matrix row i of an instance in lane i % 16 (slot i / 16).
//
// Why another layout.  The packed register-resident kernel (lqmpc_spec.hip) factors the masked n x n matrix in
// every active-set iteration: ~4000 instructions per iteration and wave, one wave per SIMD (512 registers, 40
// KiB of LDS), so the launch lasts as long as the wave holding the instances that stay constrained for all T
// steps (C3: 78 iterations, 1.1 ms).  Here an iteration costs a few hundred instructions because it only ever
// solves the SMALLER side of the active-set system:
//   with A the active set (signs s), F the free set, r_A = v_unc,A - s h_A and W = P^-1,
//     |A| <= |F| (dual side):    W_AA lam = r_A,          v_F = v_unc,F - W_FA lam,          gradient on A = -lam
//     |A| >  |F| (primal side):  P_FF dlt = P_FA r_A,     v_F = v_unc,F + dlt,               gradient on A = P_AF dlt - P_AA r_A
//   both are the same iterate of the primal-dual active-set method the other kernels use; min(|A|, |F|) <= 16.
// P and W live packed in LDS (3.4 KB per instance at n = 20), the gathered system is solved in registers by
// Gauss-Jordan elimination with DPP row broadcasts (v_fmac_f64_dpp row_newbcast: the pivot row reaches the 16
// lanes of an instance without leaving the vector registers).  W itself comes from P by the same elimination.
// Register and LDS needs are small enough for two to three waves per SIMD.
//
// An instance whose active set does not settle within MAXIT iterations is reported with status 3; the host
// re-runs exactly those instances on the packed kernel (interior point + active-set finishing).
#include "lqmpc_wg_linalg.h"
#include <cstdio>

namespace lqmpc {

using wg::ldsd;
using wg::ldsi;
using wg::rowb;
using wg::fmac_rowb;

template <int NX, int NU, int N>
struct R16 {
    static constexpr int n = N * NU;
    static constexpr int RB = (n + 15) / 16;
    static constexpr int PK = n * (n + 1) / 2;
    static constexpr int VEC = 16 * RB;
    // LDS per instance, in doubles: P packed | W packed | vu | r | w | list (16 ints)
    static constexpr int oP = 0, oW = PK, oVU = 2 * PK, oR = oVU + VEC, oX = oR + VEC, oL = oX + VEC;
    static constexpr int INST = oL + 8;
    static constexpr int MAXIT = 12;
    static_assert(n <= 32, "one or two row slots per lane");
    static_assert(N * NX * NU <= PK, "the A^m B table aliases the W region while condensing");
};

__device__ __forceinline__ int pidx(int i, int j) { return i * (i + 1) / 2 + j; }                  // i >= j
__device__ __forceinline__ int sidx(int i, int j) { return i >= j ? pidx(i, j) : pidx(j, i); }

// the 16 bits of a wave ballot that belong to my 16-lane row
__device__ __forceinline__ unsigned ballot16(bool c, int q) { return (unsigned)((__ballot(c) >> (16 * q)) & 0xFFFFull); }

template <int NX, int NU, int N>
__global__ void __launch_bounds__(64, 1) lqmpc_r16_kernel(KParams p)
{
    using C = R16<NX, NU, N>;
    constexpr int n = C::n, RB = C::RB;
    constexpr int REC = NX * NX + NX * NU + NX;
    __shared__ double lds_raw[4 * C::INST];
    const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
    ldsd *L = (ldsd *)lds_raw + q * C::INST;
    ldsd *Pp = L + C::oP, *Wp = L + C::oW, *vuL = L + C::oVU, *rL = L + C::oR, *xL = L + C::oX;
    ldsi *list = (ldsi *)(L + C::oL);
    const long long Bsz = p.Bsz;
    const long long b_raw = (long long)blockIdx.x * 4 + q;
    const bool valid = b_raw < Bsz;
    const long long slot = valid ? b_raw : Bsz - 1;
    const long long b = p.perm ? (long long)p.perm[slot] : slot;
    const double *sh = p.sh;
    const unsigned nmask = (n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);

    int rw[RB];
    bool vrow[RB];
    double h[RB], ctr[RB];
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        rw[s] = i + 16 * s;
        vrow[s] = rw[s] < n;
        const int k = rw[s] % NU;
        h[s] = vrow[s] ? 0.5 * (sh[p.so.ub + k] - sh[p.so.lb + k]) : 1.0;
        ctr[s] = vrow[s] ? 0.5 * (sh[p.so.ub + k] + sh[p.so.lb + k]) : 0.0;
    }

    // ---------------- condensing (same accumulation as lqmpc_spec.hip; utils_class.py:62-75) ----------------
    double G[RB][NX], vr[RB];
    {
        double A[NX][NX], Bm[NX][NU];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
#pragma unroll
            for (int c = 0; c < NX; ++c) A[a][c] = p.rec ? p.rec[b * REC + a * NX + c] : p.A[(long long)(a * NX + c) * Bsz + b];
#pragma unroll
            for (int k = 0; k < NU; ++k) Bm[a][k] = p.rec ? p.rec[b * REC + NX * NX + a * NU + k] : p.B[(long long)(a * NU + k) * Bsz + b];
        }
        // M[m] = A^m B staged at Wp[(m*NU + k)*NX + a]; every lane runs the chain, lane e % 16 stores element e
        {
            double Mc[NX][NU];
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int k = 0; k < NU; ++k) Mc[a][k] = Bm[a][k];
#pragma unroll
            for (int m = 0; m < N; ++m) {
                if (m > 0) {
                    double T[NX][NU];
#pragma unroll
                    for (int a = 0; a < NX; ++a)
#pragma unroll
                        for (int k = 0; k < NU; ++k) {
                            double t = 0.0;
#pragma unroll
                            for (int c = 0; c < NX; ++c) t = __builtin_fma(A[a][c], Mc[c][k], t);
                            T[a][k] = t;
                        }
#pragma unroll
                    for (int a = 0; a < NX; ++a)
#pragma unroll
                        for (int k = 0; k < NU; ++k) Mc[a][k] = T[a][k];
                }
#pragma unroll
                for (int k = 0; k < NU; ++k)
#pragma unroll
                    for (int a = 0; a < NX; ++a) {
                        const int e = (m * NU + k) * NX + a;
                        if ((e & 15) == i) Wp[e] = Mc[a][k];
                    }
            }
        }
        __syncthreads();
        double Pacc[RB][n], Facc[RB][NX], qacc[RB];
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            qacc[s] = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) Pacc[s][j] = 0.0;
#pragma unroll
            for (int a = 0; a < NX; ++a) Facc[s][a] = 0.0;
        }
        bool has_lin = p.has_ref != 0;
#pragma unroll
        for (int k = 0; k < NU; ++k) has_lin = has_lin || (sh[p.so.ub + k] + sh[p.so.lb + k] != 0.0);
        double Ap[NX][NX], sc[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            sc[a] = 0.0;
#pragma unroll
            for (int c = 0; c < NX; ++c) Ap[a][c] = (a == c) ? 1.0 : 0.0;
        }
#pragma unroll 1
        for (int rt = 0; rt < N; ++rt) {
            {
                double T[NX][NX];
#pragma unroll
                for (int a = 0; a < NX; ++a)
#pragma unroll
                    for (int c = 0; c < NX; ++c) {
                        double t = 0.0;
#pragma unroll
                        for (int l = 0; l < NX; ++l) t = __builtin_fma(A[a][l], Ap[l][c], t);
                        T[a][c] = t;
                    }
#pragma unroll
                for (int a = 0; a < NX; ++a)
#pragma unroll
                    for (int c = 0; c < NX; ++c) Ap[a][c] = T[a][c];
            }
            double e[NX];
            if (has_lin) {
                double T[NX];
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    double t = 0.0;
#pragma unroll
                    for (int c = 0; c < NX; ++c) t = __builtin_fma(A[a][c], sc[c], t);
#pragma unroll
                    for (int k = 0; k < NU; ++k) t = __builtin_fma(Bm[a][k], 0.5 * (sh[p.so.ub + k] + sh[p.so.lb + k]), t);
                    T[a] = t;
                }
#pragma unroll
                for (int a = 0; a < NX; ++a) { sc[a] = T[a]; e[a] = T[a] - (p.has_ref ? sh[p.so.xref + a * N + rt] : 0.0); }
            }
            const int oQ = (rt < N - 1) ? p.so.Q : p.so.P;
            double w[RB][NX];
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const int bi = rw[s] / NU, ui = rw[s] % NU;
                const int m = rt - bi, mc = (m < 0 || !vrow[s]) ? 0 : m;
                double g[NX];
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    const double t = Wp[(mc * NU + ui) * NX + a];
                    g[a] = (m < 0 || !vrow[s]) ? 0.0 : t;
                }
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    double t = 0.0;
#pragma unroll
                    for (int c = 0; c < NX; ++c) t = __builtin_fma(sh[oQ + a * NX + c], g[c], t);
                    w[s][a] = t;
                }
#pragma unroll
                for (int aa = 0; aa < NX; ++aa) {
                    double t = Facc[s][aa];
#pragma unroll
                    for (int a = 0; a < NX; ++a) t = __builtin_fma(w[s][a], Ap[a][aa], t);
                    Facc[s][aa] = t;
                }
                if (has_lin) {
                    double t = qacc[s];
#pragma unroll
                    for (int a = 0; a < NX; ++a) t = __builtin_fma(w[s][a], e[a], t);
                    qacc[s] = t;
                }
            }
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const int bj = j / NU, uj = j % NU;
                if (bj > rt) continue;       // wave-uniform
                double mcol[NX];
#pragma unroll
                for (int a = 0; a < NX; ++a) mcol[a] = Wp[((rt - bj) * NU + uj) * NX + a];
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    double t = Pacc[s][j];
#pragma unroll
                    for (int a = 0; a < NX; ++a) t = __builtin_fma(w[s][a], mcol[a], t);
                    Pacc[s][j] = t;
                }
            }
        }
        __syncthreads();                       // the A^m B table (in the W region) is dead from here
        // P = 2 (H + Rbar): full rows in registers (identity on the padding rows), lower triangle to LDS
        double qr[RB];
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int bi = rw[s] / NU, ui = rw[s] % NU;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const int bj = j / NU, uj = j % NU;
                const double val = 2.0 * (Pacc[s][j] + ((bj == bi) ? sh[p.so.R + ui * NU + uj] : 0.0));
                Pacc[s][j] = vrow[s] ? val : ((j == rw[s]) ? 1.0 : 0.0);
                if (vrow[s] && j <= rw[s]) Pp[pidx(rw[s], j)] = val;
            }
            double tq = qacc[s];
            if (has_lin && vrow[s]) {
#pragma unroll
                for (int uj = 0; uj < NU; ++uj) {
                    const double cu = 0.5 * (sh[p.so.ub + uj] + sh[p.so.lb + uj]);
                    const double ur = p.has_ref ? sh[p.so.uref + uj * N + bi] : 0.0;
                    tq = __builtin_fma(sh[p.so.R + ui * NU + uj], cu - ur, tq);
                }
            }
            qr[s] = vrow[s] ? 2.0 * tq : 0.0;
#pragma unroll
            for (int a = 0; a < NX; ++a) Facc[s][a] = vrow[s] ? 2.0 * Facc[s][a] : 0.0;
        }
        // W = P^-1 by Gauss-Jordan elimination in place: per pivot k the pivot row reaches every lane through
        // the DPP operand of the update, row_i += g_i * row_k with g_i = -a_ik / a_kk (g_k = 1/a_kk - 1 scales
        // the pivot row itself); column k is set to e_k first so that it ends up holding column k of the inverse.
        bool spd = true;
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const int sk = k / 16, lk = k % 16;
            const double d = rowb(Pacc[sk][k], lk);
            spd = spd && (d > 0.0);
            const double inv = frcp(d);
            double g[RB];
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const bool isk = (rw[s] == k);
                g[s] = isk ? (inv - 1.0) : -Pacc[s][k] * inv;
                Pacc[s][k] = isk ? 1.0 : 0.0;
            }
#pragma unroll
            for (int j = 0; j < n; ++j) {
#pragma unroll
                for (int s = 0; s < RB; ++s)
                    if (s != sk) fmac_rowb(Pacc[s][j], Pacc[sk][j], g[s], lk);     // the pivot row's own slot last:
                fmac_rowb(Pacc[sk][j], Pacc[sk][j], g[sk], lk);                    // it rescales the row the others read
            }
        }
        // [G | v_r] = -W [Fq | qr]: row i of W is in my registers, row j of [Fq | qr] comes by row broadcast
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            vr[s] = 0.0;
#pragma unroll
            for (int a = 0; a < NX; ++a) G[s][a] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const int sj = j / 16, lj = j % 16;
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const double nw = -Pacc[s][j];
#pragma unroll
                for (int a = 0; a < NX; ++a) fmac_rowb(G[s][a], Facc[sj][a], nw, lj);
                if (has_lin) fmac_rowb(vr[s], qr[sj], nw, lj);
            }
        }
#pragma unroll
        for (int s = 0; s < RB; ++s) {
#pragma unroll
            for (int j = 0; j < n; ++j)
                if (vrow[s] && j <= rw[s]) Wp[pidx(rw[s], j)] = Pacc[s][j];
            if (!vrow[s]) {
                vr[s] = 0.0;
#pragma unroll
                for (int a = 0; a < NX; ++a) G[s][a] = 0.0;
            }
        }
        if (!spd) {
#pragma unroll
            for (int s = 0; s < RB; ++s) vr[s] = __builtin_nan("");
        }
        __syncthreads();
    }

    // ---------------- closed loop (utils_class.py:266-283) ----------------
    double x[NX];
#pragma unroll
    for (int a = 0; a < NX; ++a) x[a] = p.rec ? p.rec[b * REC + NX * NX + NX * NU + a] : p.x0[(long long)a * Bsz + b];
    double cost = 0.0;
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) cost = __builtin_fma(x[a] * sh[p.so.Q + a * NX + c], x[c], cost);
    const bool writer = valid && i == 0;
    if (p.X && writer) {
#pragma unroll
        for (int a = 0; a < NX; ++a) p.X[((long long)a * (p.T + 1)) * Bsz + b] = x[a];
    }
    unsigned pL = 0, pU = 0;                  // active set of the previous step (row-uniform bit masks)
    int iters = 0, status = 0;
    for (int t = 0; t < p.T; ++t) {
        double vu[RB], v[RB];
        unsigned cl = 0, cu = 0;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            double acc = vr[s];
#pragma unroll
            for (int a = 0; a < NX; ++a) acc = __builtin_fma(G[s][a], x[a], acc);
            vu[s] = acc; v[s] = acc;
            cl |= ballot16(vrow[s] && acc < -h[s], q) << (16 * s);
            cu |= ballot16(vrow[s] && acc > h[s], q) << (16 * s);
            bad = bad || (vrow[s] && !(fabs(acc) < 1e300));
        }
        const bool rowbad = ballot16(bad, q) != 0;
        bool busy = ((cl | cu) != 0) && !rowbad;              // row-uniform
        unsigned mL = 0, mU = 0;
        if (busy) {
            if (p.warm_start && p.max_iter != 52 && (pL | pU) != 0) {
                // the previous face shifted by one stage; the last stage keeps its flags
                const unsigned top = nmask & ~(nmask >> NU);
                mL = (pL >> NU) | (pL & top);
                mU = (pU >> NU) | (pU & top);
            } else {
                mL = cl; mU = cu;
            }
        }
        bool failed = false;
        if (__ballot(busy) != 0ull) {
#pragma unroll 1
            for (int it = 0; it < C::MAXIT; ++it) {
                const unsigned mA = mL | mU;
                const int m = __popc(mA);
                const bool dual = (p.max_iter == 51) ? (m <= 16) : ((p.max_iter == 53) ? (n - m > 16) : (2 * m <= n));   // DEBUG knobs
                const unsigned mC = busy ? (dual ? mA : (~mA & nmask)) : 0u;
                const int c = __popc(mC);
                int cw = c;                                      // wave maximum: uniform loop bound
                cw = max(cw, __shfl_xor(cw, 16)); cw = max(cw, __shfl_xor(cw, 32));
                cw = __builtin_amdgcn_readfirstlane(cw);
                const bool any_primal = __ballot(busy && !dual) != 0ull;
                // publish v_unc, r = v_unc - s h on the active rows (0 elsewhere), the list of the chosen side
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    const unsigned bit = 1u << rw[s];
                    const double sg = (mL & bit) ? -1.0 : ((mU & bit) ? 1.0 : 0.0);
                    vuL[rw[s]] = vu[s];
                    rL[rw[s]] = (sg != 0.0) ? vu[s] - sg * h[s] : 0.0;
                    xL[rw[s]] = 0.0;
                    if (vrow[s] && (mC & bit)) list[__popc(mC & (bit - 1u))] = rw[s];
                }
                __syncthreads();
                const ldsd *Mx = dual ? Wp : Pp;
                const int la = (i < c) ? list[i] : 0;
                double S[16], rhs;
#pragma unroll
                for (int bb = 0; bb < 16; ++bb) {
                    S[bb] = (bb == i) ? 1.0 : 0.0;
                    if (bb < cw) {                               // uniform
                        const int lb = (bb < c) ? list[bb] : 0;
                        const double val = Mx[sidx(la, lb)];
                        if (i < c && bb < c) S[bb] = val;
                    }
                }
                rhs = (i < c && dual) ? rL[la] : 0.0;
                if (any_primal) {
                    double tp = 0.0;
#pragma unroll
                    for (int j = 0; j < n; ++j) tp = __builtin_fma(Pp[sidx(la, j)], rL[j], tp);
                    if (i < c && !dual) rhs = tp;
                }
                // Gauss-Jordan on [S | rhs]: afterwards S = I and rhs = the solution
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < cw) {                                // uniform
                        const double d = rowb(S[k], k);
                        ok = ok && (d > 0.0);
                        const double inv = frcp(d);
                        const double g = (i == k) ? (inv - 1.0) : -S[k] * inv;
#pragma unroll
                        for (int j = k + 1; j < 16; ++j)
                            if (j < cw) fmac_rowb(S[j], S[j], g, k);
                        fmac_rowb(rhs, rhs, g, k);
                    }
                }
                const bool rowfail = ballot16(!ok, q) != 0;
                if (i < c) xL[la] = rhs;
                __syncthreads();
                // t = (M y)_row with y = x (dual, M = W) or x - r (primal, M = P); x is zero off the chosen side
                double tol = 0.0, gl[RB];
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        const double y = dual ? xL[j] : xL[j] - rL[j];
                        tt = __builtin_fma(Mx[sidx(vrow[s] ? rw[s] : 0, j)], y, tt);
                    }
                    const unsigned bit = 1u << rw[s];
                    const bool act = vrow[s] && (mA & bit);
                    const double sg = (mL & bit) ? -1.0 : 1.0;
                    const double xs = xL[rw[s]];
                    // free rows: the new value; active rows: the bound, and the gradient there (rows that are done keep theirs)
                    const double nv = act ? sg * h[s] : (dual ? vu[s] - tt : vu[s] + xs);
                    v[s] = busy ? nv : v[s];
                    gl[s] = act ? (dual ? -xs : tt) : 0.0;
                    tol = fmax(tol, fabs(gl[s]));
                }
                tol = fmax(tol, __shfl_xor(tol, 1)); tol = fmax(tol, __shfl_xor(tol, 2));
                tol = fmax(tol, __shfl_xor(tol, 4)); tol = fmax(tol, __shfl_xor(tol, 8));
                tol *= 1e-10;
                unsigned nL = 0, nU = 0;
                bool nf = false;
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    const unsigned bit = 1u << rw[s];
                    const bool act = vrow[s] && (mA & bit);
                    const bool lo = act ? ((mL & bit) && gl[s] >= -tol) : (vrow[s] && v[s] < -h[s] * (1.0 + 1e-12));
                    const bool up = act ? ((mU & bit) && gl[s] <= tol) : (vrow[s] && v[s] > h[s] * (1.0 + 1e-12));
                    nL |= ballot16(lo, q) << (16 * s);
                    nU |= ballot16(up, q) << (16 * s);
                    nf = nf || (vrow[s] && !(fabs(v[s]) < 1e300));
                }
                const bool rownf = ballot16(nf, q) != 0;
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
        if (busy) failed = true;
        if (failed || rowbad) {
            status = 3; pL = 0; pU = 0;
#pragma unroll
            for (int s = 0; s < RB; ++s) v[s] = fmin(fmax(vu[s], -h[s]), h[s]);
        } else {
            pL = mL; pU = mU;
        }
        // u_k = clipped v of row k + centre, from lane k of the row
        double u[NU], xn[NX];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const double uk = fmin(fmax(v[k / 16], -h[k / 16]), h[k / 16]) + ctr[k / 16];
            u[k] = rowb(uk, k % 16);
        }
        if (p.true_per_instance) {
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < NX; ++c) acc = __builtin_fma(p.At[(long long)(a * NX + c) * Bsz + b], x[c], acc);
#pragma unroll
                for (int k = 0; k < NU; ++k) acc = __builtin_fma(p.Bt[(long long)(a * NU + k) * Bsz + b], u[k], acc);
                xn[a] = acc;
            }
        } else {
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < NX; ++c) acc = __builtin_fma(sh[p.so.At + a * NX + c], x[c], acc);
#pragma unroll
                for (int k = 0; k < NU; ++k) acc = __builtin_fma(sh[p.so.Bt + a * NU + k], u[k], acc);
                xn[a] = acc;
            }
        }
#pragma unroll
        for (int a = 0; a < NX; ++a) x[a] = xn[a];
#pragma unroll
        for (int a = 0; a < NX; ++a)
#pragma unroll
            for (int c = 0; c < NX; ++c) cost = __builtin_fma(xn[a] * sh[p.so.Q + a * NX + c], xn[c], cost);
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int j = 0; j < NU; ++j) cost = __builtin_fma(u[k] * sh[p.so.R + k * NU + j], u[j], cost);
        if (writer) {
            if (p.X) {
#pragma unroll
                for (int a = 0; a < NX; ++a) p.X[((long long)a * (p.T + 1) + t + 1) * Bsz + b] = xn[a];
            }
            if (p.U) {
#pragma unroll
                for (int k = 0; k < NU; ++k) p.U[((long long)k * p.T + t) * Bsz + b] = u[k];
            }
        }
    }
    if (writer) {
        p.JT[b] = cost;
        if (p.status) p.status[b] = status;
        if (p.iters) p.iters[b] = iters;
        if (status == 3 && p.fail_list) p.fail_list[atomicAdd(p.fail_count, 1)] = (int)b;
    }
}

struct R16Entry {
    int nx, nu, N;
    const char *name;
    void (*launch)(const KParams &, hipStream_t);
};

template <int NX, int NU, int N>
static void launch_r16_one(const KParams &p, hipStream_t stream)
{
    hipLaunchKernelGGL((lqmpc_r16_kernel<NX, NU, N>), dim3((unsigned)((p.Bsz + 3) / 4)), dim3(64), 0, stream, p);
}

#define R16E(NX, NU, N) {NX, NU, N, "lqmpc_r16_kernel<" #NX "," #NU "," #N ">", launch_r16_one<NX, NU, N>}
static const R16Entry g_r16[] = {
    R16E(4, 2, 10),     // C3 (headline)
};

static const R16Entry *find_r16(int nx, int nu, int N)
{
    for (const R16Entry &e : g_r16)
        if (e.nx == nx && e.nu == nu && e.N == N) return &e;
    return nullptr;
}

bool r16_available(int nx, int nu, int N) { return find_r16(nx, nu, N) != nullptr; }

bool launch_r16(const KParams &p, hipStream_t stream, const char **name)
{
    const R16Entry *e = find_r16(p.nx, p.nu, p.N);
    if (!e) return false;
    e->launch(p, stream);
    if (name) *name = e->name;
    return true;
}

}  // namespace lqmpc
