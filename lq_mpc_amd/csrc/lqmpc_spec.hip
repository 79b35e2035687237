// lqmpc_spec.hip -- register-resident, lane-cooperative kernels for the BASELINE shapes (the throughput path).
//
// Mapping (gfx950, wave64, one wave per 64-thread workgroup, one wave per SIMD):
//   LPS lanes cooperate on one (A,B) instance ("group"); a wave holds 64/LPS instances.  Row i of the
//   condensed n x n system belongs to lane r = i % LPS of its group, as row-block jb = i / LPS.
//   LPS = 1, 2, 4 (DPP quad_perm groups, n a multiple of LPS, n <= 32) or 64 (one row per lane, n <= 64).
//   Every loop over rows / columns is fully unrolled so that each register index is a compile-time constant.
// Storage:
//   VGPRs  the working matrix K -> L of the in-place Cholesky (own rows) and the solver vectors;
//   AGPRs  the constant rows of P = 2(Gamma'Qbar Gamma + Rbar), pinned with the "a" constraint (AccArr);
//   LDS    a column-major XOR-swizzled mirror of L (backward substitution, Cholesky broadcasts), the rows of
//          Fq = 2 Gamma'Qbar Phi (or G = -P^-1 Fq with the presolve), and A^m B while condensing.
// Cross-lane traffic: v_mov_b32_dpp / v_readlane for the latency-critical values, group-uniform ds_read_b128
//   for the rest.  HBM sees the inputs (A, B, x0) and the results, nothing else.
// Algorithm (Spec::solve_qp): presolve -> primal-dual active-set warm start -> staged Mehrotra
//   predictor-corrector interior point with active-set finishing.
//
// Sorted rollouts run in two tiers (lqmpc_spec_tiered_kernel below): the hardest instances of the order use the
// 16-lane-row layout of lqmpc_r16_body.h, everything else the packed layout described here.
//
// Restates /root/reference/utils_class.py:48-91 (solve) and 245-285 (simulate); the solver replaces cvxpy's
// QP back-end (utils_class.py:84-88).
#include <cstdio>
#include "lqmpc_common.h"
#include "lqmpc_r16_body.h"
#include "lqmpc_probe.h"
#include <cstdlib>

#include <type_traits>

namespace lqmpc {

// ---------------- cross-lane primitives inside a group of LPS lanes ----------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // every source lane of a quad_perm is valid, so the destination's old value is never kept:
    // mov_dpp (old = undef) lets the compiler read the source register in place (no copy, no s_nop)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov_u32(unsigned x)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true);
}

// value of sub-lane SRC of my group, in every lane of the group
template <int LPS, int SRC>
__device__ __forceinline__ double bcast(double x)
{
    static_assert(LPS == 1 || LPS == 2 || LPS == 4 || LPS == 64, "groups of 1, 2, 4 lanes (quad_perm) or the whole wave");
    if constexpr (LPS == 1) return x;
    else if constexpr (LPS == 2) return dpp_mov<(SRC == 0) ? 0xA0 : 0xF5>(x);   // [0,0,2,2] / [1,1,3,3]
    else if constexpr (LPS == 4) return dpp_mov<SRC * 0x55>(x);                  // [s,s,s,s]
    else {
        // whole wave: v_readlane_b32 puts lane SRC's value in SGPRs, i.e. the broadcast is a scalar operand
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(x), SRC);
        return __hiloint2double(hi, lo);
    }
}

// wave-wide reductions (LPS == 64): quad_perm x2, row_half_mirror, row_mirror leave every lane with its
// 16-lane row's result; row_bcast15 / row_bcast31 carry it across rows into lane 63, which is broadcast.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_rows(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);   // lanes outside ROWMASK keep x
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_reduce(double x)
{
#define LQ_OP(a, b) (MAX ? fmax((a), (b)) : (a) + (b))
    x = LQ_OP(x, dpp_mov<0xB1>(x));
    x = LQ_OP(x, dpp_mov<0x4E>(x));
    x = LQ_OP(x, (dpp_rows<0x141, 0xF>(x)));                      // row_half_mirror
    x = LQ_OP(x, (dpp_rows<0x140, 0xF>(x)));                      // row_mirror
    const int row = threadIdx.x >> 4;
    { const double y = dpp_rows<0x142, 0xA>(x); if (row & 1) x = LQ_OP(x, y); }   // row_bcast15 into rows 1, 3
    { const double y = dpp_rows<0x143, 0xC>(x); if (row & 2) x = LQ_OP(x, y); }   // row_bcast31 into rows 2, 3
#undef LQ_OP
    return bcast<64, 63>(x);
}

template <int LPS>
__device__ __forceinline__ double group_sum(double x)
{
    if constexpr (LPS == 64) return wave_reduce<false>(x);
    if constexpr (LPS >= 2) x += dpp_mov<0xB1>(x);   // [1,0,3,2]
    if constexpr (LPS >= 4) x += dpp_mov<0x4E>(x);   // [2,3,0,1]
    return x;
}
template <int LPS>
__device__ __forceinline__ double group_max(double x)
{
    if constexpr (LPS == 64) return wave_reduce<true>(x);
    if constexpr (LPS >= 2) x = fmax(x, dpp_mov<0xB1>(x));
    if constexpr (LPS >= 4) x = fmax(x, dpp_mov<0x4E>(x));
    return x;
}
template <int LPS, typename M>
__device__ __forceinline__ M group_or(M x)
{
    if constexpr (LPS == 64) {
        // one row per lane: a lane's mask can only hold its own bit, so the union is a ballot
        return (M)__ballot(x != 0);
    } else {
        unsigned y = (unsigned)x;
        if constexpr (LPS >= 2) y |= dpp_mov_u32<0xB1>(y);
        if constexpr (LPS >= 4) y |= dpp_mov_u32<0x4E>(y);
        return (M)y;
    }
}
template <int LPS>
__device__ __forceinline__ unsigned group_and(unsigned x)     // x in {0, 1}
{
    if constexpr (LPS == 64) return __all(x != 0) ? 1u : 0u;
    if constexpr (LPS >= 2) x &= dpp_mov_u32<0xB1>(x);
    if constexpr (LPS >= 4) x &= dpp_mov_u32<0x4E>(x);
    return x;
}

// ---------------- read-mostly per-lane data parked in the accumulator register file ----------------
// gfx950 has one 512-entry register file per lane, but VALU instructions address only the first
// 256 (VGPRs); the upper half (AGPRs) is reachable with v_accvgpr_read/write.  The constant rows
// of P and Fq are read once per interior-point iteration / once per QP, so they are pinned there
// explicitly ("a" constraint) and the allocator keeps the VGPR half for the Cholesky working set.
template <int CNT>
struct AccArr {
    int lo[CNT], hi[CNT];
    __device__ __forceinline__ void set(int i, double x)
    {
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo[i]) : "v"(__double2loint(x)));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi[i]) : "v"(__double2hiint(x)));
    }
    __device__ __forceinline__ double get(int i) const
    {
        int l, h;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo[i]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi[i]));
        return __hiloint2double(h, l);
    }
};

// ---------------- the kernel, specialised on the problem shape ----------------
template <int NX, int NU, int N, int LPS>
struct Spec {
    static constexpr int n = N * NU;
    static_assert(LPS == 64 ? (n > 4 && n <= 64) : (n % LPS == 0 && n <= 32), "n must be a multiple of the group width (<= 32), or <= 64 for a whole wave");
    static_assert(LPS == 1 || LPS % NU == 0, "group width must be a multiple of nu");
    typedef typename std::conditional<(n > 32), unsigned long long, unsigned>::type mask_t;   // one bit per row
    static constexpr int RB = (n + LPS - 1) / LPS;  // rows per lane (LPS == 64: one, lanes >= n are padding)
    static constexpr int SPW = 64 / LPS;            // instances per wave
    __host__ __device__ static constexpr int rowlen(int jb) { return (jb + 1) * LPS < n ? (jb + 1) * LPS : n; }
    __host__ __device__ static constexpr int off(int jb) { return LPS == 64 ? jb * n : LPS * jb * (jb + 1) / 2; }   // closed form: stays a constant after unrolling
    static constexpr int TRI = off(RB);
    // LDS mirror of L, column-major: column c holds row-blocks jb >= c/LPS, 64 doubles each.
    // Inside a 64-double block the element of instance s, row-sub-lane rr sits at
    // s*LPS + (rr ^ (c % LPS)): the Cholesky's column writes (all lanes in one column) and the
    // backward substitution's reads (the LPS lanes of a group in LPS different columns, same row)
    // both touch 64 distinct slots -- bank-conflict-free without padding, so that mirror + Fq fill
    // the wave's 40 KiB share of LDS exactly (4 waves per CU = 160 KiB).
    __host__ __device__ static constexpr int colstart(int c)
    {
        int s = 0;
        for (int cc = 0; cc < c; ++cc) s += (RB - cc / LPS) * 64;
        return s;
    }
    static constexpr int MIRROR = (LPS == 1) ? 0 : colstart(n);
    static constexpr int STAGE = N * NX * NU * SPW;            // A^m B staging during condensing (aliases the mirror)
    static constexpr int FQ0 = MIRROR > STAGE ? MIRROR : STAGE; // Fq rows: [(jb*NX + a)*64 + lane]
    static constexpr int LDS_DOUBLES = FQ0 + RB * NX * 64;

    // ---- per-lane state (all in VGPRs) ----
    AccArr<TRI> Pm;        // own rows of P, row-block jb padded to rowlen(jb) columns (AGPRs)
    double a[TRI];         // K, then L; diagonal blocks are zero on and above the diagonal after chol()
    double invd[RB];       // 1 / l_ii of own rows
    AccArr<RB> qr;         // own rows of the constant part of q (references, box centre); with presolve: of v_r = -P^-1 qr
    AccArr<RB> qs;         // q of the QP being solved (kept for the polish)
    double sl[RB], su[RB], zl[RB], zu[RB], rd[RB];   // interior-point state; v = sl - h is implied
    double v[RB];          // result of the last QP (shifted inputs), set at the end of solve_qp
    double x[NX];          // current state (replicated in the group)
    int r, s, lane;        // sub-lane in group, group in wave, lane in wave
    bool pad;              // LPS == 64: this lane holds no row (r >= n); it must not influence any reduction
    mask_t prevL, prevU;   // active sets (all rows of the group) of the previous QP of a rollout, 0 if it was interior
    double *lds;

    // group-wide reductions; with one row per lane (LPS == 64) the lanes without a row are neutral
    __device__ __forceinline__ double gsum(double x) const { return group_sum<LPS>((LPS == 64 && pad) ? 0.0 : x); }
    __device__ __forceinline__ double gmax(double x) const { return group_max<LPS>((LPS == 64 && pad) ? -1e308 : x); }
    __device__ __forceinline__ mask_t gor(mask_t x) const { return group_or<LPS, mask_t>((LPS == 64 && pad) ? mask_t(0) : x); }
    __device__ __forceinline__ unsigned gand(unsigned x) const { return group_and<LPS>((LPS == 64 && pad) ? 1u : x); }
    __device__ __forceinline__ mask_t rowbit(int jb) const { return mask_t(1) << (jb * LPS + r_or0()); }

    // half-width and centre of own row jb
    __device__ __forceinline__ double hh(const KParams &p, int jb) const
    {
        // LPS > 1: all own rows belong to input r % NU (LPS is a multiple of NU); two cached loads, no register kept
        const int k = (LPS == 1) ? jb % NU : r % NU;
        return 0.5 * (p.sh[p.so.ub + k] - p.sh[p.so.lb + k]);
    }

    // ---- in-place Cholesky of the row-distributed matrix in a[] (right-looking, column k) ----
    // On exit: strict lower part = L, invd = 1/diag(L), diagonal blocks zero on/above the
    // diagonal, and (LPS > 1) L mirrored into LDS column-major.  Returns false on a bad pivot.
    //
    // Cross-lane traffic of step k (LPS > 1): every lane needs l_ck for all c > k.  Only l_{k+1,k}
    // travels by DPP -- it feeds the look-ahead update of column k+1 and so the next pivot, whose
    // rsqrt chain then overlaps the rest of the step; all other l_ck are read back from the column
    // just written to the LDS mirror (group-uniform addresses, two values per ds_read_b128), which
    // costs a quarter of the VALU slots of a DPP broadcast and whose latency the look-ahead hides.
    __device__ __forceinline__ bool chol()
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        bool ok = true;
        double d = bcast_rt(a[off(0)], 0);
        double inv = frsqrt(d);
        ok = ok && (d > 0.0);
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const int kb = k / LPS, kr = k % LPS;
            if (LPS == 1 || r == kr) invd[kb] = inv;
            // scale column k: rows below the diagonal; zero the diagonal block on/above it
            if constexpr (LPS == 1) a[off(kb) + k] = 0.0;
            else a[off(kb) + k] = (r > kr) ? a[off(kb) + k] * inv : 0.0;
#pragma unroll
            for (int jb = kb + 1; jb < RB; ++jb) a[off(jb) + k] *= inv;
            if constexpr (LPS > 1) {
#pragma unroll
                for (int jb = kb; jb < RB; ++jb) lds[colstart(k) + (jb - kb) * 64 + s * LPS + (r ^ kr)] = a[off(jb) + k];
            }
            double inv_next = 0.0;
            if (k + 1 < n) {
                // look-ahead: finish column k+1, then start the next pivot's reciprocal square root
                const int c = k + 1, cb = c / LPS, cr = c % LPS;
                const double lck = bcast_rt(a[off(cb) + k], cr);
#pragma unroll
                for (int jb = cb; jb < RB; ++jb) a[off(jb) + c] = __builtin_fma(-a[off(jb) + k], lck, a[off(jb) + c]);
                d = bcast_rt(a[off(cb) + c], cr);
                ok = ok && (d > 0.0);
                inv_next = frsqrt(d);
            }
            // trailing update of the columns c >= k+2: a[i][c] -= l_ik * l_ck, rows i >= c
            if constexpr (LPS == 1) {
#pragma unroll
                for (int c = k + 2; c < n; ++c) {
                    const double lck = a[off(c) + k];
#pragma unroll
                    for (int jb = c; jb < RB; ++jb) a[off(jb) + c] = __builtin_fma(-a[off(jb) + k], lck, a[off(jb) + c]);
                }
            } else if constexpr (LPS == 64) {
                // one row per lane: l_ck comes out of lane c as a scalar operand (rows above c update padding only)
#pragma unroll
                for (int c = k + 2; c < n; ++c) a[c] = __builtin_fma(-a[k], bcast_rt(a[k], c), a[c]);
            } else {
#pragma unroll
                for (int c0 = (k + 2) & ~1; c0 < n; c0 += 2) {
                    // columns c0, c0+1 sit in one aligned pair of the mirror block (positions cr ^ kr)
                    const int cb = c0 / LPS, pr = c0 % LPS;
                    const d2_t pair = *reinterpret_cast<const d2_t *>(&lds[colstart(k) + (cb - kb) * 64 + s * LPS + (pr ^ (kr & ~1))]);
                    const double l0 = (kr & 1) ? pair.y : pair.x, l1 = (kr & 1) ? pair.x : pair.y;
                    if (c0 >= k + 2) {
#pragma unroll
                        for (int jb = cb; jb < RB; ++jb) a[off(jb) + c0] = __builtin_fma(-a[off(jb) + k], l0, a[off(jb) + c0]);
                    }
                    if (c0 + 1 >= k + 2) {
#pragma unroll
                        for (int jb = cb; jb < RB; ++jb) a[off(jb) + c0 + 1] = __builtin_fma(-a[off(jb) + k], l1, a[off(jb) + c0 + 1]);
                    }
                }
            }
            inv = inv_next;
            if constexpr (LPS > 1) { if (k % LPS == LPS - 1) __builtin_amdgcn_sched_barrier(0); }
        }
        return ok;
    }

    // bcast with a source known at compile time after unrolling (switch folds away)
    __device__ __forceinline__ double bcast_rt(double xv, int src) const
    {
        if constexpr (LPS == 1) return xv;
        else if constexpr (LPS == 2) return src == 0 ? bcast<2, 0>(xv) : bcast<2, 1>(xv);
        else if constexpr (LPS == 64) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(xv), src), hi = __builtin_amdgcn_readlane(__double2hiint(xv), src);
            return __hiloint2double(hi, lo);
        } else {
            switch (src) {
            case 0: return bcast<4, 0>(xv);
            case 1: return bcast<4, 1>(xv);
            case 2: return bcast<4, 2>(xv);
            default: return bcast<4, 3>(xv);
            }
        }
    }

    // ---- solve (L L') y = b in place, b row-distributed ----
    __device__ __forceinline__ void solve(double (&b)[RB])
    {
        // forward, column-oriented: after step k every row i > k has b_i -= l_ik y_k.
        // b keeps the un-scaled residuals; y_i = b_i * invd_i is applied at the end.
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const int kb = k / LPS, kr = k % LPS;
            const double yk = bcast_rt(b[kb] * invd[kb], kr);
#pragma unroll
            for (int jb = kb; jb < RB; ++jb) {
                if (LPS == 1 && jb == kb) continue;   // its own row
                b[jb] = __builtin_fma(-a[off(jb) + k], yk, b[jb]);   // diagonal block: zeros for rows <= k
            }
            if constexpr (LPS > 1) { if (k % LPS == LPS - 1) __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) b[jb] *= invd[jb];
        // backward, column-oriented: after step k every row i < k has y_i -= l_ki x_k
        if constexpr (LPS == 1) {
#pragma unroll
            for (int k = n - 1; k >= 0; --k) {
                const double xk = b[k] * invd[k];
#pragma unroll
                for (int jb = 0; jb < k; ++jb) b[jb] = __builtin_fma(-a[off(k) + jb], xk, b[jb]);   // l_{k,jb} from my own row k
            }
        } else if constexpr (LPS == 64) {
            // one row per lane: l_{k,r} sits in mirror column r (clamped for the lanes without a row), slot k ^ r
            const int col = (r < n ? r : n - 1) * 64;
#pragma unroll
            for (int k = n - 1; k >= 0; --k) {
                const double xk = bcast_rt(b[0] * invd[0], k);
                b[0] = __builtin_fma(-lds[col + (k ^ r)], xk, b[0]);
            }
        } else {
            // l_{k,i} for my column i = jb*LPS + r sits in mirror column i, row k (zero when i >= k).  The
            // entries a row-block of steps needs are fetched one block ahead of the dependent chain
            // (double buffer), so the LDS latency is hidden behind the previous block's arithmetic.
            double cur[LPS][RB], nxt[LPS][RB];
#pragma unroll
            for (int kr = 0; kr < LPS; ++kr)
#pragma unroll
                for (int jb = 0; jb < RB; ++jb) cur[kr][jb] = lds[colstart_r(jb) + (RB - 1 - jb) * 64 + s * LPS + (kr ^ r)];
#pragma unroll
            for (int kb = RB - 1; kb >= 0; --kb) {
                if (kb > 0) {
#pragma unroll
                    for (int kr = 0; kr < LPS; ++kr)
#pragma unroll
                        for (int jb = 0; jb < kb; ++jb) nxt[kr][jb] = lds[colstart_r(jb) + (kb - 1 - jb) * 64 + s * LPS + (kr ^ r)];
                }
#pragma unroll
                for (int kr = LPS - 1; kr >= 0; --kr) {
                    const double xk = bcast_rt(b[kb] * invd[kb], kr);
#pragma unroll
                    for (int jb = 0; jb <= kb; ++jb) b[jb] = __builtin_fma(-cur[kr][jb], xk, b[jb]);
                }
                if (kb > 0) {
#pragma unroll
                    for (int kr = 0; kr < LPS; ++kr)
#pragma unroll
                        for (int jb = 0; jb < kb; ++jb) cur[kr][jb] = nxt[kr][jb];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) b[jb] *= invd[jb];
    }

    // colstart(jb*LPS + r) for the run-time sub-lane r: the columns of one row-block have equal sizes
    __device__ __forceinline__ int colstart_r(int jb) const { return colstart(jb * LPS) + r * ((RB - jb) * 64); }

    // ---- q = Fq x + qr for own rows ----
    __device__ __forceinline__ void linear_term(double (&q)[RB]) const
    {
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            double t = qr.get(jb);
#pragma unroll
            for (int aa = 0; aa < NX; ++aa) t = __builtin_fma(lds[FQ0 + (jb * NX + aa) * 64 + lane], x[aa], t);
            q[jb] = t;
        }
    }

    // ---- y = P w for a row-distributed w (symmetric product from the stored lower part) ----
    __device__ __forceinline__ void symv(const double (&w)[RB], double (&y)[RB]) const
    {
        if constexpr (LPS == 64) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) acc = __builtin_fma(Pm.get(j), bcast_rt(w[0], j), acc);
            y[0] = acc;
            return;
        }
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) y[jb] = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            // column j: lower part adds P_ij w_j to my rows i >= j; upper part adds P_ij w_i (i > j) to y_j
            const double wj = bcast_rt(w[j / LPS], j % LPS);
            double tj = 0.0;
#pragma unroll
            for (int jb = j / LPS; jb < RB; ++jb) {
                const bool low = (LPS == 1) ? true : (jb * LPS > j || r >= j - jb * LPS);        // i >= j
                const bool strict = (LPS == 1) ? (jb > j) : (jb * LPS > j || r > j - jb * LPS);  // i >  j
                const double pij = Pm.get(off(jb) + j);
                y[jb] = __builtin_fma(low ? pij : 0.0, wj, y[jb]);
                tj = __builtin_fma(strict ? pij : 0.0, w[jb], tj);
            }
            tj = gsum(tj);
            if (LPS == 1 || r == j % LPS) y[j / LPS] += tj;
            if constexpr (LPS > 1) { if (j % LPS == LPS - 1) __builtin_amdgcn_sched_barrier(0); }
        }
    }

    // ---- primal-dual active-set iterations ----
    // One iteration solves the QP restricted to the face given by the sets (rows in myL / myU sit on their
    // lower / upper bound, the others solve P_FF v_F = -(q_F + P_FA v_A)) and re-derives the sets from the
    // KKT signs: a free row outside the box becomes active, an active row whose multiplier has the wrong
    // sign is released.  A fixed point satisfies every KKT condition, i.e. it is the exact optimum; it is
    // committed to v[] only then.  Serves as warm start (sets guessed from the unconstrained minimiser)
    // and as polish (sets read off the interior-point iterate).  Groups with run == false are untouched.
    __device__ __forceinline__ bool pdas(const KParams &p, mask_t &myL, mask_t &myU, bool run, int maxit, double gtol, int &nfact)
    {
        bool conv = !run, dead = false;
        for (int k = 0; k < maxit; ++k) {
            if (!__any(!conv && !dead)) break;
            const mask_t colmask = gor(myL | myU);
            double rhs[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const mask_t bit = rowbit(jb);
                const double h = hh(p, jb);
                rhs[jb] = (myL & bit) ? -h : ((myU & bit) ? h : 0.0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (__any(colmask != 0)) {
                double pd[RB];
                symv(rhs, pd);
#pragma unroll
                for (int jb = 0; jb < RB; ++jb)
                    if (!((myL | myU) & rowbit(jb))) rhs[jb] = -(qs.get(jb) + pd[jb]);
            } else {
#pragma unroll
                for (int jb = 0; jb < RB; ++jb) rhs[jb] = -qs.get(jb);
            }
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const bool ai = ((myL | myU) & rowbit(jb)) != 0;
#pragma unroll
                for (int j = 0; j < rowlen(jb); ++j) {
                    const bool diag = (LPS == 1) ? (j == jb) : (j - jb * LPS == r);
                    const bool aj = ((colmask >> j) & 1u) != 0;
                    const double pv = Pm.get(off(jb) + j);     // read unconditionally: no branch per element
                    a[off(jb) + j] = (ai || aj) ? (diag ? 1.0 : 0.0) : pv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool ok = chol();
            __builtin_amdgcn_sched_barrier(0);
            solve(rhs);
            __builtin_amdgcn_sched_barrier(0);
            double g[RB];
            if (__any(colmask != 0)) symv(rhs, g);      // gradient P v + q on the active rows (zero on the free ones)
            else {
#pragma unroll
                for (int jb = 0; jb < RB; ++jb) g[jb] = -qs.get(jb);
            }
            mask_t nl = 0, nu = 0;
            unsigned fin = ok ? 1u : 0u;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const mask_t bit = rowbit(jb);
                const double h = hh(p, jb), vi = rhs[jb], gi = g[jb] + qs.get(jb);
                const bool isL = (myL & bit) != 0, isU = (myU & bit) != 0;
                const bool toL = isL ? (gi >= -gtol) : (!isU && vi < -h * (1.0 + 1e-12));
                const bool toU = isU ? (gi <= gtol) : (!isL && vi > h * (1.0 + 1e-12));
                if (toL) nl |= bit;
                if (toU) nu |= bit;
                if (!(fabs(vi) < 1e300)) fin = 0u;
            }
            const bool changed = gor((nl ^ myL) | (nu ^ myU)) != 0;
            fin = gand(fin);
            if (!conv && !dead) {
                nfact += 1;
                if (!fin) dead = true;
                else {
                    myL = nl; myU = nu;
                    if (!changed) {
                        conv = true;
#pragma unroll
                        for (int jb = 0; jb < RB; ++jb) v[jb] = rhs[jb];
                    }
                }
            }
        }
        return conv;
    }

    // ---- Mehrotra predictor-corrector interior point on min 1/2 v'Pv + qs'v, |v| <= h ----
    // ipm_init places the iterate at the box centre; ipm_run advances it until the complementarity gap and
    // the dual residual are below eps_rel (relative to |q|_inf) or the wave's iteration budget is spent, and
    // can be called again with a tighter eps_rel.  Groups with run == false are frozen.  The iterate lives
    // in sl, su, zl, zu; returns 0 converged, 1 budget exhausted, 2 non-finite.
    __device__ __forceinline__ void ipm_init(const KParams &p, double scale)
    {
        const double z0 = p.z0_scale * scale;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            const double h = hh(p, jb);
            sl[jb] = h; su[jb] = h; zl[jb] = z0; zu[jb] = z0; rd[jb] = qs.get(jb);
        }
    }
    __device__ __forceinline__ int ipm_run(const KParams &p, bool run, double scale, double eps_rel, int &budget, int &iters)
    {
        double hmin = 1e300;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) hmin = fmin(hmin, hh(p, jb));
        hmin = gmax(-hmin); hmin = -hmin;
        const double inv2n = 1.0 / (2.0 * n);
        const double mu_tol = eps_rel * scale * hmin, rd_tol = eps_rel * scale;
        int status = 1;
        bool live = run;                  // this group still iterates
        for (; budget > 0; --budget) {
            double mu = 0.0, rn = 0.0;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                mu = __builtin_fma(sl[jb], zl[jb], mu); mu = __builtin_fma(su[jb], zu[jb], mu);
                rn = fmax(rn, fabs(rd[jb]));
            }
            mu = gsum(mu) * inv2n;
            rn = gmax(rn);
            if (live) {
                if (!(mu < 1e300) || !(rn < 1e300)) { status = 2; live = false; }
                else if (mu <= mu_tol && rn <= rd_tol) { status = 0; live = false; }
            }
            if (!__any(live)) break;
            iters += live ? 1 : 0;
            // K = P + diag(zl/sl + zu/su)
            double isl[RB], isu[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                isl[jb] = frcp(sl[jb]); isu[jb] = frcp(su[jb]);
                const double dg = __builtin_fma(zl[jb], isl[jb], zu[jb] * isu[jb]);
#pragma unroll
                for (int j = 0; j < rowlen(jb); ++j) {
                    const bool diag = (LPS == 1) ? (j == jb) : (j - jb * LPS == r);
                    a[off(jb) + j] = Pm.get(off(jb) + j) + (diag ? dg : 0.0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool okc = chol();
            __builtin_amdgcn_sched_barrier(0);
            if (live && !okc) { status = 2; live = false; }
            // predictor
            double dva[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) dva[jb] = -rd[jb] - zl[jb] + zu[jb];
            solve(dva);
            double mp = 0.0, md = 0.0;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const double e = dva[jb] * isl[jb], f = dva[jb] * isu[jb];
                mp = fmax(mp, fmax(-e, f));
                md = fmax(md, fmax(1.0 + e, 1.0 - f));      // -dz_aff / z
            }
            mp = gmax(mp); md = gmax(md);
            const double apa = mp > 1.0 ? frcp(mp) : 1.0, ada = md > 1.0 ? frcp(md) : 1.0;
            double mua = 0.0;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const double d = dva[jb];
                const double dzl = -zl[jb] * __builtin_fma(isl[jb], d, 1.0), dzu = -zu[jb] * __builtin_fma(-isu[jb], d, 1.0);
                mua = __builtin_fma(__builtin_fma(apa, d, sl[jb]), __builtin_fma(ada, dzl, zl[jb]), mua);
                mua = __builtin_fma(__builtin_fma(-apa, d, su[jb]), __builtin_fma(ada, dzu, zu[jb]), mua);
            }
            mua = gsum(mua) * inv2n;
            double sg = mua * frcp(mu);
            sg = sg * sg * sg;
            const double smu = sg * mu;
            // corrector (complementarity residuals are recomputed after the solve rather than kept live)
            double dv[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const double d = dva[jb];
                const double dzla = -zl[jb] * __builtin_fma(isl[jb], d, 1.0), dzua = -zu[jb] * __builtin_fma(-isu[jb], d, 1.0);
                const double rcl = smu - sl[jb] * zl[jb] - d * dzla, rcu = smu - su[jb] * zu[jb] + d * dzua;
                dv[jb] = -rd[jb] + rcl * isl[jb] - rcu * isu[jb];
            }
            solve(dv);
            mp = 0.0; md = 0.0;
            double dzl[RB], dzu[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const double d = dva[jb];
                const double dzla = -zl[jb] * __builtin_fma(isl[jb], d, 1.0), dzua = -zu[jb] * __builtin_fma(-isu[jb], d, 1.0);
                const double rcl = smu - sl[jb] * zl[jb] - d * dzla, rcu = smu - su[jb] * zu[jb] + d * dzua;
                dzl[jb] = (rcl - zl[jb] * dv[jb]) * isl[jb];
                dzu[jb] = (rcu + zu[jb] * dv[jb]) * isu[jb];
                mp = fmax(mp, fmax(-dv[jb] * isl[jb], dv[jb] * isu[jb]));
                md = fmax(md, fmax(-dzl[jb] * frcp(zl[jb]), -dzu[jb] * frcp(zu[jb])));
            }
            mp = gmax(mp); md = gmax(md);
            // one step length for primal and dual: with unequal lengths the dual residual of a QP is not
            // monotone and the iteration can cycle (observed on weakly active constraints)
            mp = fmax(mp, md);
            double ap = mp > p.tau ? p.tau * frcp(mp) : 1.0;
            if (!live) ap = 0.0;
            const double ad = ap;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const double st = ap * dv[jb];
                sl[jb] += st; su[jb] -= st;
                zl[jb] = __builtin_fma(ad, dzl[jb], zl[jb]); zu[jb] = __builtin_fma(ad, dzu[jb], zu[jb]);
                rd[jb] = __builtin_fma(1.0 - ap, rd[jb], (ap - ad) * (dzl[jb] - dzu[jb]));
            }
        }
        return status;
    }

    // ---- one box QP at state x; the result goes to v[] ----
    // 1. presolve: unconstrained minimiser inside the box -> done.  2. warm start: primal-dual active-set
    // iterations from the rows where that minimiser leaves the box.  3. fallback: interior point, then the
    // same active-set iterations as polish.  Every path ends in a verified KKT point (or, if the polish
    // does not settle, in the interior-point iterate).
    __device__ __forceinline__ int solve_qp(const KParams &p, int &iters)
    {
        double q[RB];
        linear_term(q);
        bool inside = false;
        mask_t myL = 0, myU = 0;
        if (p.presolve) {
            // LDS holds G = -P^-1 Fq and qr holds v_r: q is the unconstrained minimiser v_unc = G x + v_r.
            unsigned in = 1u;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const mask_t bit = rowbit(jb);
                const double h = hh(p, jb);
                v[jb] = q[jb];
                if (!(fabs(q[jb]) <= h)) in = 0u;
                if (q[jb] < -h) myL |= bit;
                if (q[jb] > h) myU |= bit;
            }
            in = gand(in);
            inside = in != 0u;
            if (!__any(!inside)) { prevL = 0; prevU = 0; return 0; }   // every instance of the wave is done, exactly
            double y[RB];
            symv(q, y);                              // q = -P v_unc for the instances that must iterate
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) q[jb] = -y[jb];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) qs.set(jb, q[jb]);
        double scale = 0.0;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) scale = fmax(scale, fabs(q[jb]));
        scale = gmax(scale);
        const bool finite_in = scale < 1e300;
        scale = fmax(scale, 1e-100);
        const double gtol = 1e-10 * scale;
        bool todo = finite_in && !inside;            // this group still needs a solution
        int status = finite_in ? 0 : 2;
        if (!finite_in) {
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) v[jb] = 0.0;
        }
        if (p.warm_start) {
            // consecutive MPC problems: the previous optimum shifted by one stage predicts the active set
            // better than the clipped unconstrained minimiser (the last stage keeps its own state)
            if ((prevL | prevU) != 0) {
                constexpr mask_t tail = ((mask_t(1) << NU) - 1) << (n - NU);
                const mask_t gl = (prevL >> NU) | (prevL & tail), gu = (prevU >> NU) | (prevU & tail);
                mask_t own = 0;
#pragma unroll
                for (int jb = 0; jb < RB; ++jb) own |= rowbit(jb);
                myL = gl & own; myU = gu & own;
            }
            if (pdas(p, myL, myU, todo, p.max_iter < 8 ? p.max_iter : 8, gtol, iters)) todo = false;   // (max_iter caps the warm start too)
        }
        if (__any(todo)) {
            // Fallback.  With the polish on, the interior-point loop is run in stages (gap 1e-6, 1e-9, eps) and
            // after each stage the active-set iterations try to finish from the face it suggests (z > s): a
            // verified fixed point ends the solve early, and slow interior-point tails (weakly active
            // constraints) never have to reach eps on their own.
            ipm_init(p, scale);
            int budget = p.max_iter;
            double e_prev = 1e300;
#pragma unroll 1
            for (int stage = 0; stage < 3; ++stage) {
                const double e = !p.polish ? p.eps : (stage == 0 ? fmax(1e-6, p.eps) : (stage == 1 ? fmax(1e-9, p.eps) : p.eps));
                if (!(e < e_prev)) continue;        // nothing tighter left to do
                e_prev = e;
                const int st = ipm_run(p, todo, scale, e, budget, iters);
                if (todo) {
                    status = st;
#pragma unroll
                    for (int jb = 0; jb < RB; ++jb) v[jb] = (st == 2) ? 0.0 : sl[jb] - hh(p, jb);   // interior-point answer
                }
                if (p.polish) {
                    mask_t pl = 0, pu = 0;
#pragma unroll
                    for (int jb = 0; jb < RB; ++jb) {
                        const mask_t bit = rowbit(jb);
                        const bool lo = zl[jb] > sl[jb], up = (!lo) && (zu[jb] > su[jb]);
                        if (lo) pl |= bit;
                        if (up) pu |= bit;
                    }
                    const bool fixed = pdas(p, pl, pu, todo && st != 2, 4, gtol, iters);
                    if (todo && st != 2 && fixed) { status = 0; myL = pl; myU = pu; todo = false; }
                    if (todo && st == 2) todo = false;
                } else {
                    todo = false;
                }
                if (!__any(todo) || budget <= 0) break;
            }
        }
        prevL = inside ? mask_t(0) : gor(myL);
        prevU = inside ? mask_t(0) : gor(myU);
        return status;
    }

    __device__ __forceinline__ int r_or0() const { return LPS == 1 ? 0 : r; }

    // optimal input i (time-major index) of the last solve, in every lane of the group, clipped to the box
    __device__ __forceinline__ double u_at(const KParams &p, int i) const
    {
        const int k = i % NU;
        const double lb = p.sh[p.so.lb + k], ub = p.sh[p.so.ub + k];
        const double vi = bcast_rt(v[i / LPS], i % LPS);
        return fmin(fmax(vi + 0.5 * (lb + ub), lb), ub);
    }
};

// A, B entry of instance b (instance-minor)
#define LD(ptr, e) (ptr)[(long long)(e) * Bsz + b]
// A, B, x0 of instance b: from the caller's instance-minor arrays, or -- when the batch is walked in
// sorted order -- from the instance-major record [A | B | x0] the probe launch staged (contiguous per
// instance, so the permuted reads use whole sectors)
#define LDA(e) (p.rec ? p.rec[b * REC + (e)] : LD(p.A, e))
#define LDB(e) (p.rec ? p.rec[b * REC + NX * NX + (e)] : LD(p.B, e))
#define LDX(e) (p.rec ? p.rec[b * REC + NX * NX + NX * NU + (e)] : LD(p.x0, e))

// One wavefront's work: the SPW instances in slots slot0 .. slot0 + SPW - 1 (slots >= slot_end are surplus).
template <int NX, int NU, int N, int LPS, int MODE>
__device__ __forceinline__ void spec_body(const KParams &p, double *lds, long long slot0, long long slot_end)
{
    using S = Spec<NX, NU, N, LPS>;
    constexpr int n = S::n, RB = S::RB, SPW = S::SPW;
    constexpr int REC = NX * NX + NX * NU + NX;      // doubles per staged instance record
    S st;
    st.lds = lds;
    const int lane = threadIdx.x;
    st.s = lane / LPS; st.r = lane % LPS; st.lane = lane;
    st.prevL = 0; st.prevU = 0;
    st.pad = (LPS == 64) && (st.r >= n);
    const int r = st.r, s = st.s;
    const long long Bsz = p.Bsz;
    const long long b_raw = slot0 + s;
    const bool valid = b_raw < slot_end;
    const long long slot = valid ? b_raw : slot_end - 1;   // surplus groups recompute the last slot, never store
    // processing order: instances sorted by difficulty so that the 16 instances of a wave leave the
    // constrained regime together and the longest waves are dispatched first
    const long long b = p.perm ? (long long)p.perm[slot] : slot;
    const double *sh = p.sh;

    // ---------------- condensing ----------------
    // H = sum_rt Gamma_rt' Q_rt Gamma_rt + Rbar with Gamma_rt[:, j] = M[rt - bj][:, uj], M[m] = A^m B.
    // M lives in LDS ([m][input][state][instance]); a lane fetches the Gamma column of ITS row with a
    // lane-dependent address and the columns of the other side with group-uniform addresses, so
    // no A^m B table is held in registers.
    {
        double A[NX][NX], Bm[NX][NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) A[i][j] = LDA(i * NX + j);
#pragma unroll
            for (int k = 0; k < NU; ++k) Bm[i][k] = LDB(i * NU + k);
        }
        // stage M[m][k][i] at lds[((m*NU + k)*NX + i)*SPW + s]; every lane of a group computes the same
        // chain, sub-lane (e % LPS) stores element e
        {
            double Mc[NX][NU];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int k = 0; k < NU; ++k) Mc[i][k] = Bm[i][k];
#pragma unroll
            for (int m = 0; m < N; ++m) {
                if (m > 0) {
                    double T[NX][NU];
#pragma unroll
                    for (int i = 0; i < NX; ++i)
#pragma unroll
                        for (int k = 0; k < NU; ++k) {
                            double t = 0.0;
#pragma unroll
                            for (int j = 0; j < NX; ++j) t = __builtin_fma(A[i][j], Mc[j][k], t);
                            T[i][k] = t;
                        }
#pragma unroll
                    for (int i = 0; i < NX; ++i)
#pragma unroll
                        for (int k = 0; k < NU; ++k) Mc[i][k] = T[i][k];
                }
#pragma unroll
                for (int k = 0; k < NU; ++k)
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const int e = (m * NU + k) * NX + i;
                        if (LPS == 1 || e % LPS == r) lds[e * SPW + s] = Mc[i][k];
                    }
            }
        }
        double Pacc[S::TRI], Facc[RB][NX];
#pragma unroll
        for (int e = 0; e < S::TRI; ++e) Pacc[e] = 0.0;
        double qacc[RB];
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            qacc[jb] = 0.0;
#pragma unroll
            for (int aa = 0; aa < NX; ++aa) Facc[jb][aa] = 0.0;
        }
        // has_lin: references or an off-centre box contribute a constant to q
        bool has_lin = p.has_ref != 0;
#pragma unroll
        for (int k = 0; k < NU; ++k) has_lin = has_lin || (sh[p.so.ub + k] + sh[p.so.lb + k] != 0.0);
        double Ap[NX][NX];      // A^{rt+1}
        double sc[NX];          // forced response to the constant centre input: s_{rt+1}
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            sc[i] = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) Ap[i][j] = (i == j) ? 1.0 : 0.0;
        }
#pragma unroll 1
        for (int rt = 0; rt < N; ++rt) {
            {   // Ap <- A * Ap
                double T[NX][NX];
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int l = 0; l < NX; ++l) t = __builtin_fma(A[i][l], Ap[l][j], t);
                        T[i][j] = t;
                    }
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = 0; j < NX; ++j) Ap[i][j] = T[i][j];
            }
            double e[NX];       // s_{rt+1} - xref_rt (only with has_lin)
            if (has_lin) {
                double T[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) t = __builtin_fma(A[i][j], sc[j], t);
#pragma unroll
                    for (int k = 0; k < NU; ++k) t = __builtin_fma(Bm[i][k], 0.5 * (sh[p.so.ub + k] + sh[p.so.lb + k]), t);
                    T[i] = t;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { sc[i] = T[i]; e[i] = T[i] - (p.has_ref ? sh[p.so.xref + i * N + rt] : 0.0); }
            }
            const int oQ = (rt < N - 1) ? p.so.Q : p.so.P;     // terminal weight on x_N (utils_class.py:67-72)
            // w[jb] = Q_rt * Gamma_rt[:, my row of block jb]
            double w[RB][NX];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const int irow = jb * LPS + ((LPS == 1) ? 0 : r), bi = irow / NU, ui = irow % NU;
                const int m = rt - bi;
                const int mc = m < 0 ? 0 : m;
                double g[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double t = lds[((mc * NU + ui) * NX + i) * SPW + s];
                    g[i] = m < 0 ? 0.0 : t;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) t = __builtin_fma(sh[oQ + i * NX + j], g[j], t);
                    w[jb][i] = t;
                }
#pragma unroll
                for (int aa = 0; aa < NX; ++aa) {
                    double t = Facc[jb][aa];
#pragma unroll
                    for (int i = 0; i < NX; ++i) t = __builtin_fma(w[jb][i], Ap[i][aa], t);
                    Facc[jb][aa] = t;
                }
                if (has_lin) {
                    double t = qacc[jb];
#pragma unroll
                    for (int i = 0; i < NX; ++i) t = __builtin_fma(w[jb][i], e[i], t);
                    qacc[jb] = t;
                }
            }
            // P[my rows][j] += w . Gamma_rt[:, j] for every column j with bj <= rt
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const int bj = j / NU, uj = j % NU;
                if (bj > rt) continue;       // wave-uniform
                double mcol[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) mcol[i] = lds[(((rt - bj) * NU + uj) * NX + i) * SPW + s];
#pragma unroll
                for (int jb = j / LPS; jb < RB; ++jb) {
                    double t = Pacc[S::off(jb) + j];
#pragma unroll
                    for (int i = 0; i < NX; ++i) t = __builtin_fma(w[jb][i], mcol[i], t);
                    Pacc[S::off(jb) + j] = t;
                }
            }
        }
        // + Rbar on the block diagonal, the R part of qr, then the factor 2
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            const int irow = jb * LPS + ((LPS == 1) ? 0 : r), bi = irow / NU, ui = irow % NU;
#pragma unroll
            for (int j = 0; j < S::rowlen(jb); ++j) {
                const int bj = j / NU, uj = j % NU;
                const double rv = sh[p.so.R + ui * NU + uj];
                st.Pm.set(S::off(jb) + j, 2.0 * (Pacc[S::off(jb) + j] + ((bj == bi) ? rv : 0.0)));
            }
#pragma unroll
            for (int aa = 0; aa < NX; ++aa) lds[S::FQ0 + (jb * NX + aa) * 64 + lane] = 2.0 * Facc[jb][aa];
            double tq = qacc[jb];
            if (has_lin) {
#pragma unroll
                for (int uj = 0; uj < NU; ++uj) {
                    const double cu = 0.5 * (sh[p.so.ub + uj] + sh[p.so.lb + uj]);
                    const double ur = p.has_ref ? sh[p.so.uref + uj * N + (bi < N ? bi : N - 1)] : 0.0;   // bi >= N: padding lane
                    tq = __builtin_fma(sh[p.so.R + ui * NU + uj], cu - ur, tq);
                }
            }
            st.qr.set(jb, 2.0 * tq);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.presolve) {
        // G = -P^-1 Fq (one column per state) and v_r = -P^-1 qr overwrite Fq and qr
#pragma unroll
        for (int e = 0; e < S::TRI; ++e) st.a[e] = st.Pm.get(e);
        st.chol();
        __builtin_amdgcn_sched_barrier(0);
        for (int aa = 0; aa <= NX; ++aa) {
            double col[RB];
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) col[jb] = (aa < NX) ? lds[S::FQ0 + (jb * NX + aa) * 64 + lane] : st.qr.get(jb);
            st.solve(col);
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                if (aa < NX) lds[S::FQ0 + (jb * NX + aa) * 64 + lane] = -col[jb];
                else st.qr.set(jb, -col[jb]);
            }
        }
    }

    // ---------------- the requested operation ----------------
    const bool writer = valid && (LPS == 1 || r == 0);
    int iters = 0, status = 0;

    auto value_fn = [&](const double (&x0)[NX]) -> double {
        // V_N = cost* + x0'Qx0 by rolling the MODEL forward with the optimal inputs (utils_class.py:62-75, 91)
        double cost = 0.0, xs[NX];
        double A[NX][NX], Bm[NX][NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) A[i][j] = LDA(i * NX + j);
#pragma unroll
            for (int k = 0; k < NU; ++k) Bm[i][k] = LDB(i * NU + k);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            xs[i] = x0[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) cost = __builtin_fma(x0[i] * sh[p.so.Q + i * NX + j], x0[j], cost);
        }
#pragma unroll
        for (int t = 0; t < N; ++t) {
            double u[NU], xn[NX];
#pragma unroll
            for (int k = 0; k < NU; ++k) u[k] = st.u_at(p, t * NU + k);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) acc = __builtin_fma(A[i][j], xs[j], acc);
#pragma unroll
                for (int k = 0; k < NU; ++k) acc = __builtin_fma(Bm[i][k], u[k], acc);
                xn[i] = acc;
            }
            const int oQ = (t < N - 1) ? p.so.Q : p.so.P;
            double dx[NX], du[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) { xs[i] = xn[i]; dx[i] = xn[i] - (p.has_ref ? sh[p.so.xref + i * N + t] : 0.0); }
#pragma unroll
            for (int k = 0; k < NU; ++k) du[k] = u[k] - (p.has_ref ? sh[p.so.uref + k * N + t] : 0.0);
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NX; ++j) cost = __builtin_fma(dx[i] * sh[oQ + i * NX + j], dx[j], cost);
#pragma unroll
            for (int k = 0; k < NU; ++k)
#pragma unroll
                for (int j = 0; j < NU; ++j) cost = __builtin_fma(du[k] * sh[p.so.R + k * NU + j], du[j], cost);
        }
        return cost;
    };

    if constexpr (MODE == MODE_SOLVE) {
        double x0[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { x0[i] = LDX(i); st.x[i] = x0[i]; }
        status = st.solve_qp(p, iters);
        const double vn = value_fn(x0);
        double u0[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) u0[k] = st.u_at(p, k);
        if (writer) {
#pragma unroll
            for (int k = 0; k < NU; ++k) p.u0[(long long)k * Bsz + b] = u0[k];
            p.VN[b] = vn;
        }
    } else if constexpr (MODE == MODE_MAXVN) {
        double best = -1e308;
        for (int k = 0; k < p.K; ++k) {
            double x0[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { x0[i] = sh[p.so.x0s + i * p.K + k]; st.x[i] = x0[i]; }
            st.prevL = 0; st.prevU = 0;              // unrelated initial states: no carry-over of the active set
            const int stt = st.solve_qp(p, iters);
            status = stt > status ? stt : status;
            const double vn = value_fn(x0);
            best = (vn > best || vn != vn) ? vn : best;
        }
        if (writer) p.MV[b] = best;
    } else {
        double cost = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) st.x[i] = LDX(i);
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) cost = __builtin_fma(st.x[i] * sh[p.so.Q + i * NX + j], st.x[j], cost);   // utils_class.py:261
        if (p.X && writer) {
#pragma unroll
            for (int i = 0; i < NX; ++i) p.X[((long long)i * (p.T + 1)) * Bsz + b] = st.x[i];
        }
        for (int t = 0; t < p.T; ++t) {                                                  // utils_class.py:266-283
            const int stt = st.solve_qp(p, iters);
            status = stt > status ? stt : status;
            double u[NU], xn[NX];
#pragma unroll
            for (int k = 0; k < NU; ++k) u[k] = st.u_at(p, k);
            // line 277: plant step.  The plant is re-read every step (scalar loads when shared, L2-resident
            // vector loads when per-instance) instead of being held in registers across the QP; the opaque
            // copies of the loop counter / base pointers keep the compiler from hoisting 24 addresses out
            // of the step loop.
            long long tb = b;
            asm volatile("" : "+v"(tb));
            if (p.true_per_instance) {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc = __builtin_fma(p.At[(long long)(i * NX + j) * Bsz + tb], st.x[j], acc);
#pragma unroll
                    for (int k = 0; k < NU; ++k) acc = __builtin_fma(p.Bt[(long long)(i * NU + k) * Bsz + tb], u[k], acc);
                    xn[i] = acc;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc = __builtin_fma(sh[p.so.At + i * NX + j], st.x[j], acc);
#pragma unroll
                    for (int k = 0; k < NU; ++k) acc = __builtin_fma(sh[p.so.Bt + i * NU + k], u[k], acc);
                    xn[i] = acc;
                }
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) st.x[i] = xn[i];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NX; ++j) cost = __builtin_fma(xn[i] * sh[p.so.Q + i * NX + j], xn[j], cost);   // 282
#pragma unroll
            for (int k = 0; k < NU; ++k)
#pragma unroll
                for (int j = 0; j < NU; ++j) cost = __builtin_fma(u[k] * sh[p.so.R + k * NU + j], u[j], cost);      // 283
            if (writer) {
                if (p.X) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) p.X[((long long)i * (p.T + 1) + t + 1) * Bsz + tb] = xn[i];
                }
                if (p.U) {
#pragma unroll
                    for (int k = 0; k < NU; ++k) p.U[((long long)k * p.T + t) * Bsz + tb] = u[k];
                }
            }
        }
        if (writer) p.JT[b] = cost;
    }
    if (writer) {
        if (p.status) p.status[b] = status;
        if (p.iters) p.iters[b] = iters;
    }
}

template <int NX, int NU, int N, int LPS, int MODE>
__global__ void __launch_bounds__(64, 1) lqmpc_spec_kernel(KParams p)
{
    using S = Spec<NX, NU, N, LPS>;
    __shared__ double lds[S::LDS_DOUBLES];
    long long slot_end = p.Bsz;
    if (p.count_dev) {                     // fallback pass over a device-side list: surplus wavefronts leave at once
        const long long cnt = *p.count_dev;
        slot_end = cnt < slot_end ? cnt : slot_end;
        if ((long long)blockIdx.x * S::SPW >= slot_end) return;
    }
    spec_body<NX, NU, N, LPS, MODE>(p, lds, (long long)blockIdx.x * S::SPW, slot_end);
}

// The fallback pass over a device-side list for the one-instance-per-wavefront shapes: a bounded grid whose workgroups walk the list
// (the plain kernel would start one workgroup per instance of the batch -- 262 144 at C4, 56 us of dispatch -- only for all of them to
// read a count of zero and leave).
template <int NX, int NU, int N, int LPS, int MODE>
__global__ void __launch_bounds__(64, 1) lqmpc_spec_list_kernel(KParams p)
{
    using S = Spec<NX, NU, N, LPS>;
    __shared__ double lds[S::LDS_DOUBLES];
    const long long cnt = *p.count_dev;
    const long long slot_end = cnt < p.Bsz ? cnt : p.Bsz;
#pragma unroll 1
    for (long long blk = blockIdx.x; blk * S::SPW < slot_end; blk += gridDim.x) {
        spec_body<NX, NU, N, LPS, MODE>(p, lds, blk * S::SPW, slot_end);
        __syncthreads();
    }
}

// Two tiers in one launch (sorted rollouts).  The launch is as long as its slowest wavefront, and in the packed
// layout that is the wavefront with the instances that stay constrained for all T steps: 16 instances, ~4000
// instructions per active-set iteration, up to 78 iterations at C3.  So the p.nwide hardest instances, first
// in the order, run in the 16-lane-row layout of lqmpc_r16_body.h (four per wavefront, a few hundred
// instructions per iteration because only the smaller side of the active-set system is solved), everything
// after them runs packed.  Workgroups dispatch in index order, so the hard ones start first.
template <int NX, int NU, int N, int LPS, int MODE>
__global__ void __launch_bounds__(64, 1) lqmpc_spec_tiered_kernel(KParams p)
{
    using SP = Spec<NX, NU, N, LPS>;
    constexpr int R16SZ = 4 * R16<NX, NU, N>::INST;
    constexpr int LDSZ = R16SZ > SP::LDS_DOUBLES ? R16SZ : SP::LDS_DOUBLES;
    __shared__ double lds[LDSZ];
    const long long blk = blockIdx.x, nb16 = (p.nwide + 3) / 4;
    if (blk < nb16) r16_body<NX, NU, N, MODE>(p, lds, blk * 4, p.nwide);
    else spec_body<NX, NU, N, LPS, MODE>(p, lds, p.nwide + (blk - nb16) * SP::SPW, p.Bsz);
}

// slot of an instance = instances in harder buckets + its position inside its bucket.  Every block scans the bucket counts
// itself (4096 counters: cheaper than another launch) and places 1024 instances.
__global__ void __launch_bounds__(256) lqmpc_order_scatter_kernel(const int2 *where, const int *hist, int *perm, long long Bsz)
{
    __shared__ int base[ORDER_CELLS];
    __shared__ int part[256];
    constexpr int PER = ORDER_CELLS / 256;
    const int t = threadIdx.x;
    int above_in_chunk[PER], sum = 0;
#pragma unroll
    for (int e = PER - 1; e >= 0; --e) { above_in_chunk[e] = sum; sum += hist[(PER * t + e) * ORDER_PAD]; }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d *= 2) {                       // inclusive suffix sums over the threads' chunks
        const int v = part[t] + (t + d < 256 ? part[t + d] : 0);
        __syncthreads();
        part[t] = v;
        __syncthreads();
    }
    const int above = part[t] - sum;
#pragma unroll
    for (int e = 0; e < PER; ++e) base[PER * t + e] = above + above_in_chunk[e];
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * 1024;
    for (long long i = i0 + t; i < i0 + 1024 && i < Bsz; i += 256) {
        const int2 w = where[i];
        perm[base[w.x] + w.y] = (int)i;
    }
}

void launch_order_scatter(const KParams &p, int *perm, hipStream_t stream)
{
    hipLaunchKernelGGL(lqmpc_order_scatter_kernel, dim3((unsigned)((p.Bsz + 1023) / 1024)), dim3(256), 0, stream,
                       (const int2 *)p.key, (const int *)p.hist, perm, (long long)p.Bsz);
}

// ---------------- registry of built specialisations ----------------
struct SpecEntry {
    int nx, nu, N, lps;
    const char *name;
    void (*launch)(const KParams &, hipStream_t);
    void (*launch_tiered)(const KParams &, hipStream_t);   // rollouts with p.nwide > 0 (null: not built)
    const char *name_tiered;
};

template <int NX, int NU, int N, int LPS>
static void launch_one(const KParams &p, hipStream_t stream)
{
    constexpr int SPW = 64 / LPS;
    const unsigned grid = (unsigned)((p.Bsz + SPW - 1) / SPW);
    if constexpr (LPS == 64) {
        if (p.count_dev && p.mode != MODE_PROBE) {
            const unsigned gl = grid < 2048u ? grid : 2048u;
            if (p.mode == MODE_SOLVE)
                hipLaunchKernelGGL((lqmpc_spec_list_kernel<NX, NU, N, LPS, MODE_SOLVE>), dim3(gl), dim3(64), 0, stream, p);
            else if (p.mode == MODE_MAXVN)
                hipLaunchKernelGGL((lqmpc_spec_list_kernel<NX, NU, N, LPS, MODE_MAXVN>), dim3(gl), dim3(64), 0, stream, p);
            else
                hipLaunchKernelGGL((lqmpc_spec_list_kernel<NX, NU, N, LPS, MODE_ROLLOUT>), dim3(gl), dim3(64), 0, stream, p);
            return;
        }
    }
    if (p.mode == MODE_SOLVE)
        hipLaunchKernelGGL((lqmpc_spec_kernel<NX, NU, N, LPS, MODE_SOLVE>), dim3(grid), dim3(64), 0, stream, p);
    else if (p.mode == MODE_MAXVN)
        hipLaunchKernelGGL((lqmpc_spec_kernel<NX, NU, N, LPS, MODE_MAXVN>), dim3(grid), dim3(64), 0, stream, p);
    else if (p.mode == MODE_PROBE)
        hipLaunchKernelGGL((lqmpc_probe_kernel<NX, NU, N>), dim3((unsigned)((p.Bsz + 63) / 64)), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL((lqmpc_spec_kernel<NX, NU, N, LPS, MODE_ROLLOUT>), dim3(grid), dim3(64), 0, stream, p);
}

template <int NX, int NU, int N, int LPS>
static void launch_tiered(const KParams &p, hipStream_t stream)
{
    constexpr int SPW = 64 / LPS;
    const unsigned grid = (unsigned)((p.nwide + 3) / 4 + (p.Bsz - p.nwide + SPW - 1) / SPW);
    hipLaunchKernelGGL((lqmpc_spec_tiered_kernel<NX, NU, N, LPS, MODE_ROLLOUT>), dim3(grid), dim3(64), 0, stream, p);
}

#define SPEC(NX, NU, N, LPS) {NX, NU, N, LPS, "lqmpc_spec_kernel<" #NX "," #NU "," #N "," #LPS ">", launch_one<NX, NU, N, LPS>, nullptr, nullptr}
#define SPEC_TIERED(NX, NU, N, LPS) {NX, NU, N, LPS, "lqmpc_spec_kernel<" #NX "," #NU "," #N "," #LPS ">", launch_one<NX, NU, N, LPS>, \
                                     launch_tiered<NX, NU, N, LPS>, "lqmpc_spec_tiered_kernel<" #NX "," #NU "," #N "," #LPS ">"}

static const SpecEntry g_specs[] = {
    SPEC(2, 1, 5, 1),     // C1  (working_example_single.py shape, N = 5)
    SPEC(2, 1, 10, 2),    // C2
    SPEC_TIERED(4, 2, 10, 4),   // C3  (headline)
    SPEC(4, 2, 20, 64),   // C4  (n = 40: one matrix row per lane, a whole wave per instance)
    SPEC(2, 1, 30, 64),   // the reference's N_opc = 30 (V_expert, working_example_multiple.py:35)
    SPEC(2, 1, 20, 4),    // mpc_test.py:12 (N_open = 20)
    SPEC(2, 1, 6, 2), SPEC(2, 1, 7, 1), SPEC(2, 1, 8, 2), SPEC(2, 1, 9, 1),   // the reference's horizon sweep 6..10
};

static const SpecEntry *find_spec(int nx, int nu, int N)
{
    for (const SpecEntry &e : g_specs)
        if (e.nx == nx && e.nu == nu && e.N == N) return &e;
    return nullptr;
}

bool spec_available(int nx, int nu, int N) { return find_spec(nx, nu, N) != nullptr; }
bool spec_tiered_available(int nx, int nu, int N) { const SpecEntry *e = find_spec(nx, nu, N); return e && e->launch_tiered; }

bool launch_spec(const KParams &p, hipStream_t stream, const char **name)
{
    const SpecEntry *e = find_spec(p.nx, p.nu, p.N);
    if (!e) return false;
    if (p.nwide > 0 && p.mode == MODE_ROLLOUT && e->launch_tiered) {
        e->launch_tiered(p, stream);
        if (name) *name = e->name_tiered;
        return true;
    }
    e->launch(p, stream);
    if (name) *name = e->name;
    return true;
}

}  // namespace lqmpc
