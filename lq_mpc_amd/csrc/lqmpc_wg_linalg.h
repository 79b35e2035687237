// lqmpc_wg_linalg.h -- workgroup-cooperative dense linear algebra in LDS for the large-n kernel.
//
// One 256-thread workgroup (4 waves) owns one n x n system, n <= 128.  The symmetric working matrix K lives
// in LDS as the lower block triangle of 16x16 blocks: block (ib, jb), jb <= ib, at index ib(ib+1)/2 + jb,
// each block row-major with a row stride of 17 doubles (272 doubles per block).  The stride makes both
// access patterns of v_mfma_f64_16x16x4_f64 cheap: its A/B operand fetch (lane -> element [lane%16][4t +
// lane/16]) and its C/D tile (lane -> rows lane/16 + 4r, column lane%16).
//
// chol_blocked(): right-looking blocked Cholesky.  Per block column kb:
//   wave 0 factors the 16x16 diagonal block with one matrix row per lane (v_readlane broadcasts) and
//   inverts it; barrier; the panel blocks X = A L_kk^-T are formed as matrix products with the inverse and
//   the trailing blocks A_ij -= L_ik L_jk' as rank-16 updates, both on the f64 MFMA, block products dealt
//   round-robin to the 4 waves; barrier.  16 barriers per factorisation of n = 128.
// After it: strict lower blocks hold L, diagonal blocks hold L_kk (lower triangle), Linv holds L_kk^-1.
#pragma once
#include <hip/hip_runtime.h>
#include "lqmpc_common.h"

namespace lqmpc {
namespace wg {

constexpr int BS = 16;          // block size
constexpr int LD = 17;          // row stride inside a block (doubles)
constexpr int BLK = BS * LD;    // doubles per block
constexpr int THREADS = 256;

__device__ __forceinline__ int blk_index(int ib, int jb) { return ib * (ib + 1) / 2 + jb; }

typedef double d4_t __attribute__((ext_vector_type(4)));

// every matrix / vector of this header lives in LDS: pointers carry the address space so that accesses are
// ds_read / ds_write even where the compiler cannot trace them back to the __shared__ array
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) int ldsi;

__device__ __forceinline__ double rdlane(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// D = C +- op(X) * op(Y)'  for 16x16 blocks X, Y in LDS (row stride LD), op = identity or transpose.
// C/D in the MFMA tile layout (element r of the result: row lane/16 + 4r, column lane%16).  One wave.
// Operand fetch of v_mfma_f64_16x16x4_f64: lane l supplies A[l%16][4t + l/16] and B'[l%16][4t + l/16].
template <bool XT, bool YT>
__device__ __forceinline__ d4_t block_mm(const ldsd *X, const ldsd *Y, d4_t c, bool negate)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a[t] = XT ? X[(4 * t + kq) * LD + i] : X[i * LD + 4 * t + kq];
        b[t] = YT ? Y[(4 * t + kq) * LD + i] : Y[i * LD + 4 * t + kq];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a[t] : a[t], b[t], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ d4_t block_xyt(const ldsd *X, const ldsd *Y, d4_t c, bool negate) { return block_mm<false, false>(X, Y, c, negate); }

__device__ __forceinline__ d4_t tile_load(const ldsd *C)
{
    const int lane = threadIdx.x & 63, col = lane & 15, r0 = lane >> 4;
    d4_t c;
    c.x = C[(r0 + 0) * LD + col]; c.y = C[(r0 + 4) * LD + col]; c.z = C[(r0 + 8) * LD + col]; c.w = C[(r0 + 12) * LD + col];
    return c;
}
__device__ __forceinline__ void tile_store(ldsd *C, d4_t c)
{
    const int lane = threadIdx.x & 63, col = lane & 15, r0 = lane >> 4;
    C[(r0 + 0) * LD + col] = c.x; C[(r0 + 4) * LD + col] = c.y; C[(r0 + 8) * LD + col] = c.z; C[(r0 + 12) * LD + col] = c.w;
}

// ---- 16-lane row broadcasts (DPP row_newbcast, gfx90a+): lane c of every 16-lane row to the whole row ----
// With one matrix row per lane of a 16-lane row these are the column broadcasts of a 16x16 factorisation;
// the value stays in vector registers (no v_readlane / SGPR round trip).
#define LQMPC_ROWB_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// (A compiler builtin on purpose: as inline asm with its own s_nop -- which would make the wait states after an inline-asm
// DPP update structural -- the value functions of the one-shot / max-V_N / sweep builds, ten broadcasts per horizon, spill 470
// registers instead of 100 and run 2-5x slower; measured round 2.  The rule stays at the call sites: dpp_settle() before reading
// through rowb() what an asm statement wrote; every built shape's set-up and iterations are checked in the -m gpu parity tests.)
__device__ __forceinline__ double rowb(double x, int c)          // c: compile-time constant after unrolling
{
    long long v = __double_as_longlong(x);
    switch (c) {
#define LQMPC_X(C) case C: v = __builtin_amdgcn_update_dpp(0ll, v, 0x150 + C, 0xF, 0xF, true); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
    return __longlong_as_double(v);
}
// acc += (lane c's x, broadcast over the row) * y, one v_fmac_f64_dpp.  A DPP read needs its source written
// two wait states earlier; the compiler does not look into inline asm, so every use carries its own s_nop 1
// (the loops below are latency-bound on the rsqrt chain, not on issue slots).
__device__ __forceinline__ void fmac_rowb(double &acc, double x, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// acc += (lane c's acc, broadcast over the row) * y: the elimination update, source and destination one operand.
// (Dropping the s_nop here was tried and is wrong on gfx950: under register pressure the compiler copies or
// reloads the operand right before the statement, and a DPP read two slots after a VALU write returns stale data.)
__device__ __forceinline__ void fmac_rowb_self(double &acc, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// Four updates with one lane and one multiplier behind a single s_nop: the operands are all inputs of the one
// statement (whatever copies the allocator needs happen before its s_nop) and no instruction inside reads through DPP
// what an earlier one wrote (a_k and x_k are distinct registers), so the elimination loops issue 5 slots per 4 updates
// instead of 8.
__device__ __forceinline__ void fmac_rowb4(double &a0, double &a1, double &a2, double &a3, double x0, double x1, double x2, double x3, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %8 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %5, %8 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %3, %7, %8 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}
__device__ __forceinline__ void fmac_rowb_self4(double &a0, double &a1, double &a2, double &a3, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %1, %4 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %2, %2, %4 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %3, %3, %4 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// two wait states tied to x: a DPP read of x that follows in program order is safe even if x was written by
// the inline asm right before (the compiler's hazard recogniser does not see those writes)
__device__ __forceinline__ void dpp_settle(double &x) { asm volatile("s_nop 1" : "+v"(x)); }

// 1/sqrt(x): v_rsq_f64 (relative error 2^-24 on gfx950, tools/wg_test) + one cubic step -> ~1 ulp
__device__ __forceinline__ double frsqrt1(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

// One wave: factor the diagonal block D (in place: lower triangle <- L, upper <- 0) and write its inverse
// (lower triangular) to Dinv.  Returns false on a non-positive pivot.
//   factor:  one matrix row per lane (lanes 16..63 mirror lanes 0..15); per column k: broadcast the pivot,
//            rsqrt, scale, then one v_fmac_f64_dpp per remaining column (row_newbcast brings l_ck) --
//            16 dependent steps, each one rsqrt chain long;
//   inverse: L goes to LDS, then lane j solves L z = e_j by forward substitution; every L_ik it needs is the
//            same for all lanes, i.e. a broadcast LDS read, and nothing crosses lanes.
__device__ __forceinline__ bool diag_factor_invert(ldsd *D, ldsd *Dinv)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    double row[BS], sinv[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) row[j] = D[i * LD + j];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        const double d = rowb(row[k], k);
        ok = ok && (d > 0.0);
        const double s = frsqrt1(d);
        sinv[k] = s;
        row[k] *= s;                                             // l_kk = d / sqrt(d) in lane k, l_ik = a_ik / l_kk below it
        const double nl = -row[k];
#pragma unroll
        for (int c = k + 1; c < BS; ++c) fmac_rowb(row[c], row[k], nl, c);   // a_ic -= l_ik l_ck
    }
    if (lane < BS) {
#pragma unroll
        for (int j = 0; j < BS; ++j) D[i * LD + j] = (j <= i) ? row[j] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    double z[BS];                                                // column i of L^-1: z_r = (delta_ri - sum_{k<r} l_rk z_k) / l_rr
#pragma unroll
    for (int r = 0; r < BS; ++r) z[r] = (r == i) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        z[k] *= sinv[k];
#pragma unroll
        for (int r = k + 1; r < BS; ++r) z[r] = __builtin_fma(-D[r * LD + k], z[k], z[r]);   // independent across r
    }
    if (lane < BS) {
#pragma unroll
        for (int r = 0; r < BS; ++r) Dinv[r * LD + i] = z[r];
    }
    return ok;
}

// One wave: solve S x = r for an m x m SPD system, m <= 16, S in one block (row stride LD, rows/columns >= m
// hold the identity), r and x in LDS (x may alias r).  T: a block of LDS scratch.  Returns false on a
// non-positive pivot.  Row per lane; the forward substitution rides along with the factorisation, the
// backward one uses the transpose of L fetched back from T.
__device__ __forceinline__ bool small_spd_solve(const ldsd *S, ldsd *T, const ldsd *r, ldsd *x, int m)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    double row[BS], sinv[BS], lt[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) row[j] = S[i * LD + j];
    double acc = r[i], y = 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        sinv[k] = 1.0;
        if (k < m) {                                             // uniform
            const double d = rowb(row[k], k);
            ok = ok && (d > 0.0);
            const double s = frsqrt1(d);
            sinv[k] = s;
            row[k] *= s;
            const double nl = -row[k];
            const double yk = rowb(acc, k) * s;                  // forward substitution, column k
            y = (i == k) ? yk : y;
            acc = __builtin_fma(nl, yk, acc);
    #pragma unroll
            for (int c = k + 1; c < BS; ++c)
                if (c < m) fmac_rowb(row[c], row[k], nl, c);
        }
    }
    if (lane < BS) {
#pragma unroll
        for (int j = 0; j < BS; ++j) T[i * LD + j] = row[j];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < BS; ++k) lt[k] = T[k * LD + i];          // L[k][lane]
    double xs = 0.0;
    acc = y;
#pragma unroll
    for (int k = BS - 1; k >= 0; --k) {
        if (k < m) {
            const double xk = rowb(acc, k) * sinv[k];
            xs = (i == k) ? xk : xs;
            acc = __builtin_fma(-lt[k], xk, acc);                // lanes j < k: y_j - sum_{r > j} l_rj x_r
        }
    }
    if (lane < BS) x[i] = (i < m) ? xs : 0.0;
    return ok;
}

// Blocked Cholesky of the nb x nb block matrix at K (LDS); Linv: nb diagonal-block inverses.  All 256 threads.
__device__ __noinline__ bool chol_blocked(ldsd *K, ldsd *Linv, int nb, ldsi *flag)
{
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) *flag = 1;
    for (int kb = 0; kb < nb; ++kb) {
        __syncthreads();
        if (wave == 0) {
            const bool ok = diag_factor_invert(K + blk_index(kb, kb) * BLK, Linv + kb * BLK);
            if (!ok && (threadIdx.x & 63) == 0) *flag = 0;
        }
        __syncthreads();
        // panel: X_ib = A_ib,kb * Linv_kk'   (ib > kb), in place
        for (int ib = kb + 1 + wave; ib < nb; ib += 4) {
            ldsd *A = K + blk_index(ib, kb) * BLK;
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            c = block_xyt(A, Linv + kb * BLK, c, false);
            tile_store(A, c);           // the wave read all of A's operands before this store (same wave, in order)
        }
        __syncthreads();
        // trailing update: A_ib,jb -= X_ib X_jb'   (kb < jb <= ib)
        int cnt = 0;
        for (int ib = kb + 1; ib < nb; ++ib)
            for (int jb = kb + 1; jb <= ib; ++jb, ++cnt) {
                if ((cnt & 3) != wave) continue;
                ldsd *C = K + blk_index(ib, jb) * BLK;
                d4_t c = tile_load(C);
                c = block_xyt(K + blk_index(ib, kb) * BLK, K + blk_index(jb, kb) * BLK, c, true);
                tile_store(C, c);
            }
    }
    __syncthreads();
    return *flag != 0;
}

// Solve (L L') x = b.  b: LDS vector of nb*16 doubles (in place).  tmp: LDS scratch of 16 doubles.
__device__ __noinline__ void solve_blocked(const ldsd *K, const ldsd *Linv, int nb, ldsd *b)
{
    const int t = threadIdx.x;
    // forward: y_kb = Linv_kk b_kb ; b_ib -= L_ib,kb y_kb
    for (int kb = 0; kb < nb; ++kb) {
        __syncthreads();
        double y = 0.0;
        if (t < BS) {
            const ldsd *Z = Linv + kb * BLK + t * LD;
#pragma unroll
            for (int j = 0; j < BS; ++j) y = __builtin_fma(Z[j], b[kb * BS + j], y);
        }
        __syncthreads();
        if (t < BS) b[kb * BS + t] = y;
        __syncthreads();
        const int row = (kb + 1) * BS + t;
        if (row < nb * BS) {
            const int ib = row / BS, r = row % BS;
            const ldsd *Lr = K + blk_index(ib, kb) * BLK + r * LD;
            double acc = b[row];
#pragma unroll
            for (int j = 0; j < BS; ++j) acc = __builtin_fma(-Lr[j], b[kb * BS + j], acc);
            b[row] = acc;
        }
    }
    // backward: x_kb = Linv_kk' y_kb ; b_jb -= L_kb,jb' x_kb  (jb < kb)
    for (int kb = nb - 1; kb >= 0; --kb) {
        __syncthreads();
        double x = 0.0;
        if (t < BS) {
            const ldsd *Z = Linv + kb * BLK;
#pragma unroll
            for (int j = 0; j < BS; ++j) x = __builtin_fma(Z[j * LD + t], b[kb * BS + j], x);     // (Linv')[t][j] = Linv[j][t]
        }
        __syncthreads();
        if (t < BS) b[kb * BS + t] = x;
        __syncthreads();
        if (t < kb * BS) {
            const int jb = t / BS, c = t % BS;
            const ldsd *Lb = K + blk_index(kb, jb) * BLK;
            double acc = b[t];
#pragma unroll
            for (int j = 0; j < BS; ++j) acc = __builtin_fma(-Lb[j * LD + c], b[kb * BS + j], acc);     // (L_kb,jb')[c][j]
            b[t] = acc;
        }
    }
    __syncthreads();
}

// In place Z = L^-1 for the blocked factor at K (diagonal-block inverses in Linv), right to left by block
// column:  X_m = L_mj Z_jj (m > j);  Z_ij = -sum_{m=j+1..i} Z_im X_mj.  All 256 threads; nb <= 8.
__device__ __noinline__ void tri_invert_blocked(ldsd *K, const ldsd *Linv, int nb)
{
    const int wave = threadIdx.x >> 6;
    for (int j = nb - 1; j >= 0; --j) {
        __syncthreads();
        for (int m = j + 1 + wave; m < nb; m += 4) {                 // X_m = L_mj * Z_jj, in place
            ldsd *A = K + blk_index(m, j) * BLK;
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            c = block_mm<false, true>(A, Linv + j * BLK, c, false);
            tile_store(A, c);
        }
        __syncthreads();
        auto zij = [&](int i) {
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            for (int m = j + 1; m <= i; ++m) {
                const ldsd *Z = (m == i) ? Linv + i * BLK : K + blk_index(i, m) * BLK;
                c = block_mm<false, true>(Z, K + blk_index(m, j) * BLK, c, true);
            }
            return c;
        };
        const int i0 = j + 1 + wave, i1 = i0 + 4;                    // nb <= 8: at most two blocks per wave
        d4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
        if (i0 < nb) a0 = zij(i0);
        if (i1 < nb) a1 = zij(i1);
        __syncthreads();
        if (i0 < nb) tile_store(K + blk_index(i0, j) * BLK, a0);
        if (i1 < nb) tile_store(K + blk_index(i1, j) * BLK, a1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nb * BLK; e += THREADS) {          // diagonal blocks of Z
        const int j = e / BLK;
        K[blk_index(j, j) * BLK + (e - j * BLK)] = Linv[e];
    }
    __syncthreads();
}

// In place W = Z'Z (= (L L')^-1) for the lower-triangular block matrix Z at K; diagonal blocks of W come
// out full (both triangles).  Block row i of W needs block rows >= i of Z only: ascending i, in place.
__device__ __noinline__ void ztz_blocked(ldsd *K, int nb)
{
    const int wave = threadIdx.x >> 6;
    for (int i = 0; i < nb; ++i) {
        auto wij = [&](int j) {
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            for (int k = i; k < nb; ++k) c = block_mm<true, true>(K + blk_index(k, i) * BLK, K + blk_index(k, j) * BLK, c, false);
            return c;
        };
        const int j0 = wave, j1 = wave + 4;
        d4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
        if (j0 <= i) a0 = wij(j0);
        if (j1 <= i) a1 = wij(j1);
        __syncthreads();
        if (j0 <= i) tile_store(K + blk_index(i, j0) * BLK, a0);
        if (j1 <= i) tile_store(K + blk_index(i, j1) * BLK, a1);
        __syncthreads();
    }
}

}  // namespace wg
}  // namespace lqmpc
