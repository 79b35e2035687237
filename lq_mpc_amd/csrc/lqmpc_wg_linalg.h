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

__device__ __forceinline__ double rdlane(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// D = C + sgn * X * Y'  for 16x16 blocks X, Y in LDS (row stride LD); C/D in the MFMA tile layout
// (element r of the result: row lane/16 + 4r, column lane%16).  One wave.
__device__ __forceinline__ d4_t block_xyt(const double *X, const double *Y, d4_t c, bool negate)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a[t] = X[i * LD + 4 * t + kq];
        b[t] = Y[i * LD + 4 * t + kq];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a[t] : a[t], b[t], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ d4_t tile_load(const double *C)
{
    const int lane = threadIdx.x & 63, col = lane & 15, r0 = lane >> 4;
    d4_t c;
    c.x = C[(r0 + 0) * LD + col]; c.y = C[(r0 + 4) * LD + col]; c.z = C[(r0 + 8) * LD + col]; c.w = C[(r0 + 12) * LD + col];
    return c;
}
__device__ __forceinline__ void tile_store(double *C, d4_t c)
{
    const int lane = threadIdx.x & 63, col = lane & 15, r0 = lane >> 4;
    C[(r0 + 0) * LD + col] = c.x; C[(r0 + 4) * LD + col] = c.y; C[(r0 + 8) * LD + col] = c.z; C[(r0 + 12) * LD + col] = c.w;
}

// Wave 0, lanes 0..15: factor the diagonal block D (in place: lower triangle <- L, upper untouched) and
// write its inverse (lower triangular, upper part zero) to Dinv.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool diag_factor_invert(double *D, double *Dinv)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    double row[BS], inv[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) row[j] = D[i * LD + j];
    bool ok = true;
    double invd_own = 1.0;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        const double d = rdlane(row[k], k);
        ok = ok && (d > 0.0);
        const double s = frsqrt(d);
        if (i == k) invd_own = s;
        row[k] = (i > k) ? row[k] * s : ((i == k) ? d * s : 0.0);          // l_ik, l_kk = sqrt(d); rows above: unused
#pragma unroll
        for (int c = k + 1; c < BS; ++c) row[c] = __builtin_fma(-row[k], rdlane(row[k], c), row[c]);
    }
    // inverse by forward substitution on the identity, one row of Z = L^-1 per lane:
    //   z_ij = (delta_ij - sum_{k<i} l_ik z_kj) / l_ii, rows in order (row i needs rows k < i: readlane)
#pragma unroll
    for (int j = 0; j < BS; ++j) inv[j] = 0.0;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        // finalise row k of Z (in lane k), then eliminate it from the rows below
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            if (j > k) continue;
            double zkj = (j == k) ? 1.0 : 0.0;
            zkj = (i == k) ? (zkj + inv[j]) * invd_own : 0.0;       // lane k: (delta - accumulated) / l_kk, accumulated holds -sum
            if (i == k) inv[j] = zkj;
            const double zb = rdlane(inv[j], k);
            if (i > k) inv[j] = __builtin_fma(-row[k], zb, inv[j]);
        }
    }
    if (lane < BS) {
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            D[i * LD + j] = (j <= i) ? row[j] : 0.0;
            Dinv[i * LD + j] = (j <= i) ? inv[j] : 0.0;
        }
    }
    return ok;
}

// Blocked Cholesky of the nb x nb block matrix at K (LDS); Linv: nb diagonal-block inverses.  All 256 threads.
__device__ __forceinline__ bool chol_blocked(double *K, double *Linv, int nb, int *flag)
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
            double *A = K + blk_index(ib, kb) * BLK;
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
                double *C = K + blk_index(ib, jb) * BLK;
                d4_t c = tile_load(C);
                c = block_xyt(K + blk_index(ib, kb) * BLK, K + blk_index(jb, kb) * BLK, c, true);
                tile_store(C, c);
            }
    }
    __syncthreads();
    return *flag != 0;
}

// Solve (L L') x = b.  b: LDS vector of nb*16 doubles (in place).  tmp: LDS scratch of 16 doubles.
__device__ __forceinline__ void solve_blocked(const double *K, const double *Linv, int nb, double *b)
{
    const int t = threadIdx.x;
    // forward: y_kb = Linv_kk b_kb ; b_ib -= L_ib,kb y_kb
    for (int kb = 0; kb < nb; ++kb) {
        __syncthreads();
        double y = 0.0;
        if (t < BS) {
            const double *Z = Linv + kb * BLK + t * LD;
#pragma unroll
            for (int j = 0; j < BS; ++j) y = __builtin_fma(Z[j], b[kb * BS + j], y);
        }
        __syncthreads();
        if (t < BS) b[kb * BS + t] = y;
        __syncthreads();
        const int row = (kb + 1) * BS + t;
        if (row < nb * BS) {
            const int ib = row / BS, r = row % BS;
            const double *Lr = K + blk_index(ib, kb) * BLK + r * LD;
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
            const double *Z = Linv + kb * BLK;
#pragma unroll
            for (int j = 0; j < BS; ++j) x = __builtin_fma(Z[j * LD + t], b[kb * BS + j], x);     // (Linv')[t][j] = Linv[j][t]
        }
        __syncthreads();
        if (t < BS) b[kb * BS + t] = x;
        __syncthreads();
        if (t < kb * BS) {
            const int jb = t / BS, c = t % BS;
            const double *Lb = K + blk_index(kb, jb) * BLK;
            double acc = b[t];
#pragma unroll
            for (int j = 0; j < BS; ++j) acc = __builtin_fma(-Lb[j * LD + c], b[kb * BS + j], acc);     // (L_kb,jb')[c][j]
            b[t] = acc;
        }
    }
    __syncthreads();
}

}  // namespace wg
}  // namespace lqmpc
