// lqmpc_wg_linalg.h -- workgroup-cooperative dense linear algebra in LDS for the large-n kernel.
//
// One 256-thread workgroup (4 waves) owns one n x n system, n <= 128.  The symmetric working matrix K lives
// in LDS as the lower block triangle of 16x16 blocks: block (ib, jb), jb <= ib, at index ib(ib+1)/2 + jb,
// each block row-major with a row stride of 17 doubles (272 doubles per block).  The stride makes both
// access patterns of v_mfma_f64_16x16x4_f64 cheap: its A/B operand fetch (lane -> element [lane%16][4t +
// lane/16]) and its C/D tile (lane -> rows lane/16 + 4r, column lane%16).
//
// chol_blocked(): right-looking blocked Cholesky with look-ahead.  Per block column kb: the panel blocks X = A L_kk^-T are
//   formed as matrix products with the inverse of the diagonal block (4 waves); barrier; in the trailing update A_ij -= L_ik L_jk'
//   (rank-16 updates on the f64 MFMA) wave 0 takes the next diagonal block only, then factors it with one matrix row per lane (DPP
//   row broadcasts) and inverts it in the same pass, while waves 1..3 update the rest; barrier.
// After it: strict lower blocks hold L, diagonal blocks hold L_kk (lower triangle), Linv holds L_kk^-1.
#pragma once
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

// Arguments of a non-inlined device function arrive in vector registers and count as divergent: every loop on them would run under
// exec masks with its block addresses on the vector ALU (v_mad_u64_u32 ...).  They are workgroup-uniform here; say so.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class T> __device__ __forceinline__ T *uni(T *p) { return (T *)(size_t)__builtin_amdgcn_readfirstlane((int)(size_t)p); }

// D = C +- op(X) * op(Y)'  for 16x16 blocks X, Y in LDS (row stride LD), op = identity or transpose.
// C/D in the MFMA tile layout (element r of the result: row lane/16 + 4r, column lane%16).  One wave.
// Operand fetch of v_mfma_f64_16x16x4_f64: lane l supplies A[l%16][4t + l/16] and B'[l%16][4t + l/16].
// kt: number of 4-wide slabs of the contraction index that are not all zero (kt < 4: the operands' columns 4 kt .. 15 are padding)
template <bool XT, bool YT>
__device__ __forceinline__ d4_t block_mm(const ldsd *X, const ldsd *Y, d4_t c, bool negate, int kt = 4)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < kt) {
            a[t] = XT ? X[(4 * t + kq) * LD + i] : X[i * LD + 4 * t + kq];
            b[t] = YT ? Y[(4 * t + kq) * LD + i] : Y[i * LD + 4 * t + kq];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < kt) c = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a[t] : a[t], b[t], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ d4_t block_xyt(const ldsd *X, const ldsd *Y, d4_t c, bool negate, int kt = 4) { return block_mm<false, false>(X, Y, c, negate, kt); }

// The operands of one block product as the MFMA wants them (8 doubles per lane), so that a sum of products can fetch the next
// pair of blocks while the matrix core works on the current one (one wavefront per SIMD here: nothing else hides the LDS trip).
struct BOps { double a[4], b[4]; };
template <bool XT, bool YT>
__device__ __forceinline__ void load_ops(const ldsd *X, const ldsd *Y, BOps &o)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        o.a[t] = XT ? X[(4 * t + kq) * LD + i] : X[i * LD + 4 * t + kq];
        o.b[t] = YT ? Y[(4 * t + kq) * LD + i] : Y[i * LD + 4 * t + kq];
    }
}
__device__ __forceinline__ d4_t mfma_ops(const BOps &o, d4_t c, bool negate)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -o.a[t] : o.a[t], o.b[t], c, 0, 0, 0);
    return c;
}
// c +- sum_{m = m0 .. m1-1} op(X(m)) op(Y(m))', the loads of product m+1 issued before the MFMAs of product m
template <bool XT, bool YT, class FX, class FY>
__device__ __forceinline__ d4_t block_sum(int m0, int m1, FX X, FY Y, d4_t c, bool negate)
{
    if (m0 >= m1) return c;
    BOps cur, nxt;
    load_ops<XT, YT>(X(m0), Y(m0), cur);
    for (int m = m0; m < m1; m += 2) {                           // two products per round: the operand sets ping-pong, no copies
        const int mn = (m + 1 < m1) ? m + 1 : m;                 // (an odd tail re-reads its own operands: harmless)
        load_ops<XT, YT>(X(mn), Y(mn), nxt);
        __builtin_amdgcn_sched_barrier(0);                       // (the scheduler otherwise sinks the loads to their first use)
        c = mfma_ops(cur, c, negate);
        __builtin_amdgcn_sched_barrier(0);
        if (m + 1 >= m1) break;
        const int mc = (m + 2 < m1) ? m + 2 : m + 1;
        load_ops<XT, YT>(X(mc), Y(mc), cur);
        __builtin_amdgcn_sched_barrier(0);
        c = mfma_ops(nxt, c, negate);
        __builtin_amdgcn_sched_barrier(0);
    }
    return c;
}

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

// a_j += (lane c0+j's x, broadcast over the row) * y for four consecutive lanes c0 .. c0+3 behind one s_nop: four columns of the
// factorisation's update a_ic -= l_ik l_ck (x = the pivot column, one entry per lane; y = -l_ik)
#define LQMPC_COLS4_CASES(X) X(0, 1, 2, 3) X(1, 2, 3, 4) X(2, 3, 4, 5) X(3, 4, 5, 6) X(4, 5, 6, 7) X(5, 6, 7, 8) X(6, 7, 8, 9) X(7, 8, 9, 10) \
    X(8, 9, 10, 11) X(9, 10, 11, 12) X(10, 11, 12, 13) X(11, 12, 13, 14) X(12, 13, 14, 15)
__device__ __forceinline__ void fmac_rowb_cols4(double &a0, double &a1, double &a2, double &a3, double x, double y, int c0)
{
    switch (c0) {
#define LQMPC_X(C0, C1, C2, C3) case C0: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %5 row_newbcast:" #C0 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:" #C1 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:" #C2 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:" #C3 " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y)); break;
        LQMPC_COLS4_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// a_j += (lane c's x, broadcast over the row) * y_j for three accumulators behind one s_nop: one state component into three dot products
__device__ __forceinline__ void fmac_rowb_3y(double &a0, double &a1, double &a2, double x, double y0, double y1, double y2, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %4 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %3, %5 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %2, %3, %6 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(y0), "v"(y1), "v"(y2)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// a_s += sum over four consecutive lanes c0 .. c0+3 of (lane's x, broadcast over the row) * y_k,s behind one s_nop: four columns of a
// matrix-vector product whose vector sits one entry per lane (RB = 1 or 2 accumulators: the row slots of the 16-lane-row kernels)
#define LQMPC_LANES4_CASES(X) X(0, 1, 2, 3) X(4, 5, 6, 7) X(8, 9, 10, 11) X(12, 13, 14, 15)
__device__ __forceinline__ void fmac_rowb_lanes4(double &a0, double x, double y0, double y1, double y2, double y3, int c0)
{
    switch (c0) {
#define LQMPC_X(C0, C1, C2, C3) case C0: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C0 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:" #C1 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:" #C2 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:" #C3 " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0) : "v"(x), "v"(y0), "v"(y1), "v"(y2), "v"(y3)); break;
        LQMPC_LANES4_CASES(LQMPC_X)
#undef LQMPC_X
    }
}
__device__ __forceinline__ void fmac_rowb_lanes4x2(double &a0, double &a1, double x, double y00, double y01, double y10, double y11,
                                                   double y20, double y21, double y30, double y31, int c0)
{
    switch (c0) {
#define LQMPC_X(C0, C1, C2, C3) case C0: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:" #C0 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:" #C0 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %2, %5 row_newbcast:" #C1 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %2, %6 row_newbcast:" #C1 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %2, %7 row_newbcast:" #C2 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %2, %8 row_newbcast:" #C2 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %0, %2, %9 row_newbcast:" #C3 " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %2, %10 row_newbcast:" #C3 " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1) : "v"(x), "v"(y00), "v"(y01), "v"(y10), "v"(y11), "v"(y20), "v"(y21), "v"(y30), "v"(y31)); break;
        LQMPC_LANES4_CASES(LQMPC_X)
#undef LQMPC_X
    }
}

// two / three self-updates (acc += lane c's acc * y) behind one s_nop: the tail of a pivot's own column group plus the right-hand side
__device__ __forceinline__ void fmac_rowb_self2(double &a0, double &a1, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1) : "v"(y)); break;
        LQMPC_ROWB_CASES(LQMPC_X)
#undef LQMPC_X
    }
}
__device__ __forceinline__ void fmac_rowb_self3(double &a0, double &a1, double &a2, double y, int c)
{
    switch (c) {
#define LQMPC_X(C) case C: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %1, %1, %3 row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t" \
                                        "v_fmac_f64_dpp %2, %2, %3 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                        : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(y)); break;
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

// compile-time loop: f(integral_constant<int, I>) for I = I0 .. I1-1.  The pivot loops below use it instead of "#pragma unroll" because
// the unroller prices the lane switch of every rowb helper at all 16 cases and then refuses to unroll fully -- which would leave the
// row registers indexed at run time (s_set_gpr_idx) and every broadcast behind a jump table.
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f) { sfor<I0, I1>(f); }

// One wave: factor the diagonal block D (in place: lower triangle <- L, upper <- 0) and write its inverse
// (lower triangular) to Dinv.  Returns false on a non-positive pivot.
// One matrix row per lane (lanes 16..63 mirror lanes 0..15).  Per column k: broadcast the pivot, rsqrt, scale, then one
// v_fmac_f64_dpp per remaining column (row_newbcast brings l_ck) -- 16 dependent steps, each one rsqrt chain long.  The inverse
// rides in the shadow of that chain: the same row operations applied to the identity (row i of Z = L^-1 in lane i),
//     z_k <- s_k z_k,   z_i <- z_i - l_ik z_k  (i > k),
// are one v_fmac_f64_dpp per entry of z_k with the per-lane multiplier m_i = (i == k) ? s_k - 1 : (i > k) ? -l_ik s_k : 0 and need
// neither LDS nor the finished L (k + 1 entries at step k: 136 updates, issued while the next pivot's rsqrt is in flight).
__device__ __forceinline__ bool diag_factor_invert(ldsd *D, ldsd *Dinv)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    double row[BS], z[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) { row[j] = D[i * LD + j]; z[j] = (j == i) ? 1.0 : 0.0; }
    double d = rowb(row[0], 0);
    bool ok = d > 0.0;
    double y = __builtin_amdgcn_rsq(d);
    static_for<0, BS>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // the chain: pivot k's rsqrt (issued one step ago) -> l_.k -> column k+1 -> pivot k+1 -> its rsqrt
        const double e = __builtin_fma(-d * y, y, 1.0);
        const double s = __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
        row[k] *= s;                                             // l_kk = d / sqrt(d) in lane k, l_ik = a_ik / l_kk below it
        const double nl = -row[k];
        const double m = (i == k) ? s - 1.0 : ((i > k) ? nl * s : 0.0);
        if constexpr (k + 1 < BS) {
            fmac_rowb(row[k + 1], row[k], nl, k + 1);            // a_i,k+1 -= l_ik l_k+1,k
            dpp_settle(row[k + 1]);
            d = rowb(row[k + 1], k + 1);
            ok = ok && (d > 0.0);
            y = __builtin_amdgcn_rsq(d);
        }
        __builtin_amdgcn_sched_barrier(0);
        // in the shadow of that rsqrt: the other columns (a_ic -= l_ik l_ck) and the inverse's rows (z_k <- s z_k, z_i -= l_ik z_k)
        constexpr int G = (BS - k - 2 > 0) ? (BS - k - 2) / 4 : 0;
        static_for<0, G>([&](auto gc) {
            constexpr int c = k + 2 + 4 * decltype(gc)::value;
            fmac_rowb_cols4(row[c], row[c + 1], row[c + 2], row[c + 3], row[k], nl, c);
        });
        static_for<k + 2 + 4 * G, BS>([&](auto cc) { constexpr int c = decltype(cc)::value; fmac_rowb(row[c], row[k], nl, c); });
        static_for<0, (k + 1) / 4>([&](auto gc) {
            constexpr int j = 4 * decltype(gc)::value;
            fmac_rowb_self4(z[j], z[j + 1], z[j + 2], z[j + 3], m, k);
        });
        static_for<((k + 1) / 4) * 4, k + 1>([&](auto jc) { fmac_rowb_self(z[decltype(jc)::value], m, k); });
        __builtin_amdgcn_sched_barrier(0);
    });
    if (lane < BS) {
#pragma unroll
        for (int j = 0; j < BS; ++j) { D[i * LD + j] = (j <= i) ? row[j] : 0.0; Dinv[i * LD + j] = z[j]; }
    }
    return ok;
}

// One wave: solve S x = r for an m x m SPD system, m <= 16, S in one block (row stride LD, rows/columns >= m
// hold the identity), r and x in LDS (x may alias r).  T: unused (a block of LDS scratch of the Cholesky version).  Returns false
// on a non-positive pivot.  Row per lane.  Compiled for MB = 4, 8, 12, 16 pivots (the identity rows up to MB eliminate to
// themselves), so that no update sits behind a branch on m.
template <int MB>
__device__ __forceinline__ bool small_spd_solve_mb(const ldsd *S, ldsd *T, const ldsd *r, ldsd *x, int m)
{
    (void)T;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    double row[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) row[j] = S[i * LD + j];
    double rhs = r[i];
    // Gauss-Jordan on [S | r] without pivoting (S is SPD: the pivots are those of its Cholesky factorisation squared): after pivot k
    // column k is e_k in every row, so the right-hand side ends up as the solution and there is no back substitution (the
    // Cholesky route needed the factor transposed through LDS and a second 16-step chain).  Per pivot: broadcast, reciprocal with
    // one Newton step, the per-lane multiplier g = 1/d - 1 in the pivot row and -s_ik/d elsewhere, then one DPP FMA per live column;
    // the next pivot's column comes first and its reciprocal is under way while the rest is updated.
    double d = rowb(row[0], 0);
    bool ok = d > 0.0;
    double inv = frcp1(d);
    static_for<0, MB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double g = (i == k) ? inv - 1.0 : -row[k] * inv;
        if constexpr (k + 1 < MB) {
            fmac_rowb_self(row[k + 1], g, k);
            dpp_settle(row[k + 1]);
            d = rowb(row[k + 1], k + 1);
            ok = ok && (d > 0.0);
            inv = frcp1(d);
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NR = (MB - k - 2 > 0) ? MB - k - 2 : 0, G4 = NR / 4;
        static_for<0, G4>([&](auto gc) {
            constexpr int c = k + 2 + 4 * decltype(gc)::value;
            fmac_rowb_self4(row[c], row[c + 1], row[c + 2], row[c + 3], g, k);
        });
        static_for<k + 2 + 4 * G4, MB>([&](auto cc) { fmac_rowb_self(row[decltype(cc)::value], g, k); });
        fmac_rowb_self(rhs, g, k);
        __builtin_amdgcn_sched_barrier(0);
    });
    if (lane < BS) x[i] = (i < m) ? rhs : 0.0;
    return ok;
}
__device__ __forceinline__ bool small_spd_solve(const ldsd *S, ldsd *T, const ldsd *r, ldsd *x, int m)
{
    if (m <= 4) return small_spd_solve_mb<4>(S, T, r, x, m);     // uniform
    if (m <= 8) return small_spd_solve_mb<8>(S, T, r, x, m);
    if (m <= 12) return small_spd_solve_mb<12>(S, T, r, x, m);
    return small_spd_solve_mb<16>(S, T, r, x, m);
}

// Blocked Cholesky of the nb x nb block matrix at K (LDS); Linv: nb diagonal-block inverses.  All 256 threads.
// Right-looking with look-ahead: in the trailing update of block column kb wave 0 takes the next diagonal block only and goes
// straight on to factor and invert it (the 16-pivot chain is the critical path of the whole factorisation), waves 1..3 share the
// other blocks, operands of the next block fetched while the matrix core works on the current one.  Two barriers per block column.
__device__ __noinline__ bool chol_blocked(ldsd *K, ldsd *Linv, int nb, ldsi *flag)
{
    K = uni(K); Linv = uni(Linv); nb = uni(nb); flag = uni(flag);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: loops and block addresses on the scalar unit
    if (threadIdx.x == 0) *flag = 1;
    __syncthreads();
    if (wave == 0) {
        const bool ok = diag_factor_invert(K, Linv);
        if (!ok && (threadIdx.x & 63) == 0) *flag = 0;
    }
    for (int kb = 0; kb < nb - 1; ++kb) {
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
        if (wave == 0) {
            ldsd *C = K + blk_index(kb + 1, kb + 1) * BLK, *X = K + blk_index(kb + 1, kb) * BLK;
            d4_t c = tile_load(C);
            c = block_xyt(X, X, c, true);
            tile_store(C, c);
            const bool ok = diag_factor_invert(C, Linv + (kb + 1) * BLK);
            if (!ok && (threadIdx.x & 63) == 0) *flag = 0;
        } else {
            auto step = [&](int &ib, int &jb, int cnt) { for (int q = 0; q < cnt; ++q) if (++jb > ib) { ++ib; jb = kb + 1; } };
            int ib = kb + 1, jb = kb + 1;
            step(ib, jb, wave);                                  // blocks 1, 2, 3 (+ 3 each round) of the trailing triangle in storage order
            if (ib < nb) {
                BOps cur, nxt;
                load_ops<false, false>(K + blk_index(ib, kb) * BLK, K + blk_index(jb, kb) * BLK, cur);
                d4_t cc = tile_load(K + blk_index(ib, jb) * BLK), cn;
                while (ib < nb) {
                    int ib2 = ib, jb2 = jb;
                    step(ib2, jb2, 3);
                    const int ibn = (ib2 < nb) ? ib2 : ib, jbn = (ib2 < nb) ? jb2 : jb;      // (the last round re-reads its own block)
                    load_ops<false, false>(K + blk_index(ibn, kb) * BLK, K + blk_index(jbn, kb) * BLK, nxt);
                    cn = tile_load(K + blk_index(ibn, jbn) * BLK);
                    __builtin_amdgcn_sched_barrier(0);
                    cc = mfma_ops(cur, cc, true);
                    __builtin_amdgcn_sched_barrier(0);
                    tile_store(K + blk_index(ib, jb) * BLK, cc);
                    cur = nxt; cc = cn; ib = ib2; jb = jb2;
                }
            }
        }
    }
    __syncthreads();
    return *flag != 0;
}

// chol_blocked() with the inverse of the factor riding along: on return K holds Z = L^-1 (strict lower blocks; the diagonal blocks
// of Z are in Linv and are copied into K), ready for ztz_blocked().  Top-down by block rows,
//     Z_ij = -Z_ii S_ij,   S_ij = sum_{m=j}^{i-1} L_im Z_mj   (Z_jj = L_jj^-1),
// and the work is done by waves 1..3 in the time they would otherwise wait for wave 0's diagonal block (two thirds of the
// factorisation: the trailing update of block column kb is short against the 16-pivot chain of block kb+1).  In the trailing phase of
// block column kb a wave finishes row kb of Z (Z_kb,kb = L_kb,kb^-1 exists since the phase before) and forms the sums of row kb+1
// (that row of L is complete after panel kb); the sums wait in Sc (one block per column) for the next phase because Z_kb+1,kb+1 is
// being computed by wave 0 right now.  A block column always belongs to the same wave: a sum reads Z blocks of its own column
// only, i.e. blocks the same wave wrote (the LDS executes a wavefront's accesses in order), and nobody else reads or writes the
// blocks it overwrites (row kb left of the diagonal: dead once its sums exist), so the two barriers per block column of the
// factorisation are all the synchronisation there is.  Sc: nb - 1 blocks of LDS scratch.
__device__ __noinline__ bool chol_inverse_blocked(ldsd *K, ldsd *Linv, int nb, ldsi *flag, ldsd *Sc)
{
    K = uni(K); Linv = uni(Linv); nb = uni(nb); flag = uni(flag); Sc = uni(Sc);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *flag = 1;
    __syncthreads();
    if (wave == 0) {
        const bool ok = diag_factor_invert(K, Linv);
        if (!ok && (threadIdx.x & 63) == 0) *flag = 0;
    }
    for (int kb = 0; kb < nb - 1; ++kb) {
        __syncthreads();
        for (int ib = kb + 1 + wave; ib < nb; ib += 4) {             // panel: X_ib = A_ib,kb * Linv_kk', in place
            ldsd *A = K + blk_index(ib, kb) * BLK;
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            c = block_xyt(A, Linv + kb * BLK, c, false);
            tile_store(A, c);
        }
        __syncthreads();
        if (wave == 0) {
            ldsd *C = K + blk_index(kb + 1, kb + 1) * BLK, *X = K + blk_index(kb + 1, kb) * BLK;
            d4_t c = tile_load(C);
            c = block_xyt(X, X, c, true);
            tile_store(C, c);
            const bool ok = diag_factor_invert(C, Linv + (kb + 1) * BLK);
            if (!ok && (threadIdx.x & 63) == 0) *flag = 0;
        } else {
            // trailing update: A_ib,jb -= X_ib X_jb'   (kb < jb <= ib), the diagonal block kb+1 excepted (wave 0)
            auto step = [&](int &ib, int &jb, int cnt) { for (int q = 0; q < cnt; ++q) if (++jb > ib) { ++ib; jb = kb + 1; } };
            int ib = kb + 1, jb = kb + 1;
            step(ib, jb, wave);
            if (ib < nb) {
                BOps cur, nxt;
                load_ops<false, false>(K + blk_index(ib, kb) * BLK, K + blk_index(jb, kb) * BLK, cur);
                d4_t cc = tile_load(K + blk_index(ib, jb) * BLK), cn;
                while (ib < nb) {
                    int ib2 = ib, jb2 = jb;
                    step(ib2, jb2, 3);
                    const int ibn = (ib2 < nb) ? ib2 : ib, jbn = (ib2 < nb) ? jb2 : jb;
                    load_ops<false, false>(K + blk_index(ibn, kb) * BLK, K + blk_index(jbn, kb) * BLK, nxt);
                    cn = tile_load(K + blk_index(ibn, jbn) * BLK);
                    __builtin_amdgcn_sched_barrier(0);
                    cc = mfma_ops(cur, cc, true);
                    __builtin_amdgcn_sched_barrier(0);
                    tile_store(K + blk_index(ib, jb) * BLK, cc);
                    cur = nxt; cc = cn; ib = ib2; jb = jb2;
                }
            }
            // the inverse: my block columns j <= kb (columns go to the three waves in snake order 1 2 3 3 2 1 1 2 ...: the sums of
            // column j have kb + 1 - j products, and the snake keeps the three loads within a product or two of each other)
            for (int j = 0; j <= kb; ++j) {
                const int r6 = j % 6;
                if (1 + (r6 < 3 ? r6 : 5 - r6) != wave) continue;
                if (j < kb) {                                        // finish row kb: Z_kb,j = -Z_kb,kb S_kb,j
                    d4_t z = {0.0, 0.0, 0.0, 0.0};
                    z = block_mm<false, true>(Linv + kb * BLK, Sc + j * BLK, z, true);
                    tile_store(K + blk_index(kb, j) * BLK, z);
                }
                d4_t sacc = {0.0, 0.0, 0.0, 0.0};                     // S_kb+1,j = sum_{m=j}^{kb} L_kb+1,m Z_mj
                sacc = block_sum<false, true>(j, kb + 1, [&](int m) { return K + blk_index(kb + 1, m) * BLK; },
                                              [&](int m) { return (m == j) ? Linv + j * BLK : K + blk_index(m, j) * BLK; }, sacc, false);
                tile_store(Sc + j * BLK, sacc);
            }
        }
    }
    __syncthreads();
    for (int j = wave; j < nb - 1; j += 4) {                         // the last row: Z_nb-1,j = -Z_nb-1,nb-1 S_nb-1,j
        d4_t z = {0.0, 0.0, 0.0, 0.0};
        z = block_mm<false, true>(Linv + (nb - 1) * BLK, Sc + j * BLK, z, true);
        tile_store(K + blk_index(nb - 1, j) * BLK, z);
    }
    for (int e = threadIdx.x; e < nb * BLK; e += THREADS) {          // diagonal blocks of Z
        const int j = e / BLK;
        K[blk_index(j, j) * BLK + (e - j * BLK)] = Linv[e];
    }
    __syncthreads();
    return *flag != 0;
}

// Solve (L L') x = b.  b: LDS vector of nb*16 doubles (in place).  tmp: LDS scratch of 16 doubles.
__device__ __noinline__ void solve_blocked(const ldsd *K, const ldsd *Linv, int nb, ldsd *b)
{
    K = uni(K); Linv = uni(Linv); nb = uni(nb); b = uni(b);
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
// The sums are pipelined (block_sum) and dealt so that a wave gets a short and a long one (i - j products for block row i).
__device__ __noinline__ void tri_invert_blocked(ldsd *K, const ldsd *Linv, int nb)
{
    K = uni(K); Linv = uni(Linv); nb = uni(nb);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: loops and block addresses on the scalar unit
    __syncthreads();
    for (int j = nb - 1; j >= 0; --j) {
        // (no barrier here: the panel touches block column j only, which the stores of column j+1 just before do not)
        {                                                            // X_m = L_mj * Z_jj, in place; at most two blocks per wave
            const int m0 = j + 1 + wave, m1 = m0 + 4;
            ldsd *A0 = K + blk_index(m0 < nb ? m0 : j, j) * BLK, *A1 = K + blk_index(m1 < nb ? m1 : j, j) * BLK;
            BOps o0, o1;
            if (m0 < nb) load_ops<false, true>(A0, Linv + j * BLK, o0);
            if (m1 < nb) load_ops<false, true>(A1, Linv + j * BLK, o1);
            __builtin_amdgcn_sched_barrier(0);
            if (m0 < nb) tile_store(A0, mfma_ops(o0, d4_t{0.0, 0.0, 0.0, 0.0}, false));
            if (m1 < nb) tile_store(A1, mfma_ops(o1, d4_t{0.0, 0.0, 0.0, 0.0}, false));
        }
        __syncthreads();
        auto zij = [&](int i) {
            d4_t c = {0.0, 0.0, 0.0, 0.0};
            return block_sum<false, true>(j + 1, i + 1,
                                          [&](int m) { return (m == i) ? Linv + i * BLK : K + blk_index(i, m) * BLK; },
                                          [&](int m) { return K + blk_index(m, j) * BLK; }, c, true);
        };
        const int i0 = j + 1 + wave, i1 = nb - 1 - wave;             // nb <= 8: block rows j+1 .. nb-1 are covered by 4 such pairs
        d4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
        if (i0 <= i1) a0 = zij(i0);
        if (i0 < i1) a1 = zij(i1);
        __syncthreads();
        if (i0 <= i1) tile_store(K + blk_index(i0, j) * BLK, a0);
        if (i0 < i1) tile_store(K + blk_index(i1, j) * BLK, a1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nb * BLK; e += THREADS) {          // diagonal blocks of Z
        const int j = e / BLK;
        K[blk_index(j, j) * BLK + (e - j * BLK)] = Linv[e];
    }
    __syncthreads();
}

// In place W = Z'Z (= (L L')^-1) for the lower-triangular block matrix Z at K; diagonal blocks of W come
// out full (both triangles).  W_ij = sum_{k >= i} Z_ki' Z_kj (j <= i): every block of W is computed from Z before any is
// stored -- 36 blocks at nb = 8, nine accumulators per wave (dealt round-robin in storage order: 33 / 31 / 29 / 27 products)
// -- so the whole product costs two barriers.
#ifndef LQMPC_ZTZ_T
#define LQMPC_ZTZ_T(k) do { } while (0)
#endif
constexpr int ZTZ_SLOTS = 9;             // ceil(36 / 4)
__device__ __noinline__ void ztz_blocked(ldsd *K, int nb)
{
    K = uni(K); nb = uni(nb);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: loops and block addresses on the scalar unit
    const int nblk = nb * (nb + 1) / 2;
    d4_t acc[ZTZ_SLOTS];
    LQMPC_ZTZ_T(0);
    {
        int i = 0, j = 0;                                            // block (i, j) of storage index q
        for (int s = 0; s < wave; ++s) { if (++j > i) { ++i; j = 0; } }
#pragma unroll
        for (int s = 0; s < ZTZ_SLOTS; ++s) {
            acc[s] = d4_t{0.0, 0.0, 0.0, 0.0};
            if (wave + 4 * s < nblk)
                acc[s] = block_sum<true, true>(i, nb, [&](int k) { return K + blk_index(k, i) * BLK; },
                                               [&](int k) { return K + blk_index(k, j) * BLK; }, acc[s], false);
            for (int r = 0; r < 4; ++r) { if (++j > i) { ++i; j = 0; } }
        }
    }
    LQMPC_ZTZ_T(1);
    __syncthreads();
    LQMPC_ZTZ_T(2);
#pragma unroll
    for (int s = 0; s < ZTZ_SLOTS; ++s)
        if (wave + 4 * s < nblk) tile_store(K + (wave + 4 * s) * BLK, acc[s]);
    __syncthreads();
    LQMPC_ZTZ_T(3);
}

}  // namespace wg
}  // namespace lqmpc
