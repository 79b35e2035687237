// lqmpc_wg.hip -- large-n kernel: one 256-thread workgroup per (A,B) instance, 32 < n <= 128 (C5: n = 120).
//
// Where things live (gfx950, 160 KiB LDS and 512 KiB of registers per CU, one workgroup per CU):
//   registers  the constant matrix P = 2(Gamma'Qbar Gamma + Rbar): the lower block triangle of 16x16
//              blocks, 39 doubles per thread (flat over the workgroup, same order as the LDS image);
//              one element of every solver vector per thread (thread i <-> row i);
//   LDS        W = P^-1 in the same block layout (the matrix every step works with), G = -W Fq (n x nx),
//              the batch-shared data (Q, R, P_T, box, references, plant), a small workspace for the
//              active-set systems, work vectors, A^m B while condensing;
//   HBM        inputs and results only.
// Linear algebra: lqmpc_wg_linalg.h -- blocked Cholesky, triangular inverse and Z'Z whose block products run
// on v_mfma_f64_16x16x4_f64 (the one place of this path that is a dense contraction); condensing and
// G = -W Fq are block products on the same instruction.
//
// Algorithm per QP.  Same iterates as the primal-dual active-set method of the other
// kernels, but solved from the dual side because P is constant per instance and the active set is small:
//   v_unc = G x + v_r;  inside the box -> done (presolve);
//   active set A with signs s (warm start: the previous step's set shifted by one stage):
//       (W_AA) lam = v_unc,A - s h_A,   v = v_unc - W[:,A] lam,   gradient on A = -lam, zero elsewhere,
//   i.e. an m x m system (m = |A|, gathered from W) instead of an n x n one; update A from the signs of lam
//   and the box violations of v; repeat until A is stable.
// If A outgrows the workspace or the iteration cycles, the QP goes through the staged interior-point method
// with primal active-set finishing (n x n factorisations in the W region, P from registers), after which W
// is rebuilt.
//
// References and asymmetric boxes enter as in the other kernels: u = v + centre, |v| <= h, and the constant
// part of the linear term qr = 2 gref + P centre (gref by a costate recursion on vectors) shifts the
// unconstrained minimiser by v_r = -W qr.
#include "lqmpc_wg_linalg.h"
#include <cstdio>
#include <type_traits>

namespace lqmpc {

using namespace wg;

#ifdef LQMPC_WG_PROF
__device__ long long g_wg_prof[48];
#define PROF(k) do { const long long now_ = clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) g_wg_prof[k] += now_ - prof_t; prof_t = clock64(); } while (0)
#define PROF_START long long prof_t = clock64()
#else
#define PROF(k) do { } while (0)
#define PROF_START do { } while (0)
#endif

constexpr int PREG = 39;                 // ceil(36 blocks * 272 doubles / 256 threads)
constexpr int LDS_DOUBLES = 160 * 1024 / 8 - 8;

struct WgOff {                            // LDS offsets in doubles
    int K, Linv, G, vb, vw, act, lam, list, red, xs, AB, SH, shn;
    int U;                                // union: condensing scratch | active-set workspace | state trajectory
    int X, Xf, Lam;                       //   condensing: a block-row matrix (Q M | P_T M | Fq), A^k, costate
    int T, S, smax;                       //   scratch block, gathered system (smax x smax blocks at most)
    int Xs;                               //   predicted states for V_N
    int total;
};

__host__ __device__ inline WgOff wg_offsets(int nx, int nu, int N)
{
    const int n = N * nu, nb = (n + BS - 1) / BS, np = nb * BS;
    WgOff o;
    int c = 0;
    o.K = c;    c += nb * (nb + 1) / 2 * BLK;
    o.Linv = c; c += nb * BLK;             // also the block-row image of the A^m B while condensing
    o.G = c;    c += np * nx;
    o.vb = c;   c += np;
    o.vw = c;   c += np;
    o.act = c;  c += np;
    o.lam = c;  c += np;
    o.list = c; c += np / 2;
    o.red = c;  c += 32;
    o.xs = c;   c += 2 * nx + 8;
    o.AB = c;   c += nx * nx + nx * nu;
    o.shn = 3 * nx * nx + nu * nu + 2 * nu + (nx + nu) * N + nx * nu;     // everything of the shared block before x0s
    o.SH = c;   c += o.shn + (o.shn & 1);
    o.U = c;
    o.X = c; o.Xf = o.X + nb * BLK; o.Lam = o.Xf + N * nx * nx;
    int usize = o.Lam + 2 * nx * nx - o.U;
    o.T = c; o.S = c + BLK; o.Xs = c;
    int smax = 0;
    while (smax < nb && o.U + ((smax + 1) * (smax + 2) / 2 + 1) * BLK <= LDS_DOUBLES) ++smax;
    o.smax = smax;
    const int ssize = (smax * (smax + 1) / 2 + 1) * BLK;
    if (ssize > usize) usize = ssize;
    if ((N + 1) * nx > usize) usize = (N + 1) * nx;
    o.total = o.U + usize;
    return o;
}

size_t wg_lds_bytes(int nx, int nu, int N) { return (size_t)wg_offsets(nx, nu, N).total * sizeof(double) + 64; }

bool wg_supported(const KParams &p, const double *lb, const double *ub)
{
    (void)lb; (void)ub;
    if (p.n <= 32 || p.n > 128 || p.nx > 16 || p.nu > 8) return false;      // (nx, nu: the closed-loop update lives in one wavefront)
    const WgOff o = wg_offsets(p.nx, p.nu, p.N);
    return o.total <= LDS_DOUBLES && o.smax >= 1;
}

// ---- workgroup reductions over the threads that own a row (others pass the neutral element) ----
// Within a wavefront: four DPP row rotations give every lane its 16-lane row's result, four v_readlane pairs combine the rows (a
// __shfl_xor butterfly is six trips through the LDS crossbar: 900 ticks per block_max against 550, tools/ubench/reduce_rate.hip).
// Across the four wavefronts: one LDS slot per wave and ONE barrier; consecutive reductions alternate between two sets of slots, so a
// fast wave's next write cannot overtake a slow wave's read of this one (any other barrier in between orders them as well).
struct Red {
    ldsd *p;                  // 32 doubles: [0..8) the two sets of wave slots, [8] the factorisation's flag, [12..14) rank counters (ints), [16..20) flag slots, [24..32) interior_steps
    int par;
};
__device__ __forceinline__ double row_ror(double x, int n)       // n in {1, 2, 4, 8}: rotate within each 16-lane row
{
    long long v = __double_as_longlong(x);
    switch (n) {
    case 1: v = __builtin_amdgcn_update_dpp(0ll, v, 0x121, 0xF, 0xF, true); break;
    case 2: v = __builtin_amdgcn_update_dpp(0ll, v, 0x122, 0xF, 0xF, true); break;
    case 4: v = __builtin_amdgcn_update_dpp(0ll, v, 0x124, 0xF, 0xF, true); break;
    default: v = __builtin_amdgcn_update_dpp(0ll, v, 0x128, 0xF, 0xF, true); break;
    }
    return __longlong_as_double(v);
}
__device__ __forceinline__ double wave_sum(double x)
{
    x += row_ror(x, 8); x += row_ror(x, 4); x += row_ror(x, 2); x += row_ror(x, 1);
    return (rdlane(x, 0) + rdlane(x, 16)) + (rdlane(x, 32) + rdlane(x, 48));
}
__device__ __forceinline__ double wave_max(double x)
{
    x = fmax(x, row_ror(x, 8)); x = fmax(x, row_ror(x, 4)); x = fmax(x, row_ror(x, 2)); x = fmax(x, row_ror(x, 1));
    return fmax(fmax(rdlane(x, 0), rdlane(x, 16)), fmax(rdlane(x, 32), rdlane(x, 48)));
}
__device__ __forceinline__ double block_sum(double x, Red &R)
{
    x = wave_sum(x);
    ldsd *s = R.p + 4 * R.par;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    R.par ^= 1;
    return (s[0] + s[1]) + (s[2] + s[3]);
}
__device__ __forceinline__ double block_max(double x, Red &R)
{
    x = wave_max(x);
    ldsd *s = R.p + 4 * R.par;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    R.par ^= 1;
    return fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}
__device__ __forceinline__ bool block_any(bool f, Red &R)        // (__syncthreads_or costs two barriers and an LDS reduction: 580 ticks against 300)
{
    const int w = __ballot(f) != 0ull;
    ldsi *s = (ldsi *)(R.p + 16) + 4 * R.par;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w;
    __syncthreads();
    R.par ^= 1;
    return (s[0] | s[1] | s[2] | s[3]) != 0;
}
// three predicates in one reduction: bit k of the result = any thread's f_k
__device__ __forceinline__ int block_any3(bool f0, bool f1, bool f2, Red &R)
{
    const int w = (__ballot(f0) != 0ull ? 1 : 0) | (__ballot(f1) != 0ull ? 2 : 0) | (__ballot(f2) != 0ull ? 4 : 0);
    ldsi *s = (ldsi *)(R.p + 16) + 4 * R.par;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w;
    __syncthreads();
    R.par ^= 1;
    return s[0] | s[1] | s[2] | s[3];
}

// acc + sum_{y<len} a[y*sa] * b[y*sb] for short runtime lengths (nx, nu <= 16): loads issued eight / four at a
// time so that their LDS latencies overlap instead of adding up
__device__ __forceinline__ double ldot(const ldsd *a, int sa, const ldsd *b, int sb, int len, double acc = 0.0)
{
    int y = 0;
    for (; y + 8 <= len; y += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { av[k] = a[(y + k) * sa]; bv[k] = b[(y + k) * sb]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_fma(av[k], bv[k], acc);
    }
    if (y + 4 <= len) {
        double av[4], bv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { av[k] = a[(y + k) * sa]; bv[k] = b[(y + k) * sb]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_fma(av[k], bv[k], acc);
        y += 4;
    }
    for (; y < len; ++y) acc = __builtin_fma(a[y * sa], b[y * sb], acc);
    return acc;
}

// ---- several interior steps of a closed loop without a barrier (compile-time nx, nu; no trajectories) ----
// While the unconstrained minimiser stays inside the box the closed loop is x+ = At x + Bt (G0 x + v_r0 + ctr0), G0 = the first nu rows
// of G: nothing of it needs the workgroup.  Every wavefront keeps the state in scalar registers (v_readlane of its lanes 0..nx-1, which
// all compute the plant update; lanes 0..nu-1 of every wave carry the first rows of G besides their own) and tests its own rows of
// v_unc = G x + v_r against the box, up to K steps ahead; one reduction then finds the first step at which any row left the box, f.
// f = K: all K steps are taken.  f < K: wave 0 repeats the first f steps from the chunk's start for the cost and the state (a step
// costs a few hundred ticks against ~1200 for the one-barrier step of the general loop, so the repeat is cheap; the caller grows K
// 1, 4, 16 while chunks succeed).  The state after f steps goes to xs (LDS); dc = the stage costs of those steps in the lanes that
// carry the cost (thread i < nx: x_i (Q x)_i, thread k < nu: u_k (R u)_k, as in the general loop); returned as cost + those.
struct InteriorArgs {
    const ldsd *G, *At, *Bt, *Qm, *Rm;    // G: column-major n x nx with leading dimension np; the rest row-major
    ldsd *xs, *red;
    int np, par;
};
struct InteriorRes { int f, par; double dc; };
template <int NX, int NU>
__device__ __noinline__ InteriorRes interior_steps(InteriorArgs A, int K, bool own, double vr, double h, double ctr, double cost)
{
    static_assert((NU & (NU - 1)) == 0 && (NX & (NX - 1)) == 0 && NU <= 4, "lane & (N - 1) picks the row");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const ldsd *Gp = uni(A.G), *Atp = uni(A.At), *Btp = uni(A.Bt), *Qp = uni(A.Qm), *Rp = uni(A.Rm);
    ldsd *xsp = uni(A.xs), *red = uni(A.red);
    const int np = uni(A.np);
    int par = uni(A.par);
    K = uni(K);
    const int ju = lane & (NU - 1), jx = lane & (NX - 1);
    double g[NX], gu[NX], ar[NX], br[NU], qr[NX], rr[NU];
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        g[a] = own ? Gp[a * np + t] : 0.0;
        gu[a] = Gp[a * np + ju];
        ar[a] = Atp[jx * NX + a];
        qr[a] = Qp[jx * NX + a];
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) { br[k] = Btp[jx * NU + k]; rr[k] = Rp[ju * NU + k]; }
    ldsd *tmp = red + 24;                                        // v_r and the centre of rows 0..nu-1 (wave 0's lanes) for everybody
    if (t < NU) { tmp[t] = vr; tmp[4 + t] = ctr; }
    __syncthreads();
    const double vru = tmp[ju], ctru = tmp[4 + ju];
    double xown = 0.0, dc = 0.0;
    const double xstart = xsp[jx];                               // my component of the state (every 16-lane row holds the state twice)
    // steps from the chunk's start; with test: stop at the first step at which a row of this wavefront leaves the box.
    // The state never leaves the vector registers: component a comes to every dot product through DPP row_newbcast:a.
    auto run = [&](int steps, bool test) -> int {
        double xold = xstart;
        dc = cost;                                               // (same order of additions as the general loop)
        for (int k = 0; k < steps; ++k) {
            double vu = vr, uu = vru, xn = 0.0;
            static_for<0, NX>([&](auto ac) { constexpr int a = decltype(ac)::value; fmac_rowb_3y(vu, uu, xn, xold, g[a], gu[a], ar[a], a); });
            if (test && __ballot(own && !(fabs(vu) <= h)) != 0ull) return k;
            uu += ctru;
            static_for<0, NU>([&](auto jc) { constexpr int j = decltype(jc)::value; fmac_rowb(xn, uu, br[j], j); });
            double qx = 0.0, ru = 0.0;
            static_for<0, NX>([&](auto ac) { constexpr int a = decltype(ac)::value; fmac_rowb(qx, xn, qr[a], a); });
            static_for<0, NU>([&](auto jc) { constexpr int j = decltype(jc)::value; fmac_rowb(ru, uu, rr[j], j); });
            dc = (t < NX) ? __builtin_fma(xn, qx, dc) : dc;
            dc = (t < NU) ? __builtin_fma(uu, ru, dc) : dc;
            xold = xn;
            xown = xn;
        }
        return steps;
    };
    const int fw = run(K, true);
    ldsi *sl = (ldsi *)(red + 16) + 4 * par;
    if (lane == 0) sl[wave] = fw;
    __syncthreads();
    par ^= 1;
    const int f = min(min(sl[0], sl[1]), min(sl[2], sl[3]));
    if (f < K) {
        dc = cost;
        if (wave == 0 && f > 0) run(f, false);                   // uniform per wavefront
    }
    if (f > 0 && t < NX) xsp[t] = xown;
    __syncthreads();
    return InteriorRes{uni(f), par, dc};
}

struct Wg {
    const KParams &p;
    WgOff o;
    ldsd *lds;
    int n, nb, np, nx, nu, N, t;
    double preg[PREG];        // my share of P (flat index t + 256 m over the block image)
    double h, ctr, vr;        // half-width and centre of my row's input box (row < n; 1, 0 otherwise); v_r of my row
    bool own;                 // this thread owns row t (t < n)
    Red R;                    // workgroup-reduction slots and their parity
    // solver state of my row
    double sl, su, zl, zu, rd, v, qs, act_prev;
    int last_m;               // size of the active set the last QP ended with (uniform; 0: it was interior)

    __device__ __forceinline__ ldsi *flag() { return (ldsi *)(lds + o.red + 8); }
    __device__ __forceinline__ const ldsd *shd() const { return lds + o.SH; }      // the batch-shared block, staged in LDS

    // element (i, j) of flat index e in the block image; false for the padding column / rows outside
    __device__ __forceinline__ bool decode(int e, int &i, int &j) const
    {
        const int bidx = e / BLK, w = e - bidx * BLK, r = w / LD, c = w - r * LD;
        int ib = 0;
        while ((ib + 1) * (ib + 2) / 2 <= bidx) ++ib;
        const int jb = bidx - ib * (ib + 1) / 2;
        i = ib * BS + r; j = jb * BS + c;
        return c < BS && ib < nb;
    }
    __device__ __forceinline__ ldsd *kaddr(int i, int j) const      // (i, j) with j's block <= i's block
    {
        return lds + o.K + blk_index(i / BS, j / BS) * BLK + (i % BS) * LD + (j % BS);
    }
    __device__ __forceinline__ double wsym(int i, int j) const { return (i / BS >= j / BS) ? *kaddr(i, j) : *kaddr(j, i); }

    // K <- P (+ dg on my diagonal element when add_diag)
    __device__ __forceinline__ void load_K(bool add_diag, double dg)
    {
        const int cnt = nb * (nb + 1) / 2 * BLK;
#pragma unroll
        for (int m = 0; m < PREG; ++m) {
            const int e = t + THREADS * m;
            if (e < cnt) lds[o.K + e] = preg[m];
        }
        __syncthreads();
        if (add_diag && t < np) *kaddr(t, t) += (own ? dg : 0.0);
        __syncthreads();
    }
    // K rows / columns of the active set -> identity (act flags in LDS)
    __device__ __forceinline__ void mask_K()
    {
        const int cnt = nb * (nb + 1) / 2 * BLK;
        const ldsd *act = lds + o.act;
#pragma unroll
        for (int m = 0; m < PREG; ++m) {
            const int e = t + THREADS * m;
            int i, j;
            if (e < cnt && decode(e, i, j)) {
                if (act[i] != 0.0 || act[j] != 0.0) lds[o.K + e] = (i == j) ? 1.0 : 0.0;
            }
        }
        __syncthreads();
    }
    // y_t = (M w)_t for the symmetric matrix whose image is in K (w: LDS vector of np doubles)
    __device__ __forceinline__ double symv_row(const ldsd *w) const
    {
        double acc = 0.0;
        if (t < np) {
            const int ib = t / BS, r = t % BS;
            for (int jb = 0; jb <= ib; ++jb) {                    // row t, blocks left of and on the diagonal (full block stored)
                const ldsd *B = lds + o.K + blk_index(ib, jb) * BLK + r * LD;
#pragma unroll
                for (int c = 0; c < BS; ++c) acc = __builtin_fma(B[c], w[jb * BS + c], acc);
            }
            for (int kb = ib + 1; kb < nb; ++kb) {                // column t of the blocks below
                const ldsd *B = lds + o.K + blk_index(kb, ib) * BLK + r;
#pragma unroll
                for (int c = 0; c < BS; ++c) acc = __builtin_fma(B[c * LD], w[kb * BS + c], acc);
            }
        }
        return acc;
    }
    __device__ __forceinline__ bool factor() { return chol_blocked(lds + o.K, lds + o.Linv, nb, flag()); }
    __device__ __forceinline__ void solve_vb() { solve_blocked(lds + o.K, lds + o.Linv, nb, lds + o.vb); }

    // K region: P -> W = P^-1.  Cholesky, inverse of the factor, Z'Z; all block products on the MFMA.
    __device__ __forceinline__ bool make_W(bool reload)
    {
        PROF_START;
        if (reload) load_K(false, 0.0);
        bool ok;
        if (o.Xf + (nb - 1) * BLK <= o.total) {                  // (uniform) room for one block per column behind the Fq image
            ok = chol_inverse_blocked(lds + o.K, lds + o.Linv, nb, flag(), lds + o.Xf);
            PROF(8);
        } else {
            ok = factor();
            PROF(8);
            tri_invert_blocked(lds + o.K, lds + o.Linv, nb);
        }
        PROF(9);
        ztz_blocked(lds + o.K, nb);
        PROF(10);
        return ok;
    }

    // rows of the block-row matrix Y (np x 16, block layout) <- W_T * (rows of Ma), W_T an nx x nx weight in LDS
    __device__ __forceinline__ void weight_rows(const ldsd *Ma, const ldsd *Wt, ldsd *Y)
    {
        const int r = t & (THREADS / 2 - 1), half = t / (THREADS / 2);      // two threads per row (np <= 128): even / odd columns
        if (r < np) {
            const ldsd *src = Ma + (r / BS) * BLK + (r % BS) * LD;
            ldsd *dst = Y + (r / BS) * BLK + (r % BS) * LD;
            for (int x = half; x < nx; x += 2) dst[x] = ldot(Wt + x * nx, 1, src, 1, nx);
        }
    }

    // ---------------- condensing (utils_class.py:62-75 in matrix form) ----------------
    // Row i = bi*nu + ui of Gamma'... belongs to stage bi; with a = N-1-bi "stages to go" and M_a = A^a B,
    //   H(i, j) = M_a' P_T M_b + sum_{s>=1} M_{a-s}' Q M_{b-s}
    //           = C1(i, j) + sum_{s>=1} Cq(i + s nu, j + s nu),   C1 = Ma (P_T Ma)', Cq = Ma (Q Ma)',
    // with Ma the n x nx matrix whose row i is column ui of M_a: two block-row products on the MFMA and a
    // suffix sum along the stage diagonals.
    __device__ __forceinline__ void setup_impl(long long b)
    {
        stage(b);
        condense_P();
        finish_condensed();
    }
    // The C5 shape runs the set-up on a copy of this object whose dimensions and LDS offsets are compile-time constants: with one
    // wavefront per SIMD its phases are bound by instruction issue, and every run-time stride costs a multiply (or a divide) per
    // address.  The steps stay on run-time dimensions: with the block counts known the compiler unrolls the row products and gathers
    // of the active-set iterations over all 8 blocks, spills, and the 30 steps get slower (measured: kernel-wide constants 19.4 ms,
    // set-up only 18.x ms per C5 launch).
    __device__ __forceinline__ void setup(long long b)
    {
        if (nx == 8 && nu == 4 && N == 30 && n == 120) {
            Wg c{p, wg_offsets(8, 4, 30), lds, 120, 8, 128, 8, 4, 30, t};
            c.own = own; c.h = h; c.ctr = ctr; c.vr = 0.0; c.R = R;
            c.setup_impl(b);
#pragma unroll
            for (int m = 0; m < PREG; ++m) preg[m] = c.preg[m];
            vr = c.vr; R.par = c.R.par;
        } else {
            setup_impl(b);
        }
    }

    __device__ __forceinline__ void stage(long long b)
    {
        const long long Bsz = p.Bsz;
        ldsd *Ma = lds + o.Linv, *X = lds + o.X;
        ldsd *Am = lds + o.AB, *Bm = Am + nx * nx;
        {   // stage the shared block, the plant and the model
            ldsd *S = lds + o.SH;
            for (int e = t; e < o.shn; e += THREADS) S[e] = p.sh[e];
            const long long rs = (long long)nx * nx + nx * nu + nx;
            for (int e = t; e < nx * nx; e += THREADS) Am[e] = p.rec ? p.rec[b * rs + e] : p.A[(long long)e * Bsz + b];
            for (int e = t; e < nx * nu; e += THREADS) Bm[e] = p.rec ? p.rec[b * rs + nx * nx + e] : p.B[(long long)e * Bsz + b];
            for (int e = t; e < nb * BLK; e += THREADS) { Ma[e] = 0.0; X[e] = 0.0; }
            __syncthreads();
            if (p.true_per_instance) {
                for (int e = t; e < nx * nx; e += THREADS) S[p.so.At + e] = p.At[(long long)e * Bsz + b];
                for (int e = t; e < nx * nu; e += THREADS) S[p.so.Bt + e] = p.Bt[(long long)e * Bsz + b];
            }
        }
        __syncthreads();
    }

    // P = 2 (Gamma'Qbar Gamma + Rbar) into the K region and into the registers
    __device__ __forceinline__ void condense_P()
    {
        PROF_START;
        ldsd *Ma = lds + o.Linv, *X = lds + o.X, *Xf = lds + o.Xf;
        ldsd *Am = lds + o.AB, *Bm = Am + nx * nx;
        const ldsd *sh = shd();
        for (int e = t; e < nb * BLK; e += THREADS) { Ma[e] = 0.0; X[e] = 0.0; }
        __syncthreads();
        // The powers of A by doubling, A^(L+i) = A^L A^i for i = 1..L (log2 N dependent levels instead of N: every level is
        // a barrier and a latency-bound dot product with one wavefront per SIMD), then all M_m = A^m B at once.
        auto powers = [&](auto nxc, auto nuc) {
            const int nx = decltype(nxc)::value ? decltype(nxc)::value : this->nx;
            const int nu = decltype(nuc)::value ? decltype(nuc)::value : this->nu;
            auto ma_row = [&](int m, int u) { const int i = (N - 1 - m) * nu + u; return Ma + (i / BS) * BLK + (i % BS) * LD; };
            const int nn = nx * nx, nb_ = nx * nu;
            for (int e = t; e < nn; e += THREADS) Xf[e] = Am[e];                      // Xf[m] = A^(m+1)
            for (int e = t; e < nb_; e += THREADS) { const int x = e / nu, c = e - x * nu; ma_row(0, c)[x] = Bm[e]; }
            __syncthreads();
            for (int L = 1; L < N; L *= 2) {
                const int cntp = (L < N - L ? L : N - L) * nn;
                for (int e = t; e < cntp; e += THREADS) {
                    const int i = e / nn, r = e - i * nn, x = r / nx, y = r - x * nx;
                    Xf[(L + i) * nn + r] = ldot(Xf + (L - 1) * nn + x * nx, 1, Xf + i * nn + y, nx, nx);
                }
                __syncthreads();
            }
            for (int e = t; e < (N - 1) * nb_; e += THREADS) {
                const int m1 = e / nb_, r = e - m1 * nb_, x = r / nu, c = r - x * nu;
                ma_row(m1 + 1, c)[x] = ldot(Xf + m1 * nn + x * nx, 1, Bm + c, nu, nx);
            }
            __syncthreads();
        };
        if (nx == 8 && nu == 4) powers(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
        else powers(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        PROF(0);
        const int wave = t >> 6;
        // Cq = Ma (Q Ma)' into the lower blocks of K
        weight_rows(Ma, sh + p.so.Q, X);
        __syncthreads();
        {
            int cnt = 0;
            for (int ib = 0; ib < nb; ++ib)
                for (int jb = 0; jb <= ib; ++jb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    d4_t c = {0.0, 0.0, 0.0, 0.0};
                    c = block_xyt(Ma + ib * BLK, X + jb * BLK, c, false, (nx + 3) / 4);      // Ma, X: n x nx, columns nx .. 15 are zero
                    tile_store(lds + o.K + blk_index(ib, jb) * BLK, c);
                }
        }
        __syncthreads();
        PROF(1);
        // suffix sums along the stage diagonals, in place; R on the stage-diagonal blocks
        // (a chain that starts in row i0 has (n - 1 - i0) / nu + 1 elements: odd rounds deal the chains in reverse, so that a thread
        // gets a long and a short one)
        for (int base = 0, rnd = 0; base < n * nu; base += THREADS, ++rnd) {
            const int cnt = (n * nu - base < THREADS) ? n * nu - base : THREADS;
            if (t >= cnt) continue;
            const int ch = (rnd & 1) ? base + cnt - 1 - t : base + t;
            const int i0 = ch / nu, j0 = ch - i0 * nu;
            if (i0 < j0) continue;                                   // upper triangle of a stage block: mirrored below
            const double radd = (i0 < nu) ? sh[p.so.R + i0 * nu + j0] : 0.0;
            double run = 0.0;
            int s = (n - 1 - i0) / nu;
            for (; s >= 7; s -= 8) {                                 // eight loads in flight, then eight stores
                ldsd *e[8];
                double tmp[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { e[k] = kaddr(i0 + (s - k) * nu, j0 + (s - k) * nu); tmp[k] = *e[k]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) { *e[k] = run + radd; run += tmp[k]; }
            }
            if (s >= 3) {
                ldsd *e[4];
                double tmp[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[k] = kaddr(i0 + (s - k) * nu, j0 + (s - k) * nu); tmp[k] = *e[k]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { *e[k] = run + radd; run += tmp[k]; }
                s -= 4;
            }
            for (; s >= 0; --s) {
                ldsd *e = kaddr(i0 + s * nu, j0 + s * nu);
                const double tmp = *e;
                *e = run + radd;
                run += tmp;
            }
        }
        weight_rows(Ma, sh + p.so.P, X);                             // X is not read by the pass above
        __syncthreads();
        {   // K <- 2 (K + Ma (P_T Ma)')
            int cnt = 0;
            for (int ib = 0; ib < nb; ++ib)
                for (int jb = 0; jb <= ib; ++jb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    ldsd *C = lds + o.K + blk_index(ib, jb) * BLK;
                    d4_t c = tile_load(C);
                    c = block_xyt(Ma + ib * BLK, X + jb * BLK, c, false, (nx + 3) / 4);
                    c *= 2.0;
                    tile_store(C, c);
                }
        }
        __syncthreads();
        {   // diagonal blocks: upper triangle from the lower one; unit diagonal on the padding rows
            const int r = t / BS, c = t % BS;
            for (int kb = 0; kb < nb; ++kb) {
                ldsd *D = lds + o.K + blk_index(kb, kb) * BLK;
                if (r < c) D[r * LD + c] = D[c * LD + r];
            }
            __syncthreads();
            if (t >= n && t < np) *kaddr(t, t) = 1.0;
            for (int e = t; e < nb * BLK; e += THREADS) X[e] = 0.0;  // becomes the block-row image of [Fq | qr]
        }
        __syncthreads();
        PROF(2);
        const int total = nb * (nb + 1) / 2 * BLK;
#pragma unroll
        for (int m = 0; m < PREG; ++m) {
            const int e = t + THREADS * m;
            preg[m] = (e < total) ? lds[o.K + e] : 0.0;
        }
        __syncthreads();
    }

    // constant part of the linear term from the references: gq = 2 gref, with
    //   d_r = -xref_r,  lam_r = Q_r d_r + A' lam_{r+1},  gref_r = B' lam_r - R uref_r   (columns r = 0..N-1 <-> x_{r+1}, u_r)
    __device__ __forceinline__ void ref_linear_term(ldsd *gq)
    {
        ldsd *Lam = lds + o.Lam;
        ldsd *Am = lds + o.AB, *Bm = Am + nx * nx;
        const ldsd *sh = shd();
        if (t < np) gq[t] = 0.0;
        if (p.has_ref) {
            const ldsd *xr = sh + p.so.xref, *ur = sh + p.so.uref;
            ldsd *lam = Lam, *lam2 = Lam + nx;
            if (t < nx) lam[t] = 0.0;
            for (int r = N - 1; r >= 0; --r) {
                const ldsd *Qr = sh + ((r < N - 1) ? p.so.Q : p.so.P);
                __syncthreads();
                if (t < nx) {
                    double a = 0.0;
                    for (int y = 0; y < nx; ++y) a = __builtin_fma(Qr[t * nx + y], -xr[y * N + r], a);
                    for (int y = 0; y < nx; ++y) a = __builtin_fma(Am[y * nx + t], lam[y], a);
                    lam2[t] = a;
                }
                __syncthreads();
                if (t < nx) lam[t] = lam2[t];
                if (t < nu) {
                    double a = 0.0;
                    for (int x = 0; x < nx; ++x) a = __builtin_fma(Bm[x * nu + t], lam2[x], a);
                    for (int j = 0; j < nu; ++j) a = __builtin_fma(-sh[p.so.R + t * nu + j], ur[j * N + r], a);
                    gq[r * nu + t] = 2.0 * a;
                }
            }
        }
        __syncthreads();
    }

    // Fq, the constant linear term, W = P^-1 by Cholesky, [G | v_r] = -W [Fq | qr]   (K holds P on entry)
    __device__ __forceinline__ void finish_condensed()
    {
        PROF_START;
        ldsd *X = lds + o.X, *Xf = lds + o.Xf, *Lam = lds + o.Lam;
        ldsd *Am = lds + o.AB, *Bm = Am + nx * nx;
        const ldsd *sh = shd();
        const int wave = t >> 6;
        // Fq = 2 Gamma' Qbar Phi from the costate matrices  Lam_N = P_T A^N,  Lam_k = Q A^k + A' Lam_{k+1}:
        //   Fq block row bi = 2 B' Lam_{bi+1}.
        // Unrolled, Lam_k = sum_{j >= k} (A')^(j-k) W_j A^j (W_N = P_T, W_j = Q): partial sums of length L double by
        //   S_k <- S_k + (A^L)' S_{k+L}
        // -- log2 N levels of independent nx x nx products (the powers of A are in Xf) instead of N dependent stages, each a barrier
        // and two latency-bound dot products with one wavefront per SIMD.  S lives in the Linv region (free between the condensing
        // and the factorisation) when N nx^2 doubles fit there; updated in place, each thread's results held back over a barrier.
        // (compile-time nx, nu for the C5 shape: constant strides turn every dot product's address arithmetic into immediates --
        // with one wavefront per SIMD these phases are bound by instruction issue, not by the LDS)
        auto fq_doubling = [&](auto nxc, auto nuc) {
            const int nx = decltype(nxc)::value ? decltype(nxc)::value : this->nx;
            const int nu = decltype(nuc)::value ? decltype(nuc)::value : this->nu;
            const int nn = nx * nx;
            ldsd *S = lds + o.Linv;
            const int dk = THREADS / nn, dr = THREADS % nn;          // e -> e + THREADS: (k1, r) -> (k1 + dk, r + dr) with carry
            for (int e = t; e < N * nn; e += THREADS) {
                const int k1 = e / nn, r = e - k1 * nn, x = r / nx, y = r - x * nx;
                S[e] = ldot(sh + ((k1 == N - 1) ? p.so.P : p.so.Q) + x * nx, 1, Xf + k1 * nn + y, nx, nx);      // W_k A^k, k = k1 + 1
            }
            __syncthreads();
            for (int L = 1; L < N; L *= 2) {
                const ldsd *AL = Xf + (L - 1) * nn;
                const int total = (N - L) * nn;
                for (int c0 = 0; c0 < total; c0 += 8 * THREADS) {
                    double v[8];
                    int k1 = (c0 + t) / nn, r = (c0 + t) - k1 * nn, x = r / nx, y = r - x * nx;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = c0 + q * THREADS + t;
                        v[q] = (e < total) ? ldot(AL + x, nx, S + (k1 + L) * nn + y, nx, nx, S[e]) : 0.0;
                        k1 += dk; r += dr;
                        if (dr) { if (r >= nn) { r -= nn; ++k1; } x = r / nx; y = r - x * nx; }
                    }
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = c0 + q * THREADS + t;
                        if (e < total) S[e] = v[q];
                    }
                    __syncthreads();
                }
            }
            for (int e = t; e < N * nu * nx; e += THREADS) {
                const int i = e / nx, a = e - i * nx, bi = i / nu, ui = i - bi * nu;
                X[(i / BS) * BLK + (i % BS) * LD + a] = 2.0 * ldot(Bm + ui, nu, S + bi * nn + a, nx, nx);
            }
            __syncthreads();
        };
        if (N * nx * nx <= nb * BLK) {
            if (nx == 8 && nu == 4) fq_doubling(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
            else fq_doubling(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        } else {
            // (the same by the recursion itself; rows of Fq are written one step behind it by a second group of threads)
            const bool mine = t < nx * nx;
            const int x = mine ? t / nx : 0, y = mine ? t % nx : 0;
            const int t2 = t - 64;
            const bool mine2 = t2 >= 0 && t2 < nu * nx;
            const int ui = mine2 ? t2 / nx : 0, a = mine2 ? t2 % nx : 0;
            int cur = 0;
            for (int k = N; k >= 0; --k) {
                if (mine && k >= 1) {
                    const ldsd *Ak = Xf + (k - 1) * nx * nx;       // A^k
                    const ldsd *Wt = sh + ((k == N) ? p.so.P : p.so.Q);
                    double s = ldot(Wt + x * nx, 1, Ak + y, nx, nx);
                    if (k < N) s = ldot(Am + x, nx, Lam + cur * nx * nx + y, nx, nx, s);
                    Lam[(cur ^ 1) * nx * nx + t] = s;
                }
                if (mine2 && k < N) {                                // Lam[cur] = Lam_{k+1}: row block bi = k
                    const double s = ldot(Bm + ui, nu, Lam + cur * nx * nx + a, nx, nx);
                    const int i = k * nu + ui;
                    X[(i / BS) * BLK + (i % BS) * LD + a] = 2.0 * s;
                }
                cur ^= 1;
                __syncthreads();
            }
        }
        PROF(3);
        // constant part of the linear term: qr = 2 gref + P centre
        ldsd *gq = lds + o.vw;
        ref_linear_term(gq);
        double qr = (t < np) ? gq[t] : 0.0;
        if (block_any(own && ctr != 0.0, R)) {
            if (t < np) gq[t] = own ? ctr : 0.0;
            __syncthreads();
            qr += symv_row(gq);                                      // K still holds P here
        }
        const bool have_qr = block_any(own && qr != 0.0, R);
        const bool qr_rides = have_qr && nx < BS;                    // as column nx of the block-row image of Fq
        if (qr_rides && t < np) X[(t / BS) * BLK + (t % BS) * LD + nx] = own ? qr : 0.0;
        PROF(4);
        // W = P^-1 in place of P;  [G | v_r] <- -W [Fq | qr] as block products
        make_W(false);
        PROF(5);
        vr = 0.0;
        {
            ldsd *G = lds + o.G;
            const int lane = t & 63, col = lane & 15, r0 = lane >> 4;
            for (int ib = wave; ib < nb; ib += 4) {
                d4_t c = {0.0, 0.0, 0.0, 0.0};
                c = block_sum<false, true>(0, ib + 1, [&](int jb) { return lds + o.K + blk_index(ib, jb) * BLK; }, [&](int jb) { return X + jb * BLK; }, c, true);
                c = block_sum<true, true>(ib + 1, nb, [&](int jb) { return lds + o.K + blk_index(jb, ib) * BLK; }, [&](int jb) { return X + jb * BLK; }, c, true);
                const double cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ib * BS + r0 + 4 * r;
                    if (col < nx) G[col * np + i] = cv[r];
                    else if (col == nx) lds[o.vb + i] = cv[r];
                }
            }
            __syncthreads();
            if (qr_rides) vr = own ? lds[o.vb + t] : 0.0;
            else if (have_qr) {
                if (t < np) gq[t] = own ? qr : 0.0;
                __syncthreads();
                vr = own ? -symv_row(gq) : 0.0;
            }
            __syncthreads();
        }
        PROF(6);
    }

    // v_unc = G x + v_r for my row (x in LDS at o.xs)
    __device__ __forceinline__ double vunc() const
    {
        return own ? ldot(lds + o.G + t, np, lds + o.xs, 1, nx, vr) : 0.0;
    }

    // rank of my row among the rows with flag a (thread order), and their number m
    __device__ __forceinline__ int rank_active(bool a, int &m)
    {
        const unsigned long long bal = __ballot(a);
        const int lane = t & 63, wave = t >> 6;
        ldsi *wc = (ldsi *)(lds + o.red + 12);
        if (lane == 0) wc[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        m = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wc[w]; off += (w < wave) ? c : 0; m += c; }
        __syncthreads();
        return off + __popcll(bal & ((1ull << lane) - 1ull));
    }

    // ---- active-set iterations from the dual side (see the header): 0 converged, 1 hand over to the fallback ----
    __device__ __forceinline__ int pdas_dual(double vu, double &act, int maxit, int &nfact, double &vout)
    {
        ldsd *S = lds + o.S, *T = lds + o.T, *rb = lds + o.vb, *lamv = lds + o.lam;
        ldsi *list = (ldsi *)(lds + o.list);
        PROF_START;
        for (int it = 0; it < maxit; ++it) {
            int m;
#ifdef LQMPC_WG_PROF
            const long long it_t0 = clock64();
#endif
            const bool a = own && act != 0.0;
            const int rk = rank_active(a, m);
            last_m = m;
            PROF(16);
            if (m > o.smax * BS) return 1;
            double vi = vu, lmine = 0.0;
            if (m > 0) {
                const int nbm = (m + BS - 1) / BS, mp = nbm * BS;
                if (a) { list[rk] = t; rb[rk] = vu - act * h; }
                if (t >= m && t < mp) rb[t] = 0.0;
                __syncthreads();
                const int cnt = nbm * (nbm + 1) / 2 * BLK;
                if (nbm == 1) {                                   // one block: thread t <-> element (t / 16, t % 16), no decoding
                    const int ia = t / BS, ib = t % BS;
                    double val = (ia == ib) ? 1.0 : 0.0;
                    if (ia < m && ib < m) val = wsym(list[ia], list[ib]);
                    S[ia * LD + ib] = val;
                } else
                for (int e = t; e < cnt; e += THREADS) {          // S = W_AA, identity outside
                    const int bidx = e / BLK, w = e - bidx * BLK, r = w / LD, c = w - r * LD;
                    int ab = 0;
                    while ((ab + 1) * (ab + 2) / 2 <= bidx) ++ab;
                    const int bb = bidx - ab * (ab + 1) / 2, ia = ab * BS + r, ib = bb * BS + c;
                    double val = (ia == ib) ? 1.0 : 0.0;
                    if (c < BS && ia < m && ib < m) val = wsym(list[ia], list[ib]);
                    S[e] = val;
                }
                __syncthreads();
                PROF(17);
                const ldsd *sol;
                bool ok = true;
                if (nbm == 1) {
                    if (t < 64) ok = small_spd_solve(S, T, rb, lamv, m);
                    ok = !block_any(!ok, R);
                    sol = lamv;
                } else {
                    ok = chol_blocked(S, lds + o.Linv, nbm, flag());
                    solve_blocked(S, lds + o.Linv, nbm, rb);
                    sol = rb;
                }
                if (!ok) return 1;
                nfact += 1;
                PROF(18);
                if (own) for (int k = 0; k < m; ++k) vi = __builtin_fma(-wsym(t, list[k]), sol[k], vi);
                if (a) { lmine = sol[rk]; vi = act * h; }
                PROF(19);
            }
            const double gtol = 1e-10 * block_max(a ? fabs(lmine) : 0.0, R);
            double na = act;
            if (own) {
                if (!a) na = (vi < -h * (1.0 + 1e-12)) ? -1.0 : ((vi > h * (1.0 + 1e-12)) ? 1.0 : 0.0);
                else na = (act < 0.0) ? ((lmine <= gtol) ? -1.0 : 0.0) : ((lmine >= -gtol) ? 1.0 : 0.0);
            }
            const int chk = block_any3(own && !(fabs(vi) < 1e300), own && na != act, false, R);
            const bool anybad = (chk & 1) != 0, anych = (chk & 2) != 0;
            PROF(20);
#ifdef LQMPC_WG_PROF
            if (threadIdx.x == 0 && blockIdx.x == 0) { g_wg_prof[21] += 1; g_wg_prof[22] += m; }
            if (threadIdx.x == 0) {                                  // all blocks: histogram of m and the ticks spent per bucket
                atomicAdd((unsigned long long *)&g_wg_prof[24 + (m > 63 ? 7 : m / 8)], 1ull);
                atomicAdd((unsigned long long *)&g_wg_prof[32 + (m > 63 ? 7 : m / 8)], (unsigned long long)(clock64() - it_t0));
            }
#endif
            if (anybad) return 1;
            act = na;
            if (!anych) { vout = vi; return 0; }
        }
        return 1;
    }

    // ---- primal-dual active-set iterations on the n x n system, from the face in `act` (fallback path) ----
    __device__ __forceinline__ bool pdas(double &act, double scale, int maxit, int &nfact, double &vout)
    {
        const double gtol = 1e-10 * scale;
        ldsd *vb = lds + o.vb, *vw = lds + o.vw, *actv = lds + o.act;
        for (int k = 0; k < maxit; ++k) {
            if (t < np) { actv[t] = own ? act : 0.0; vw[t] = own ? act * h : 0.0; }
            load_K(false, 0.0);                                   // K = P (barrier inside publishes act / vw)
            const double pd = symv_row(vw);                       // (P v_A) of my row
            __syncthreads();
            if (t < np) vb[t] = own ? ((act != 0.0) ? act * h : -(qs + pd)) : 0.0;
            mask_K();
            const bool ok = factor();
            solve_vb();
            const double vi = (t < np) ? vb[t] : 0.0;
            nfact += 1;
            // gradient P v + q on the active rows
            load_K(false, 0.0);
            const double gi = symv_row(vb) + qs;
            double na = act;
            if (own) {
                if (act == 0.0) na = (vi < -h * (1.0 + 1e-12)) ? -1.0 : ((vi > h * (1.0 + 1e-12)) ? 1.0 : 0.0);
                else na = (act < 0.0) ? ((gi >= -gtol) ? -1.0 : 0.0) : ((gi <= gtol) ? 1.0 : 0.0);
            }
            const bool bad = !ok || (own && !(fabs(vi) < 1e300));
            const bool changed = own && (na != act);
            const int chk = block_any3(bad, changed, false, R);
            const bool anybad = (chk & 1) != 0, anych = (chk & 2) != 0;
            if (anybad) return false;
            act = na;
            if (!anych) { vout = vi; return true; }
        }
        return false;
    }

    // ---- Mehrotra predictor-corrector from the current iterate until gap / residual <= eps_rel (fallback path) ----
    __device__ __forceinline__ int ipm_run(double scale, double hmin, double eps_rel, int &budget, int &iters)
    {
        ldsd *vb = lds + o.vb;
        const double inv2n = 1.0 / (2.0 * n), mu_tol = eps_rel * scale * hmin, rd_tol = eps_rel * scale;
        for (; budget > 0; --budget) {
            const double mu = block_sum(own ? sl * zl + su * zu : 0.0, R) * inv2n;
            const double rn = block_max(own ? fabs(rd) : 0.0, R);
            if (!(mu < 1e300) || !(rn < 1e300)) return 2;
            if (mu <= mu_tol && rn <= rd_tol) return 0;
            iters += 1;
            const double isl = own ? 1.0 / sl : 1.0, isu = own ? 1.0 / su : 1.0;
            load_K(true, zl * isl + zu * isu);
            if (!factor()) return 2;
            if (t < np) vb[t] = own ? (-rd - zl + zu) : 0.0;
            solve_vb();
            const double dva = own ? vb[t] : 0.0;
            double mp = block_max(own ? fmax(-dva * isl, dva * isu) : 0.0, R);
            double md = block_max(own ? fmax(1.0 + dva * isl, 1.0 - dva * isu) : 0.0, R);
            const double apa = mp > 1.0 ? 1.0 / mp : 1.0, ada = md > 1.0 ? 1.0 / md : 1.0;
            const double dzla = -zl * (1.0 + isl * dva), dzua = -zu * (1.0 - isu * dva);
            const double mua = block_sum(own ? (sl + apa * dva) * (zl + ada * dzla) + (su - apa * dva) * (zu + ada * dzua) : 0.0, R) * inv2n;
            double sg = mua / mu; sg = sg * sg * sg;
            const double smu = sg * mu;
            const double rcl = smu - sl * zl - dva * dzla, rcu = smu - su * zu + dva * dzua;
            __syncthreads();
            if (t < np) vb[t] = own ? (-rd + rcl * isl - rcu * isu) : 0.0;
            solve_vb();
            const double dv = own ? vb[t] : 0.0;
            const double dzl = (rcl - zl * dv) * isl, dzu = (rcu + zu * dv) * isu;
            mp = block_max(own ? fmax(fmax(-dv * isl, dv * isu), fmax(-dzl / zl, -dzu / zu)) : 0.0, R);
            const double ap = mp > p.tau ? p.tau / mp : 1.0;     // one step length for primal and dual
            if (own) {
                sl += ap * dv; su -= ap * dv; zl += ap * dzl; zu += ap * dzu;
                rd = (1.0 - ap) * rd;
            }
        }
        return 1;
    }

    // ---- one box QP at the state in LDS (o.xs); result in v (my row) ----
    __device__ __forceinline__ int solve_qp(int &iters)
    {
        const double vu = vunc();
        v = vu;
        const int chk = block_any3(own && !(fabs(vu) <= h), own && !(fabs(vu) < 1e300), own && act_prev != 0.0, R);
        last_m = 0;
        if (!(chk & 1)) { act_prev = 0.0; return 0; }             // presolve: interior minimiser
        if (chk & 2) { v = 0.0; act_prev = 0.0; return 2; }
        // warm start: the previous step's face shifted by one stage, else the rows where v_unc leaves the box
        double act;
        const bool have_prev = (chk & 4) != 0;
        if (have_prev) {
            if (t < np) lds[o.act + t] = own ? act_prev : 0.0;
            __syncthreads();
            act = own ? ((t + nu < n) ? lds[o.act + t + nu] : act_prev) : 0.0;
            __syncthreads();
        } else {
            act = own ? ((vu < -h) ? -1.0 : ((vu > h) ? 1.0 : 0.0)) : 0.0;
        }
        double vsol = 0.0;
        if (p.warm_start && pdas_dual(vu, act, 10, iters, vsol) == 0) {
            v = vsol;
            act_prev = act;
            return 0;
        }
        // fallback: interior point in stages with primal active-set finishing, on n x n matrices in the W region
        if (t < np) lds[o.vw + t] = own ? vu : 0.0;
        load_K(false, 0.0);
        qs = -symv_row(lds + o.vw);                               // q = -P v_unc
        __syncthreads();
        double scale = block_max(own ? fabs(qs) : 0.0, R);
        scale = fmax(scale, 1e-100);
        const double hmin = -block_max(own ? -h : -1e300, R);
        const double z0 = p.z0_scale * scale;
        sl = h; su = h; zl = z0; zu = z0; rd = qs;
        int budget = p.max_iter, status = 1;
        double e_prev = 1e300;
        bool done = false;
        for (int stage = 0; stage < 3 && !done; ++stage) {
            const double e = !p.polish ? p.eps : (stage == 0 ? fmax(1e-6, p.eps) : (stage == 1 ? fmax(1e-9, p.eps) : p.eps));
            if (!(e < e_prev)) continue;
            e_prev = e;
            status = ipm_run(scale, hmin, e, budget, iters);
            if (status == 2) { vsol = 0.0; break; }
            vsol = sl - h;
            if (!p.polish) break;
            act = own ? ((zl > sl) ? -1.0 : ((zu > su) ? 1.0 : 0.0)) : 0.0;
            double vp;
            if (pdas(act, scale, 4, iters, vp)) { vsol = vp; status = 0; done = true; }
            if (budget <= 0) break;
        }
        make_W(true);
        last_m = 1;                                               // (the fallback does not count: assume a constrained step follows)
        v = vsol;
        act_prev = (status == 2) ? 0.0 : act;
        return status;
    }
};

__global__ void __launch_bounds__(256, 1) lqmpc_wg_kernel(KParams p)
{
    extern __shared__ double lds_raw[];
    ldsd *lds = (ldsd *)lds_raw;
    const int t = threadIdx.x;
    const long long slot = blockIdx.x;
    const long long b = p.perm ? (long long)p.perm[slot] : slot;
    const long long Bsz = p.Bsz;
    const int nx = p.nx, nu = p.nu, N = p.N, n = p.n;
    Wg w{p, wg_offsets(nx, nu, N), lds, n, (n + BS - 1) / BS, ((n + BS - 1) / BS) * BS, nx, nu, N, t};
    w.own = t < n;
    w.R = Red{lds + w.o.red, 0};
    w.h = w.own ? 0.5 * (p.sh[p.so.ub + t % nu] - p.sh[p.so.lb + t % nu]) : 1.0;
    w.ctr = w.own ? 0.5 * (p.sh[p.so.ub + t % nu] + p.sh[p.so.lb + t % nu]) : 0.0;
    w.vr = 0.0;
    w.last_m = 0; w.act_prev = 0.0; w.v = 0.0; w.qs = 0.0; w.sl = w.su = w.zl = w.zu = 1.0; w.rd = 0.0;
    w.setup(b);
    const ldsd *sh = w.shd();
    ldsd *xs = lds + w.o.xs;
    auto load_x0 = [&]() {
        if (t < nx) xs[t] = p.rec ? p.rec[b * (nx * nx + nx * nu + nx) + nx * nx + nx * nu + t] : p.x0[(long long)t * Bsz + b];
        __syncthreads();
    };
    // u_k of stage i after a solve: clipped v of row i*nu + k, plus the centre of the box
    auto publish_v = [&]() {
        __syncthreads();
        if (t < w.np) lds[w.o.vw + t] = w.own ? fmin(fmax(w.v, -w.h), w.h) + w.ctr : 0.0;
        __syncthreads();
    };
    int iters = 0, status = 0;
    // V_N by rolling the model forward with the optimal inputs (in LDS at vw): states by the first nx threads,
    // then one stage cost per thread and a workgroup sum.  Every thread gets the value.
    auto value_fn = [&](const ldsd *x0v) -> double {
        ldsd *Xs = lds + w.o.Xs;
        const ldsd *A = lds + w.o.AB, *Bm = A + nx * nx, *uu = lds + w.o.vw;
        if (t < nx) Xs[t] = x0v[t];
        __syncthreads();
        for (int s = 0; s < N; ++s) {
            if (t < nx) {
                Xs[(s + 1) * nx + t] = ldot(Bm + t * nu, 1, uu + s * nu, 1, nu, ldot(A + t * nx, 1, Xs + s * nx, 1, nx));
            }
            __syncthreads();
        }
        double c = 0.0;
        if (t <= N) {
            const int st = t;
            const ldsd *Qs = sh + ((st == N) ? p.so.P : p.so.Q);
            const bool rf = p.has_ref && st >= 1;
            for (int i = 0; i < nx; ++i) {
                const double di = Xs[st * nx + i] - (rf ? sh[p.so.xref + i * N + st - 1] : 0.0);
                double r = 0.0;
                for (int j = 0; j < nx; ++j) r = __builtin_fma(Qs[i * nx + j], Xs[st * nx + j] - (rf ? sh[p.so.xref + j * N + st - 1] : 0.0), r);
                c = __builtin_fma(di, r, c);
            }
            if (st < N)
                for (int k = 0; k < nu; ++k) {
                    const double dk = uu[st * nu + k] - (p.has_ref ? sh[p.so.uref + k * N + st] : 0.0);
                    double r = 0.0;
                    for (int j = 0; j < nu; ++j) r = __builtin_fma(sh[p.so.R + k * nu + j], uu[st * nu + j] - (p.has_ref ? sh[p.so.uref + j * N + st] : 0.0), r);
                    c = __builtin_fma(dk, r, c);
                }
        }
        return block_sum(c, w.R);
    };
    PROF_START;
    // one loop for the three entry points: a single QP (solve), K start states (max V_N), T closed-loop steps
    // (utils_class.py:266-283; thread i < nx carries x_i (Q x)_i of J_T, thread k < nu carries u_k (R u)_k)
    const int mode = p.mode;
    const int nsteps = (mode == MODE_SOLVE) ? 1 : ((mode == MODE_MAXVN) ? p.K : p.T);
    const ldsd *uu = lds + w.o.vw;
    double cost = 0.0, best = -1e308;
    auto stage_cost = [&](bool with_u) {
        if (t < nx) {
            cost = __builtin_fma(xs[t], ldot(sh + p.so.Q + t * nx, 1, xs, 1, nx), cost);
        }
        if (with_u && t < nu) {
            cost = __builtin_fma(uu[t], ldot(sh + p.so.R + t * nu, 1, uu, 1, nu), cost);
        }
    };
    if (mode != MODE_MAXVN) load_x0();
    if (mode == MODE_ROLLOUT) {
        stage_cost(false);
        if (p.X && t < nx) p.X[((long long)t * (p.T + 1)) * Bsz + b] = xs[t];
    }
    const bool chunked = mode == MODE_ROLLOUT && nx == 8 && nu == 4 && !p.X && !p.U;
    int kchunk = 1;
    for (int step = 0; step < nsteps; ++step) {
        if (mode == MODE_ROLLOUT) {
            // Interior steps, one barrier each (most steps of most rollouts: 27 % of the C5 launch went into them when each took
            // the seven barriers of solve_qp + publish_v + the plant update).  The next state is computed SPECULATIVELY from the
            // unclipped minimiser into the other half of the state buffer while the box test is still in flight: the test's
            // workgroup-wide OR is the one barrier, and it also publishes the state.  A step that fails the test (or holds a
            // NaN) leaves everything as it was and takes the general path below.
            if (chunked && w.last_m > 0) {
                // the last step ended on a non-empty active set: its shift is the next step's guess and an interior step is unlikely --
                // no speculative attempt (the general path's presolve still catches an interior minimiser)
            } else if (chunked) {
                // (no trajectories, C5's dimensions: barrier-free chunks of interior steps, see interior_steps)
                const InteriorArgs ia{lds + w.o.G, sh + p.so.At, sh + p.so.Bt, sh + p.so.Q, sh + p.so.R, xs, lds + w.o.red, w.np, 0};
                while (step < nsteps) {
                    const int kreq = kchunk < nsteps - step ? kchunk : nsteps - step;
                    InteriorArgs a2 = ia;
                    a2.par = w.R.par;
                    const InteriorRes r = interior_steps<8, 4>(a2, kreq, w.own, w.vr, w.h, w.ctr, cost);
                    w.R.par = r.par;
                    cost = r.dc;
                    step += r.f;
                    if (r.f > 0) w.act_prev = 0.0;
                    if (r.f < kreq) { kchunk = 1; break; }            // step `step` leaves the box: the general path below
                    kchunk = kchunk < 16 ? 4 * kchunk : 16;
                }
                if (step >= nsteps) break;
            } else {
            ldsd *uw = lds + w.o.vw;
            ldsd *cur = xs, *nxt = xs + nx;                           // the state ping-pongs between the two halves: one barrier per step
            // (NXI, NUI) > 0: the dimensions at compile time (C5's) -- my row of G and of the plant stay in registers over the
            // steps and only the state and the inputs go through LDS; (0, 0): run-time dimensions, everything read from LDS per step
            auto interior = [&](auto nxc, auto nuc) {
                constexpr int NXI = decltype(nxc)::value, NUI = decltype(nuc)::value;
                constexpr bool FIX = NXI > 0;
                double g[FIX ? NXI : 1], ar[FIX ? NXI : 1], br[FIX ? NUI : 1];    // (the weights' rows stay in LDS: they are off the chain)
                if constexpr (FIX) {
                    const int tx = t < NXI ? t : 0;
#pragma unroll
                    for (int a = 0; a < NXI; ++a) {
                        g[a] = w.own ? lds[w.o.G + a * w.np + t] : 0.0;
                        ar[a] = sh[p.so.At + tx * NXI + a];
                    }
#pragma unroll
                    for (int k = 0; k < NUI; ++k) br[k] = sh[p.so.Bt + tx * NUI + k];
                }
                for (;;) {
                    double vu, xn = 0.0;
                    if constexpr (FIX) {
                        double xv[NXI];
#pragma unroll
                        for (int a = 0; a < NXI; ++a) xv[a] = cur[a];
                        vu = w.vr;
#pragma unroll
                        for (int a = 0; a < NXI; ++a) vu = __builtin_fma(g[a], xv[a], vu);
                        vu = w.own ? vu : 0.0;
                        if (t < NUI) uw[t] = vu + w.ctr;              // same wavefront as the readers below: LDS keeps the order
                        if (t < NXI) {
#pragma unroll
                            for (int a = 0; a < NXI; ++a) xn = __builtin_fma(ar[a], xv[a], xn);
#pragma unroll
                            for (int k = 0; k < NUI; ++k) xn = __builtin_fma(br[k], uw[k], xn);
                            nxt[t] = xn;
                        }
                    } else {
                        vu = w.own ? ldot(lds + w.o.G + t, w.np, cur, 1, nx, w.vr) : 0.0;
                        if (t < nu) uw[t] = vu + w.ctr;
                        if (t < nx) {
                            xn = ldot(sh + p.so.Bt + t * nu, 1, uw, 1, nu, ldot(sh + p.so.At + t * nx, 1, cur, 1, nx));
                            nxt[t] = xn;
                        }
                    }
                    const bool out = w.own && !(fabs(vu) <= w.h);
                    if (block_any(out, w.R)) break;
                    // inside the box: commit the step (costs, trajectories); the next round writes the half nobody reads any more
                    if (t < nx) {
                        cost = __builtin_fma(xn, ldot(sh + p.so.Q + t * nx, 1, nxt, 1, nx), cost);
                        if (p.X) p.X[((long long)t * (p.T + 1) + step + 1) * Bsz + b] = xn;
                    }
                    if (t < nu) {
                        const double ut = uw[t];
                        cost = __builtin_fma(ut, ldot(sh + p.so.R + t * nu, 1, uw, 1, nu), cost);
                        if (p.U) p.U[((long long)t * p.T + step) * Bsz + b] = ut;
                    }
                    ldsd *sw = cur; cur = nxt; nxt = sw;
                    w.act_prev = 0.0;
                    if (++step >= nsteps) break;
                }
            };
            if (nx == 8 && nu == 4) interior(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
            else interior(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if (cur != xs) {                                          // uniform: the general path expects the state in the first half
                __syncthreads();
                if (t < nx) xs[t] = cur[t];
                __syncthreads();
            }
            if (step >= nsteps) break;
            }
        }
        if (mode == MODE_MAXVN) {
            __syncthreads();
            if (t < nx) xs[t] = p.sh[p.so.x0s + t * p.K + step];
            __syncthreads();
            w.act_prev = 0.0;
        }
        const int st = w.solve_qp(iters);
        status = st > status ? st : status;
        if (mode == MODE_ROLLOUT) {
            // u_0, the plant and the stage cost live in wave 0 (nx <= 16, nu <= 64): the LDS executes one wavefront's accesses in
            // order, so between the barrier that retires the readers of the old state and the one that publishes the new state
            // nothing else is needed (publish_v + the update cost four barriers)
            __syncthreads();
            ldsd *uw = lds + w.o.vw;
            if (t < nu) uw[t] = fmin(fmax(w.v, -w.h), w.h) + w.ctr;
            double xn = 0.0;
            if (t < nx) xn = ldot(sh + p.so.Bt + t * nu, 1, uw, 1, nu, ldot(sh + p.so.At + t * nx, 1, xs, 1, nx));
            if (t < nx) xs[t] = xn;
            stage_cost(true);
            if (p.X && t < nx) p.X[((long long)t * (p.T + 1) + step + 1) * Bsz + b] = xn;
            if (p.U && t < nu) p.U[((long long)t * p.T + step) * Bsz + b] = uu[t];
            __syncthreads();
        } else {
            publish_v();
            const double vn = value_fn(xs);
            best = (vn > best || vn != vn) ? vn : best;
            if (mode == MODE_SOLVE) {
                if (t == 0) p.VN[b] = vn;
                if (t < nu) p.u0[(long long)t * Bsz + b] = uu[t];
            }
        }
    }
    if (mode == MODE_ROLLOUT) {
        cost = block_sum(cost, w.R);
        if (t == 0) p.JT[b] = cost;
    } else if (mode == MODE_MAXVN) {
        if (t == 0) p.MV[b] = best;
    }
    PROF(13);
    if (t == 0) {
        if (p.status) p.status[b] = status;
        if (p.iters) p.iters[b] = iters;
    }
}

bool launch_wg(const KParams &p, hipStream_t stream, const char **name)
{
    const size_t bytes = wg_lds_bytes(p.nx, p.nu, p.N);
    void (*kern)(KParams) = lqmpc_wg_kernel;
    // dynamic-LDS opt-in (per device and per function: set on every launch, it is a cheap host-side call)
    const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "lqmpc: hipFuncSetAttribute(%zu bytes of LDS): %s\n", bytes, hipGetErrorString(e));
        return false;
    }
#ifdef LQMPC_WG_PROF
    long long z[48] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_prof), z, sizeof z);
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)p.Bsz), dim3(256), bytes, stream, p);
#ifdef LQMPC_WG_PROF
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_wg_prof), sizeof z);
    fprintf(stderr, "wg prof (block 0 ticks): chains %lld Cq %lld H %lld Fq %lld qr %lld makeW %lld [chol %lld triinv %lld ztz %lld] G/vr %lld | solve_qp %lld value_fn %lld rollout %lld\n",
            z[0], z[1], z[2], z[3], z[4], z[5], z[8], z[9], z[10], z[6], z[11], z[12], z[13]);
    fprintf(stderr, "wg prof dual iterations, all blocks, by m in [0,8) [8,16) ...: %lld %lld %lld %lld %lld %lld %lld %lld\n", z[24], z[25], z[26], z[27], z[28], z[29], z[30], z[31]);
    fprintf(stderr, "wg prof   mean ticks per iteration by the same buckets: %lld %lld %lld %lld %lld %lld %lld %lld\n", z[32] / (z[24] ? z[24] : 1), z[33] / (z[25] ? z[25] : 1),
            z[34] / (z[26] ? z[26] : 1), z[35] / (z[27] ? z[27] : 1), z[36] / (z[28] ? z[28] : 1), z[37] / (z[29] ? z[29] : 1), z[38] / (z[30] ? z[30] : 1), z[39] / (z[31] ? z[31] : 1));
    fprintf(stderr, "wg prof dual iterations (block 0): %lld iterations, mean m %.1f; ticks per iteration: rank %lld gather %lld solve %lld update %lld checks %lld\n",
            z[21], z[21] ? (double)z[22] / z[21] : 0.0, z[16] / (z[21] ? z[21] : 1), z[17] / (z[21] ? z[21] : 1), z[18] / (z[21] ? z[21] : 1), z[19] / (z[21] ? z[21] : 1), z[20] / (z[21] ? z[21] : 1));
#endif
    if (name) *name = "lqmpc_wg_kernel";
    return true;
}

}  // namespace lqmpc
