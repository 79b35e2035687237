// lqmpc_wg.hip -- large-n kernel: one 256-thread workgroup per (A,B) instance, 32 < n <= 128 (C5: n = 120).
//
// Where things live (gfx950, 160 KiB LDS and 512 KiB of registers per CU, one workgroup per CU):
//   registers  the constant matrix P = 2(Gamma'Qbar Gamma + Rbar): the lower block triangle of 16x16
//              blocks, 39 doubles per thread (flat over the workgroup, same order as the LDS image);
//              one element of every solver vector per thread (thread i <-> row i);
//   LDS        the working matrix K -> L (same block layout), the inverses of L's diagonal blocks,
//              G = -P^-1 Fq (n x nx), two work vectors, the active-set flags, A^m B while condensing.
//   HBM        inputs and results only.
// Linear algebra: lqmpc_wg_linalg.h -- blocked Cholesky whose panel and trailing updates are 16x16x16
// products on v_mfma_f64_16x16x4_f64 (the one place of this path that is a dense contraction), blocked
// substitution with the inverted diagonal blocks.
// Algorithm per QP (same as the other kernels): presolve (unconstrained minimiser
// G x inside the box -> done), primal-dual active-set iterations warm-started from the previous step's
// face, Mehrotra interior point in stages with active-set finishing as the fallback.
//
// References and asymmetric boxes enter as in the other kernels: u = v + centre, |v| <= h, and the constant
// part of the linear term qr = 2 gref + P centre (gref by a costate recursion on vectors) shifts the
// unconstrained minimiser by v_r = -P^-1 qr.
#include "lqmpc_wg_linalg.h"
#include <cstdio>

namespace lqmpc {

using namespace wg;

constexpr int PREG = 39;                 // ceil(36 blocks * 272 doubles / 256 threads)

struct WgOff {                            // LDS offsets in doubles
    int K, Linv, G, vb, vw, act, red, xs, M, PM, DM, Xf, Lam, total;
};

__host__ __device__ inline WgOff wg_offsets(int nx, int nu, int N)
{
    const int n = N * nu, nb = (n + BS - 1) / BS, np = nb * BS;
    WgOff o;
    int c = 0;
    o.K = c;    c += nb * (nb + 1) / 2 * BLK;
    o.Linv = c; c += nb * BLK;
    o.G = c;    c += np * nx;
    o.vb = c;   c += np;
    o.vw = c;   c += np;
    o.act = c;  c += np;
    o.red = c;  c += 16;
    o.xs = c;   c += 2 * nx + 8;
    o.M = c;    c += N * nx * nu;
    o.PM = c;   c += N * nx * nu;
    o.DM = c;   c += N * nx * nu;
    o.Xf = c;   c += (N + 1) * nx * nx;
    o.Lam = c;  c += 2 * nx * nx;
    o.total = c;
    return o;
}

size_t wg_lds_bytes(int nx, int nu, int N) { return (size_t)wg_offsets(nx, nu, N).total * sizeof(double) + 64; }

bool wg_supported(const KParams &p, const double *lb, const double *ub)
{
    (void)lb; (void)ub;
    if (p.n <= 32 || p.n > 128 || p.nx > 16) return false;
    return wg_lds_bytes(p.nx, p.nu, p.N) <= 160 * 1024;
}

// ---- workgroup reductions over the threads that own a row (others pass the neutral element) ----
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ double block_sum(double x, double *red)
{
    x = wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_max(double x, double *red)
{
    x = wave_max(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__device__ __forceinline__ bool block_any(bool f) { return __syncthreads_or(f ? 1 : 0) != 0; }

struct Wg {
    const KParams &p;
    WgOff o;
    double *lds;
    int n, nb, np, nx, nu, N, t;
    double preg[PREG];        // my share of P (flat index t + 256 m over the block image)
    double h, ctr, vr;        // half-width and centre of my row's input box (row < n; 1, 0 otherwise); v_r of my row
    bool own;                 // this thread owns row t (t < n)
    // solver state of my row
    double sl, su, zl, zu, rd, v, qs, act_prev;

    __device__ __forceinline__ int *flag() { return (int *)(lds + o.red + 8); }

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
    __device__ __forceinline__ double *kaddr(int i, int j) const      // (i, j) with j's block <= i's block
    {
        return lds + o.K + blk_index(i / BS, j / BS) * BLK + (i % BS) * LD + (j % BS);
    }

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
        const double *act = lds + o.act;
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
    // y_t = (P w)_t from the unfactored image of P in K (w: LDS vector of np doubles)
    __device__ __forceinline__ double symv_row(const double *w) const
    {
        double acc = 0.0;
        if (t < np) {
            const int ib = t / BS, r = t % BS;
            for (int jb = 0; jb <= ib; ++jb) {                    // row t, blocks left of and on the diagonal (full block stored)
                const double *B = lds + o.K + blk_index(ib, jb) * BLK + r * LD;
#pragma unroll
                for (int c = 0; c < BS; ++c) acc = __builtin_fma(B[c], w[jb * BS + c], acc);
            }
            for (int kb = ib + 1; kb < nb; ++kb) {                // column t of the blocks below
                const double *B = lds + o.K + blk_index(kb, ib) * BLK + r;
#pragma unroll
                for (int c = 0; c < BS; ++c) acc = __builtin_fma(B[c * LD], w[kb * BS + c], acc);
            }
        }
        return acc;
    }
    __device__ __forceinline__ bool factor() { return chol_blocked(lds + o.K, lds + o.Linv, nb, flag()); }
    __device__ __forceinline__ void solve_vb() { solve_blocked(lds + o.K, lds + o.Linv, nb, lds + o.vb); }

    // ---------------- condensing (utils_class.py:62-75 in matrix form) ----------------
    __device__ void setup(long long b)
    {
        const long long Bsz = p.Bsz;
        const double *sh = p.sh;
        double *M = lds + o.M, *PM = lds + o.PM, *DM = lds + o.DM, *Xf = lds + o.Xf, *Lam = lds + o.Lam;
        const double *A = p.rec ? nullptr : p.A;
        auto ldA = [&](int e) { return p.rec ? p.rec[b * (nx * nx + nx * nu + nx) + e] : A[(long long)e * Bsz + b]; };
        auto ldB = [&](int e) { return p.rec ? p.rec[b * (nx * nx + nx * nu + nx) + nx * nx + e] : p.B[(long long)e * Bsz + b]; };
        // A into Xf[0] (scratch for A itself: Xf block 0 holds A, blocks k >= 1 hold A^k), B into M[0]
        for (int e = t; e < nx * nx; e += THREADS) Xf[e] = ldA(e);
        for (int e = t; e < nx * nu; e += THREADS) M[e] = ldB(e);
        __syncthreads();
        // M[m] = A M[m-1];  Xf[k] = A Xf[k-1]  (A^1 = A is Xf[0] shifted: keep A in Xf[0], A^k in Xf[k-1])
        for (int m = 1; m < N; ++m) {
            for (int e = t; e < nx * nu; e += THREADS) {
                const int x = e / nu, u = e % nu;
                double s = 0.0;
                for (int y = 0; y < nx; ++y) s = __builtin_fma(Xf[x * nx + y], M[((m - 1) * nx + y) * nu + u], s);
                M[(m * nx + x) * nu + u] = s;
            }
            __syncthreads();
        }
        for (int k = 1; k < N; ++k) {                             // Xf[k] = A^(k+1)
            for (int e = t; e < nx * nx; e += THREADS) {
                const int x = e / nx, y = e % nx;
                double s = 0.0;
                for (int z = 0; z < nx; ++z) s = __builtin_fma(Xf[x * nx + z], Xf[((k - 1) * nx + z) * nx + y], s);
                Xf[(k * nx + x) * nx + y] = s;
            }
            __syncthreads();
        }
        // PM[m] = P_T M[m],  DM[m] = (Q - P_T) M[m]
        for (int e = t; e < N * nx * nu; e += THREADS) {
            const int m = e / (nx * nu), x = (e / nu) % nx, u = e % nu;
            double s1 = 0.0, s2 = 0.0;
            for (int y = 0; y < nx; ++y) {
                const double mv = M[(m * nx + y) * nu + u];
                s1 = __builtin_fma(sh[p.so.P + x * nx + y], mv, s1);
                s2 = __builtin_fma(sh[p.so.Q + x * nx + y] - sh[p.so.P + x * nx + y], mv, s2);
            }
            PM[e] = s1; DM[e] = s2;
        }
        // zero the image of K, unit diagonal on the padding rows
        const int cnt = nb * (nb + 1) / 2 * BLK;
        for (int e = t; e < cnt; e += THREADS) lds[o.K + e] = 0.0;
        __syncthreads();
        if (t >= n && t < np) *kaddr(t, t) = 1.0;
        // H by diagonals: with a, b = stages to go of the row / column block, b = a + d,
        //   S(a, b) = S(a-1, b-1) + M_a' P_T M_b + M_{a-1}' (Q - P_T) M_{b-1},   S(0, b) = M_0' P_T M_b,
        // block row bi = N-1-a, block column bj = N-1-b <= bi.  One chain per (d, ui, uj).
        for (int ch = t; ch < N * nu * nu; ch += THREADS) {
            const int d = ch / (nu * nu), ui = (ch / nu) % nu, uj = ch % nu;
            double S = 0.0;
            for (int a = 0; a + d < N; ++a) {
                const int bq = a + d;
                double s = S;
                for (int x = 0; x < nx; ++x) s = __builtin_fma(M[(a * nx + x) * nu + ui], PM[(bq * nx + x) * nu + uj], s);
                if (a > 0)
                    for (int x = 0; x < nx; ++x) s = __builtin_fma(M[((a - 1) * nx + x) * nu + ui], DM[((bq - 1) * nx + x) * nu + uj], s);
                S = s;
                const int i = (N - 1 - a) * nu + ui, j = (N - 1 - bq) * nu + uj;
                double val = S + ((d == 0) ? sh[p.so.R + ui * nu + uj] : 0.0);
                val *= 2.0;
                if (i / BS > j / BS || (i / BS == j / BS)) {
                    if (j <= i || i / BS == j / BS) {
                        if (j / BS <= i / BS) *kaddr(i, j) = val;
                    }
                }
                if (i != j && i / BS == j / BS) *kaddr(j, i) = val;           // both triangles inside a diagonal block
            }
        }
        __syncthreads();
        const int total = nb * (nb + 1) / 2 * BLK;
#pragma unroll
        for (int m = 0; m < PREG; ++m) {
            const int e = t + THREADS * m;
            preg[m] = (e < total) ? lds[o.K + e] : 0.0;
        }
        // Fq = 2 Gamma' Qbar Phi by the costate recursion on matrices:
        //   Lam_N = P_T A^N,  Lam_k = Q A^k + A' Lam_{k+1},  Fq block row bi = 2 B' Lam_{bi+1}
        double *G = lds + o.G;
        for (int e = t; e < np * nx; e += THREADS) G[e] = 0.0;
        int cur = 0;
        for (int k = N; k >= 1; --k) {
            const double *Ak = Xf + (k - 1) * nx * nx;            // A^k
            const double *W = sh + ((k == N) ? p.so.P : p.so.Q);
            __syncthreads();
            for (int e = t; e < nx * nx; e += THREADS) {
                const int x = e / nx, y = e % nx;
                double s = 0.0;
                for (int z = 0; z < nx; ++z) s = __builtin_fma(W[x * nx + z], Ak[z * nx + y], s);
                if (k < N)
                    for (int z = 0; z < nx; ++z) s = __builtin_fma(Xf[z * nx + x], Lam[cur * nx * nx + z * nx + y], s);   // A'[x][z] = A[z][x]
                Lam[(cur ^ 1) * nx * nx + e] = s;
            }
            cur ^= 1;
            __syncthreads();
            const int bi = k - 1;
            for (int e = t; e < nu * nx; e += THREADS) {
                const int ui = e / nx, a = e % nx;
                double s = 0.0;
                for (int x = 0; x < nx; ++x) s = __builtin_fma(M[x * nu + ui], Lam[cur * nx * nx + x * nx + a], s);        // B = M[0]
                G[(bi * nu + ui) * nx + a] = 2.0 * s;
            }
        }
        __syncthreads();
        // presolve data: G <- -P^-1 Fq, one column at a time through the factor of P
        load_K(false, 0.0);
        factor();
        for (int a = 0; a < nx; ++a) {
            if (t < np) lds[o.vb + t] = own ? G[t * nx + a] : 0.0;
            __syncthreads();
            solve_vb();
            if (own) G[t * nx + a] = -lds[o.vb + t];
            __syncthreads();
        }
        // constant part of the linear term: qr = 2 gref + P centre, with
        //   d_r = -xref_r,  lam_r = Q_r d_r + A' lam_{r+1},  gref_r = B' lam_r - R uref_r   (columns r = 0..N-1 <-> x_{r+1}, u_r)
        double *gq = lds + o.vw;
        if (t < np) gq[t] = 0.0;
        if (p.has_ref) {
            const double *xr = sh + p.so.xref, *ur = sh + p.so.uref;
            double *lam = Lam, *lam2 = Lam + nx;
            if (t < nx) lam[t] = 0.0;
            for (int r = N - 1; r >= 0; --r) {
                const double *Qr = sh + ((r < N - 1) ? p.so.Q : p.so.P);
                __syncthreads();
                if (t < nx) {
                    double a = 0.0;
                    for (int y = 0; y < nx; ++y) a = __builtin_fma(Qr[t * nx + y], -xr[y * N + r], a);
                    for (int y = 0; y < nx; ++y) a = __builtin_fma(Xf[y * nx + t], lam[y], a);
                    lam2[t] = a;
                }
                __syncthreads();
                if (t < nx) lam[t] = lam2[t];
                if (t < nu) {
                    double a = 0.0;
                    for (int x = 0; x < nx; ++x) a = __builtin_fma(M[x * nu + t], lam2[x], a);
                    for (int j = 0; j < nu; ++j) a = __builtin_fma(-sh[p.so.R + t * nu + j], ur[j * N + r], a);
                    gq[r * nu + t] = 2.0 * a;
                }
            }
            __syncthreads();
        }
        const double gref2 = (t < np) ? gq[t] : 0.0;
        __syncthreads();
        if (t < np) gq[t] = own ? ctr : 0.0;
        load_K(false, 0.0);
        const double qr = gref2 + symv_row(gq);
        vr = 0.0;
        if (block_any(own && qr != 0.0)) {
            __syncthreads();
            if (t < np) lds[o.vb + t] = own ? qr : 0.0;
            factor();
            solve_vb();
            vr = own ? -lds[o.vb + t] : 0.0;
            __syncthreads();
        }
    }

    // v_unc = G x for my row (x in LDS at o.xs)
    __device__ __forceinline__ double vunc() const
    {
        double s = own ? vr : 0.0;
        if (own) for (int a = 0; a < nx; ++a) s = __builtin_fma(lds[o.G + t * nx + a], lds[o.xs + a], s);
        return s;
    }

    // ---- primal-dual active-set iterations from the face in `act` (my row: -1 / 0 / +1) ----
    __device__ bool pdas(double &act, double scale, int maxit, int &nfact, double &vout)
    {
        const double gtol = 1e-10 * scale;
        double *vb = lds + o.vb, *vw = lds + o.vw, *actv = lds + o.act;
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
            const bool anybad = block_any(bad);
            const bool anych = block_any(changed);
            if (anybad) return false;
            act = na;
            if (!anych) { vout = vi; return true; }
        }
        return false;
    }

    // ---- Mehrotra predictor-corrector from the current iterate until gap / residual <= eps_rel ----
    __device__ int ipm_run(double scale, double hmin, double eps_rel, int &budget, int &iters)
    {
        double *vb = lds + o.vb, *red = lds + o.red;
        const double inv2n = 1.0 / (2.0 * n), mu_tol = eps_rel * scale * hmin, rd_tol = eps_rel * scale;
        for (; budget > 0; --budget) {
            const double mu = block_sum(own ? sl * zl + su * zu : 0.0, red) * inv2n;
            const double rn = block_max(own ? fabs(rd) : 0.0, red);
            if (!(mu < 1e300) || !(rn < 1e300)) return 2;
            if (mu <= mu_tol && rn <= rd_tol) return 0;
            iters += 1;
            const double isl = own ? 1.0 / sl : 1.0, isu = own ? 1.0 / su : 1.0;
            load_K(true, zl * isl + zu * isu);
            if (!factor()) return 2;
            if (t < np) vb[t] = own ? (-rd - zl + zu) : 0.0;
            solve_vb();
            const double dva = own ? vb[t] : 0.0;
            double mp = block_max(own ? fmax(-dva * isl, dva * isu) : 0.0, red);
            double md = block_max(own ? fmax(1.0 + dva * isl, 1.0 - dva * isu) : 0.0, red);
            const double apa = mp > 1.0 ? 1.0 / mp : 1.0, ada = md > 1.0 ? 1.0 / md : 1.0;
            const double dzla = -zl * (1.0 + isl * dva), dzua = -zu * (1.0 - isu * dva);
            const double mua = block_sum(own ? (sl + apa * dva) * (zl + ada * dzla) + (su - apa * dva) * (zu + ada * dzua) : 0.0, red) * inv2n;
            double sg = mua / mu; sg = sg * sg * sg;
            const double smu = sg * mu;
            const double rcl = smu - sl * zl - dva * dzla, rcu = smu - su * zu + dva * dzua;
            __syncthreads();
            if (t < np) vb[t] = own ? (-rd + rcl * isl - rcu * isu) : 0.0;
            solve_vb();
            const double dv = own ? vb[t] : 0.0;
            const double dzl = (rcl - zl * dv) * isl, dzu = (rcu + zu * dv) * isu;
            mp = block_max(own ? fmax(fmax(-dv * isl, dv * isu), fmax(-dzl / zl, -dzu / zu)) : 0.0, red);
            const double ap = mp > p.tau ? p.tau / mp : 1.0;     // one step length for primal and dual
            if (own) {
                sl += ap * dv; su -= ap * dv; zl += ap * dzl; zu += ap * dzu;
                rd = (1.0 - ap) * rd;
            }
        }
        return 1;
    }

    // ---- one box QP at the state in LDS (o.xs); result in v (my row) ----
    __device__ int solve_qp(int &iters)
    {
        double *red = lds + o.red;
        const double vu = vunc();
        v = vu;
        if (!block_any(own && !(fabs(vu) <= h))) { act_prev = 0.0; return 0; }     // presolve: interior minimiser
        // q = -P v_unc
        if (t < np) lds[o.vw + t] = own ? vu : 0.0;
        load_K(false, 0.0);
        qs = -symv_row(lds + o.vw);
        __syncthreads();
        double scale = block_max(own ? fabs(qs) : 0.0, red);
        if (!(scale < 1e300)) { v = 0.0; act_prev = 0.0; return 2; }
        scale = fmax(scale, 1e-100);
        // warm start: the previous step's face shifted by one stage, else the rows where v_unc leaves the box
        double act;
        const bool have_prev = block_any(own && act_prev != 0.0);
        if (have_prev) {
            if (t < np) lds[o.act + t] = own ? act_prev : 0.0;
            __syncthreads();
            act = own ? ((t + nu < n) ? lds[o.act + t + nu] : act_prev) : 0.0;
            __syncthreads();
        } else {
            act = own ? ((vu < -h) ? -1.0 : ((vu > h) ? 1.0 : 0.0)) : 0.0;
        }
        int status = 0;
        double vsol = 0.0;
        bool done = false;
        if (p.warm_start) done = pdas(act, scale, 8, iters, vsol);
        if (!done) {
            // fallback: interior point in stages, active-set finishing after each stage
            const double hmin = -block_max(own ? -h : -1e300, red);
            const double z0 = p.z0_scale * scale;
            sl = h; su = h; zl = z0; zu = z0; rd = qs;
            int budget = p.max_iter;
            double e_prev = 1e300;
            status = 1;
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
        }
        v = vsol;
        act_prev = (status == 2) ? 0.0 : act;
        return status;
    }
};

__global__ void __launch_bounds__(256, 1) lqmpc_wg_kernel(KParams p)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const long long slot = blockIdx.x;
    const long long b = p.perm ? (long long)p.perm[slot] : slot;
    const long long Bsz = p.Bsz;
    const int nx = p.nx, nu = p.nu, N = p.N, n = p.n;
    Wg w{p, wg_offsets(nx, nu, N), lds, n, (n + BS - 1) / BS, ((n + BS - 1) / BS) * BS, nx, nu, N, t};
    w.own = t < n;
    w.h = w.own ? 0.5 * (p.sh[p.so.ub + t % nu] - p.sh[p.so.lb + t % nu]) : 1.0;
    w.ctr = w.own ? 0.5 * (p.sh[p.so.ub + t % nu] + p.sh[p.so.lb + t % nu]) : 0.0;
    w.vr = 0.0;
    w.act_prev = 0.0; w.v = 0.0; w.qs = 0.0; w.sl = w.su = w.zl = w.zu = 1.0; w.rd = 0.0;
    w.setup(b);
    const double *sh = p.sh;
    double *xs = lds + w.o.xs;
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
    // V_N by rolling the model forward with the optimal inputs (thread 0; tiny)
    auto value_fn = [&](const double *x0v) -> double {
        double cost = 0.0;
        if (t == 0) {
            double xsv[16], xn[16];
            for (int i = 0; i < nx; ++i) xsv[i] = x0v[i];
            for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) cost = __builtin_fma(xsv[i] * sh[p.so.Q + i * nx + j], xsv[j], cost);
            const double *A = lds + w.o.Xf, *Bm = lds + w.o.M;
            for (int s = 0; s < N; ++s) {
                for (int i = 0; i < nx; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < nx; ++j) acc = __builtin_fma(A[i * nx + j], xsv[j], acc);
                    for (int k = 0; k < nu; ++k) acc = __builtin_fma(Bm[i * nu + k], lds[w.o.vw + s * nu + k], acc);
                    xn[i] = acc;
                }
                const int oQ = (s < N - 1) ? p.so.Q : p.so.P;
                for (int i = 0; i < nx; ++i) xsv[i] = xn[i];
                if (p.has_ref) for (int i = 0; i < nx; ++i) xn[i] -= sh[p.so.xref + i * N + s];
                for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) cost = __builtin_fma(xn[i] * sh[oQ + i * nx + j], xn[j], cost);
                for (int k = 0; k < nu; ++k) for (int j = 0; j < nu; ++j) {
                    const double dk = lds[w.o.vw + s * nu + k] - (p.has_ref ? sh[p.so.uref + k * N + s] : 0.0);
                    const double dj = lds[w.o.vw + s * nu + j] - (p.has_ref ? sh[p.so.uref + j * N + s] : 0.0);
                    cost = __builtin_fma(dk * sh[p.so.R + k * nu + j], dj, cost);
                }
            }
        }
        return cost;
    };
    if (p.mode == MODE_SOLVE) {
        load_x0();
        status = w.solve_qp(iters);
        publish_v();
        const double vn = value_fn(xs);
        if (t == 0) p.VN[b] = vn;
        if (t < nu) p.u0[(long long)t * Bsz + b] = lds[w.o.vw + t];
    } else if (p.mode == MODE_MAXVN) {
        double best = -1e308;
        for (int k = 0; k < p.K; ++k) {
            __syncthreads();
            if (t < nx) xs[t] = sh[p.so.x0s + t * p.K + k];
            __syncthreads();
            w.act_prev = 0.0;
            const int st = w.solve_qp(iters);
            status = st > status ? st : status;
            publish_v();
            const double vn = value_fn(xs);
            best = (vn > best || vn != vn) ? vn : best;
        }
        if (t == 0) p.MV[b] = best;
    } else {
        load_x0();
        double cost = 0.0;
        if (t == 0) for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) cost = __builtin_fma(xs[i] * sh[p.so.Q + i * nx + j], xs[j], cost);
        if (p.X && t < nx) p.X[((long long)t * (p.T + 1)) * Bsz + b] = xs[t];
        for (int step = 0; step < p.T; ++step) {
            const int st = w.solve_qp(iters);
            status = st > status ? st : status;
            publish_v();
            double xn = 0.0;
            if (t < nx) {
                for (int j = 0; j < nx; ++j) xn = __builtin_fma(p.true_per_instance ? p.At[(long long)(t * nx + j) * Bsz + b] : sh[p.so.At + t * nx + j], xs[j], xn);
                for (int k = 0; k < nu; ++k) xn = __builtin_fma(p.true_per_instance ? p.Bt[(long long)(t * nu + k) * Bsz + b] : sh[p.so.Bt + t * nu + k], lds[w.o.vw + k], xn);
            }
            __syncthreads();
            if (t < nx) xs[t] = xn;
            __syncthreads();
            if (t == 0) {
                for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) cost = __builtin_fma(xs[i] * sh[p.so.Q + i * nx + j], xs[j], cost);
                for (int k = 0; k < nu; ++k) for (int j = 0; j < nu; ++j) cost = __builtin_fma(lds[w.o.vw + k] * sh[p.so.R + k * nu + j], lds[w.o.vw + j], cost);
            }
            if (p.X && t < nx) p.X[((long long)t * (p.T + 1) + step + 1) * Bsz + b] = xs[t];
            if (p.U && t < nu) p.U[((long long)t * p.T + step) * Bsz + b] = lds[w.o.vw + t];
        }
        if (t == 0) p.JT[b] = cost;
    }
    if (t == 0) {
        if (p.status) p.status[b] = status;
        if (p.iters) p.iters[b] = iters;
    }
}

bool launch_wg(const KParams &p, hipStream_t stream, const char **name)
{
    const size_t bytes = wg_lds_bytes(p.nx, p.nu, p.N);
    static size_t attr_bytes = 0;            // dynamic-LDS opt-in, raised when a call needs more
    if (bytes > attr_bytes) {
        const hipError_t e = hipFuncSetAttribute((const void *)lqmpc_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            fprintf(stderr, "lqmpc: hipFuncSetAttribute(%zu bytes of LDS): %s\n", bytes, hipGetErrorString(e));
            return false;
        }
        attr_bytes = bytes;
    }
    hipLaunchKernelGGL(lqmpc_wg_kernel, dim3((unsigned)p.Bsz), dim3(256), bytes, stream, p);
    if (name) *name = "lqmpc_wg_kernel";
    return true;
}

}  // namespace lqmpc
