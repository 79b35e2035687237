// lqmpc_generic.hip -- any-dimension kernel: one (A,B) instance per wavefront lane, runtime loop
// bounds, per-instance scratch in an HBM workspace laid out instance-minor so that every scratch
// access of a wavefront is one coalesced 512-byte row.  This is the coverage path (arbitrary
// nx, nu, N up to the limits in include/lqmpc.h); the register-resident specialisations in
// lqmpc_spec.hip are the throughput path for the BASELINE configs.
//
// Algorithm per instance (restating /root/reference/utils_class.py:48-91 and 245-285):
//   condense once:  P = 2(Gamma'Qbar Gamma + Rbar)  (packed lower),  Fq = 2 Gamma'Qbar Phi,
//                   qref = reference terms, qc = P*centre (box shifted to |v| <= h)
//   per QP:         q = Fq x + qref + qc;  Mehrotra predictor-corrector interior point on
//                   min 1/2 v'Pv + q'v, |v| <= h, with a dense Cholesky of P + diag(z/s) per
//                   iteration; optional polish = exact solve on the identified active set.
#include "lqmpc_common.h"

namespace lqmpc {

struct GenOff {
    int M, Ap, W, P, L, F, qr, v, sl, su, zl, zu, rd, q, isl, isu, dva, dv, act, x, xt, lam, D, total;
};

__host__ __device__ inline GenOff gen_offsets(int nx, int nu, int N)
{
    const int n = N * nu, tri = n * (n + 1) / 2;
    GenOff o;
    int c = 0;
    o.M = c;   c += N * nx * nu;
    o.Ap = c;  c += 2 * nx * nx;
    o.W = c;   c += nx * nu;
    o.P = c;   c += tri;
    o.L = c;   c += tri;
    o.F = c;   c += n * nx;
    o.qr = c;  c += n;
    o.v = c;   c += n;
    o.sl = c;  c += n;
    o.su = c;  c += n;
    o.zl = c;  c += n;
    o.zu = c;  c += n;
    o.rd = c;  c += n;
    o.q = c;   c += n;
    o.isl = c; c += n;
    o.isu = c; c += n;
    o.dva = c; c += n;
    o.dv = c;  c += n;
    o.act = c; c += n;
    o.x = c;   c += nx;
    o.xt = c;  c += nx;
    o.lam = c; c += nx;
    o.D = c;   c += N * nx;
    o.total = c;
    return o;
}

long long generic_ws_entries(int nx, int nu, int N) { return gen_offsets(nx, nu, N).total; }

#define WS(off, e) ws[((long long)(off) + (long long)(e)) * stride + b]
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))

struct Ctx {
    const KParams &p;
    GenOff o;
    double *ws;
    long long b, stride;
    int nx, nu, N, n;
};

// In-place packed Cholesky of WS(L): on exit the strict lower part holds L and the diagonal
// holds 1/l_kk.  Returns false on a non-positive pivot.
__device__ static bool chol_packed(const Ctx &c)
{
    double *ws = c.ws; const long long b = c.b, stride = c.stride; const int n = c.n, oL = c.o.L;
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double d = WS(oL, TRI(j, j));
        for (int k = 0; k < j; ++k) { double l = WS(oL, TRI(j, k)); d = __builtin_fma(-l, l, d); }
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double inv = frsqrt(d);
        WS(oL, TRI(j, j)) = inv;
        for (int i = j + 1; i < n; ++i) {
            double s = WS(oL, TRI(i, j));
            for (int k = 0; k < j; ++k) s = __builtin_fma(-WS(oL, TRI(i, k)), WS(oL, TRI(j, k)), s);
            WS(oL, TRI(i, j)) = s * inv;
        }
    }
    return ok;
}

// Solve (L L') y = rhs in place on the vector at offset ov.
__device__ static void chol_solve_packed(const Ctx &c, int ov)
{
    double *ws = c.ws; const long long b = c.b, stride = c.stride; const int n = c.n, oL = c.o.L;
    for (int i = 0; i < n; ++i) {
        double s = WS(ov, i);
        for (int k = 0; k < i; ++k) s = __builtin_fma(-WS(oL, TRI(i, k)), WS(ov, k), s);
        WS(ov, i) = s * WS(oL, TRI(i, i));
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = WS(ov, i);
        for (int k = i + 1; k < n; ++k) s = __builtin_fma(-WS(oL, TRI(k, i)), WS(ov, k), s);
        WS(ov, i) = s * WS(oL, TRI(i, i));
    }
}

__device__ static inline double psym(const Ctx &c, int i, int j)
{
    double *ws = c.ws; const long long b = c.b, stride = c.stride;
    return i >= j ? WS(c.o.P, TRI(i, j)) : WS(c.o.P, TRI(j, i));
}

// ---- condensing: utils_class.py:62-75 in matrix form ----
__device__ static void condense(const Ctx &c)
{
    const KParams &p = c.p; double *ws = c.ws; const long long b = c.b, stride = c.stride, Bsz = p.Bsz;
    const int nx = c.nx, nu = c.nu, N = c.N, n = c.n; const GenOff &o = c.o;
    const double *sh = p.sh;
    const int tri = n * (n + 1) / 2;
    // M_0 = B, M_k = A M_{k-1}
    for (int e = 0; e < nx * nu; ++e) WS(o.M, e) = p.B[(long long)e * Bsz + b];
    for (int k = 1; k < N; ++k)
        for (int x = 0; x < nx; ++x)
            for (int u = 0; u < nu; ++u) {
                double s = 0.0;
                for (int y = 0; y < nx; ++y) s = __builtin_fma(p.A[(long long)(x * nx + y) * Bsz + b], WS(o.M, ((k - 1) * nx + y) * nu + u), s);
                WS(o.M, (k * nx + x) * nu + u) = s;
            }
    for (int e = 0; e < tri; ++e) WS(o.P, e) = 0.0;
    for (int e = 0; e < n * nx; ++e) WS(o.F, e) = 0.0;
    for (int x = 0; x < nx; ++x) for (int y = 0; y < nx; ++y) WS(o.Ap, x * nx + y) = (x == y) ? 1.0 : 0.0;
    int cur = 0;
    for (int r = 0; r < N; ++r) {
        const int nxt = cur ^ 1;     // Ap[nxt] = A * Ap[cur] = A^{r+1}
        for (int x = 0; x < nx; ++x)
            for (int y = 0; y < nx; ++y) {
                double s = 0.0;
                for (int t = 0; t < nx; ++t) s = __builtin_fma(p.A[(long long)(x * nx + t) * Bsz + b], WS(o.Ap, cur * nx * nx + t * nx + y), s);
                WS(o.Ap, nxt * nx * nx + x * nx + y) = s;
            }
        cur = nxt;
        const double *Qr = sh + ((r < N - 1) ? p.so.Q : p.so.P);   // terminal weight on x_N (lines 67-72)
        for (int bi = 0; bi <= r; ++bi) {
            for (int x = 0; x < nx; ++x)
                for (int u = 0; u < nu; ++u) {
                    double s = 0.0;
                    for (int y = 0; y < nx; ++y) s = __builtin_fma(Qr[x * nx + y], WS(o.M, ((r - bi) * nx + y) * nu + u), s);
                    WS(o.W, x * nu + u) = s;
                }
            for (int bj = 0; bj <= bi; ++bj)
                for (int ui = 0; ui < nu; ++ui)
                    for (int uj = 0; uj < nu; ++uj) {
                        const int i = bi * nu + ui, j = bj * nu + uj;
                        if (j > i) continue;
                        double s = WS(o.P, TRI(i, j));
                        for (int x = 0; x < nx; ++x) s = __builtin_fma(WS(o.W, x * nu + ui), WS(o.M, ((r - bj) * nx + x) * nu + uj), s);
                        WS(o.P, TRI(i, j)) = s;
                    }
            for (int ui = 0; ui < nu; ++ui)
                for (int y = 0; y < nx; ++y) {
                    double s = WS(o.F, (bi * nu + ui) * nx + y);
                    for (int x = 0; x < nx; ++x) s = __builtin_fma(WS(o.W, x * nu + ui), WS(o.Ap, cur * nx * nx + x * nx + y), s);
                    WS(o.F, (bi * nu + ui) * nx + y) = s;
                }
        }
    }
    // P = 2 (H + Rbar), Fq = 2 F
    const double *R = sh + p.so.R;
    for (int i = 0; i < N; ++i)
        for (int a = 0; a < nu; ++a)
            for (int bb = 0; bb <= a; ++bb) WS(o.P, TRI(i * nu + a, i * nu + bb)) += R[a * nu + bb];
    for (int e = 0; e < tri; ++e) WS(o.P, e) *= 2.0;
    for (int e = 0; e < n * nx; ++e) WS(o.F, e) *= 2.0;
    // qr = 2 gref + P*centre;  gref via the costate recursion with x0 = 0:
    //   d_r = -xref_r,  lam_r = Qr d_r + A' lam_{r+1},  gref_r = B' lam_r - R uref_r
    for (int i = 0; i < n; ++i) WS(o.qr, i) = 0.0;
    if (p.has_ref) {
        const double *xr = sh + p.so.xref, *ur = sh + p.so.uref;
        for (int x = 0; x < nx; ++x) WS(o.lam, x) = 0.0;
        for (int r = N - 1; r >= 0; --r) {
            const double *Qr = sh + ((r < N - 1) ? p.so.Q : p.so.P);
            for (int x = 0; x < nx; ++x) {
                double s = 0.0;
                for (int y = 0; y < nx; ++y) s = __builtin_fma(Qr[x * nx + y], -xr[y * N + r], s);
                for (int y = 0; y < nx; ++y) s = __builtin_fma(p.A[(long long)(y * nx + x) * Bsz + b], WS(o.lam, y), s);
                WS(o.xt, x) = s;
            }
            for (int x = 0; x < nx; ++x) WS(o.lam, x) = WS(o.xt, x);
            for (int k = 0; k < nu; ++k) {
                double s = 0.0;
                for (int x = 0; x < nx; ++x) s = __builtin_fma(p.B[(long long)(x * nu + k) * Bsz + b], WS(o.lam, x), s);
                for (int j = 0; j < nu; ++j) s = __builtin_fma(-R[k * nu + j], ur[j * N + r], s);
                WS(o.qr, r * nu + k) = 2.0 * s;
            }
        }
    }
    const double *lb = sh + p.so.lb, *ub = sh + p.so.ub;
    for (int i = 0; i < n; ++i) {
        double s = WS(o.qr, i);
        for (int j = 0; j < n; ++j) { const int k = j % nu; s = __builtin_fma(psym(c, i, j), 0.5 * (lb[k] + ub[k]), s); }
        WS(o.qr, i) = s;
    }
}

// ---- interior-point iterations from the current iterate until the gap / residual reach eps_rel ----
// (Mehrotra predictor-corrector; the iterate lives in WS(sl, su, zl, zu, rd); `it` counts iterations of the
// whole QP against max_iter).  Returns 0 converged, 1 iteration cap, 2 non-finite.
__device__ static int ipm_stage(const Ctx &c, double scale, double eps_rel, int &it)
{
    const KParams &p = c.p; double *ws = c.ws; const long long b = c.b, stride = c.stride;
    const int n = c.n; const GenOff &o = c.o;
    const double inv2n = 1.0 / (2.0 * n);
    int status = 1;
    for (; it <= p.max_iter; ++it) {
        double mu = 0.0, rn = 0.0, hmin = 1e300;
        for (int i = 0; i < n; ++i) {
            mu += WS(o.sl, i) * WS(o.zl, i) + WS(o.su, i) * WS(o.zu, i);
            rn = fmax(rn, fabs(WS(o.rd, i)));
            hmin = fmin(hmin, WS(o.sl, i) + WS(o.su, i));
        }
        mu *= inv2n;
        if (!(mu < 1e300) || !(rn < 1e300)) { status = 2; break; }
        if (mu <= eps_rel * scale * 0.5 * hmin && rn <= eps_rel * scale) { status = 0; break; }
        if (it == p.max_iter) break;
        // K = P + diag(zl/sl + zu/su), factor
        for (int i = 0; i < n; ++i) {
            const double isl = frcp(WS(o.sl, i)), isu = frcp(WS(o.su, i));
            WS(o.isl, i) = isl; WS(o.isu, i) = isu;
            for (int j = 0; j < i; ++j) WS(o.L, TRI(i, j)) = WS(o.P, TRI(i, j));
            WS(o.L, TRI(i, i)) = WS(o.P, TRI(i, i)) + WS(o.zl, i) * isl + WS(o.zu, i) * isu;
        }
        if (!chol_packed(c)) { status = 2; break; }
        // predictor: K dva = -(rd + zl - zu)
        for (int i = 0; i < n; ++i) WS(o.dva, i) = -WS(o.rd, i) - WS(o.zl, i) + WS(o.zu, i);
        chol_solve_packed(c, o.dva);
        double mp = 0.0, md = 0.0;   // largest -dx/x over primal slacks / duals
        for (int i = 0; i < n; ++i) {
            const double d = WS(o.dva, i), isl = WS(o.isl, i), isu = WS(o.isu, i);
            mp = fmax(mp, fmax(-d * isl, d * isu));
            md = fmax(md, fmax(1.0 + d * isl, 1.0 - d * isu));   // -dz_aff/z
        }
        const double apa = mp > 1.0 ? 1.0 / mp : 1.0, ada = md > 1.0 ? 1.0 / md : 1.0;
        double mua = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = WS(o.dva, i), zl = WS(o.zl, i), zu = WS(o.zu, i);
            const double dzl = -zl - zl * WS(o.isl, i) * d, dzu = -zu + zu * WS(o.isu, i) * d;
            mua += (WS(o.sl, i) + apa * d) * (zl + ada * dzl) + (WS(o.su, i) - apa * d) * (zu + ada * dzu);
        }
        mua *= inv2n;
        double sg = mua / mu; sg = sg * sg * sg;
        const double smu = sg * mu;
        // corrector
        for (int i = 0; i < n; ++i) {
            const double d = WS(o.dva, i), zl = WS(o.zl, i), zu = WS(o.zu, i), isl = WS(o.isl, i), isu = WS(o.isu, i);
            const double dzl = -zl - zl * isl * d, dzu = -zu + zu * isu * d;
            const double rcl = smu - WS(o.sl, i) * zl - d * dzl, rcu = smu - WS(o.su, i) * zu + d * dzu;
            WS(o.dv, i) = -WS(o.rd, i) + rcl * isl - rcu * isu;
        }
        chol_solve_packed(c, o.dv);
        mp = 0.0; md = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = WS(o.dva, i), dv = WS(o.dv, i), zl = WS(o.zl, i), zu = WS(o.zu, i), isl = WS(o.isl, i), isu = WS(o.isu, i);
            const double dzla = -zl - zl * isl * d, dzua = -zu + zu * isu * d;
            const double rcl = smu - WS(o.sl, i) * zl - d * dzla, rcu = smu - WS(o.su, i) * zu + d * dzua;
            const double dzl = (rcl - zl * dv) * isl, dzu = (rcu + zu * dv) * isu;
            WS(o.isl, i) = dzl; WS(o.isu, i) = dzu;   // reuse as step storage
            mp = fmax(mp, fmax(-dv * isl, dv * isu));
            md = fmax(md, fmax(-dzl / zl, -dzu / zu));
        }
        // one step length for primal and dual (unequal lengths can make the QP's dual residual oscillate)
        mp = fmax(mp, md);
        const double ap = mp > p.tau ? p.tau / mp : 1.0, ad = ap;
        for (int i = 0; i < n; ++i) {
            const double dv = WS(o.dv, i), dzl = WS(o.isl, i), dzu = WS(o.isu, i);
            WS(o.v, i) += ap * dv; WS(o.sl, i) += ap * dv; WS(o.su, i) -= ap * dv;
            WS(o.zl, i) += ad * dzl; WS(o.zu, i) += ad * dzu;
            WS(o.rd, i) = (1.0 - ap) * WS(o.rd, i) + (ap - ad) * (dzl - dzu);
        }
    }
    return status;
}

// ---- primal-dual active-set iterations from the face suggested by the interior-point iterate ----
// One iteration solves the QP on the face (WS(act): -1 lower, +1 upper, 0 free) exactly and re-derives the
// face from the KKT signs; a fixed point is the exact optimum and is committed to WS(v).
__device__ static bool pdas(const Ctx &c, double scale, int maxit, int *iters)
{
    const KParams &p = c.p; double *ws = c.ws; const long long b = c.b, stride = c.stride;
    const int nu = c.nu, n = c.n; const GenOff &o = c.o;
    const double *lb = p.sh + p.so.lb, *ub = p.sh + p.so.ub;
    const double gt = 1e-10 * scale;
    for (int i = 0; i < n; ++i)
        WS(o.act, i) = (WS(o.zl, i) > WS(o.sl, i)) ? -1.0 : ((WS(o.zu, i) > WS(o.su, i)) ? 1.0 : 0.0);
    for (int k = 0; k < maxit; ++k) {
        *iters += 1;
        for (int i = 0; i < n; ++i) {
            const int kk = i % nu; const double h = 0.5 * (ub[kk] - lb[kk]);
            WS(o.dva, i) = WS(o.act, i) * h;   // bound value (0 for free)
        }
        for (int i = 0; i < n; ++i) {
            const bool ai = WS(o.act, i) != 0.0;
            double r = ai ? WS(o.dva, i) : -WS(o.q, i);
            for (int j = 0; j < n; ++j) {
                const bool aj = WS(o.act, j) != 0.0;
                if (!ai && aj) r = __builtin_fma(-psym(c, i, j), WS(o.dva, j), r);
                if (j <= i) WS(o.L, TRI(i, j)) = (ai || aj) ? ((i == j) ? 1.0 : 0.0) : WS(o.P, TRI(i, j));
            }
            WS(o.dv, i) = r;
        }
        if (!chol_packed(c)) return false;
        chol_solve_packed(c, o.dv);
        bool changed = false;
        for (int i = 0; i < n; ++i) {
            const int kk = i % nu; const double h = 0.5 * (ub[kk] - lb[kk]);
            const double a = WS(o.act, i), vi = WS(o.dv, i);
            if (!(fabs(vi) < 1e300)) return false;
            double na;
            if (a == 0.0) na = (vi < -h * (1.0 + 1e-12)) ? -1.0 : ((vi > h * (1.0 + 1e-12)) ? 1.0 : 0.0);
            else {
                double gi = WS(o.q, i);
                for (int j = 0; j < n; ++j) gi = __builtin_fma(psym(c, i, j), WS(o.dv, j), gi);
                na = (a < 0.0) ? ((gi >= -gt) ? -1.0 : 0.0) : ((gi <= gt) ? 1.0 : 0.0);
            }
            WS(o.isl, i) = na;              // new face, applied after the sweep
            changed = changed || (na != a);
        }
        if (!changed) {
            for (int i = 0; i < n; ++i) WS(o.v, i) = WS(o.dv, i);
            return true;
        }
        for (int i = 0; i < n; ++i) WS(o.act, i) = WS(o.isl, i);
    }
    return false;
}

// ---- one box QP at the state in WS(x): result v (shifted input sequence) in WS(v) ----
// Interior point in stages (gap 1e-6, 1e-9, eps); after each stage the active-set iterations try to finish
// exactly from the face the iterate suggests.  Returns status (0 ok, 1 iteration cap, 2 non-finite) and adds
// the number of KKT factorisations to *iters.
__device__ static int solve_qp(const Ctx &c, int *iters)
{
    const KParams &p = c.p; double *ws = c.ws; const long long b = c.b, stride = c.stride;
    const int nx = c.nx, nu = c.nu, n = c.n; const GenOff &o = c.o;
    const double *lb = p.sh + p.so.lb, *ub = p.sh + p.so.ub;
    double scale = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = WS(o.qr, i);
        for (int a = 0; a < nx; ++a) s = __builtin_fma(WS(o.F, i * nx + a), WS(o.x, a), s);
        WS(o.q, i) = s;
        scale = fmax(scale, fabs(s));
    }
    if (!(scale < 1e300)) { for (int i = 0; i < n; ++i) WS(o.v, i) = 0.0; return 2; }
    scale = fmax(scale, 1e-100);
    const double z0 = p.z0_scale * scale;
    for (int i = 0; i < n; ++i) {
        const int k = i % nu; const double h = 0.5 * (ub[k] - lb[k]);
        WS(o.v, i) = 0.0; WS(o.sl, i) = h; WS(o.su, i) = h;
        WS(o.zl, i) = z0; WS(o.zu, i) = z0; WS(o.rd, i) = WS(o.q, i);
    }
    int status = 1, it = 0;
    double e_prev = 1e300;
    for (int stage = 0; stage < 3; ++stage) {
        const double e = !p.polish ? p.eps : (stage == 0 ? fmax(1e-6, p.eps) : (stage == 1 ? fmax(1e-9, p.eps) : p.eps));
        if (!(e < e_prev)) continue;
        e_prev = e;
        const int it0 = it;
        status = ipm_stage(c, scale, e, it);
        *iters += it - it0;
        if (status == 2) { for (int i = 0; i < n; ++i) WS(o.v, i) = 0.0; return 2; }
        // WS(v) holds the interior-point iterate (tracked alongside the slacks)
        if (!p.polish) break;
        if (pdas(c, scale, 6, iters)) { status = 0; break; }
        if (it >= p.max_iter) break;
    }
    return status;
}

__device__ static inline double u_of(const Ctx &c, int i)
{
    double *ws = c.ws; const long long b = c.b, stride = c.stride;
    const double *lb = c.p.sh + c.p.so.lb, *ub = c.p.sh + c.p.so.ub; const int k = i % c.nu;
    double u = WS(c.o.v, i) + 0.5 * (lb[k] + ub[k]);
    return fmin(fmax(u, lb[k]), ub[k]);
}

// V_N = cost* + x0'Qx0 by rolling the MODEL forward with the optimal inputs (utils_class.py:62-75, 91)
__device__ static double value_fn(const Ctx &c)
{
    const KParams &p = c.p; double *ws = c.ws; const long long b = c.b, stride = c.stride, Bsz = p.Bsz;
    const int nx = c.nx, nu = c.nu, N = c.N; const GenOff &o = c.o;
    const double *sh = p.sh, *Q = sh + p.so.Q, *R = sh + p.so.R;
    double cost = 0.0;
    for (int a = 0; a < nx; ++a) for (int d = 0; d < nx; ++d) cost = __builtin_fma(WS(o.x, a) * Q[a * nx + d], WS(o.x, d), cost);
    for (int a = 0; a < nx; ++a) WS(o.xt, a) = WS(o.x, a);
    for (int i = 0; i < N; ++i) {
        const double *Qi = sh + ((i < N - 1) ? p.so.Q : p.so.P);
        for (int a = 0; a < nx; ++a) {
            double s = 0.0;
            for (int d = 0; d < nx; ++d) s = __builtin_fma(p.A[(long long)(a * nx + d) * Bsz + b], WS(o.xt, d), s);
            for (int k = 0; k < nu; ++k) s = __builtin_fma(p.B[(long long)(a * nu + k) * Bsz + b], u_of(c, i * nu + k), s);
            WS(o.lam, a) = s;
        }
        for (int a = 0; a < nx; ++a) WS(o.xt, a) = WS(o.lam, a);
        for (int a = 0; a < nx; ++a) WS(o.lam, a) = WS(o.xt, a) - (p.has_ref ? sh[p.so.xref + a * N + i] : 0.0);
        for (int a = 0; a < nx; ++a) for (int d = 0; d < nx; ++d) cost = __builtin_fma(WS(o.lam, a) * Qi[a * nx + d], WS(o.lam, d), cost);
        for (int k = 0; k < nu; ++k)
            for (int j = 0; j < nu; ++j) {
                const double dk = u_of(c, i * nu + k) - (p.has_ref ? sh[p.so.uref + k * N + i] : 0.0);
                const double dj = u_of(c, i * nu + j) - (p.has_ref ? sh[p.so.uref + j * N + i] : 0.0);
                cost = __builtin_fma(dk * R[k * nu + j], dj, cost);
            }
    }
    return cost;
}

// one instance: b = its index in the caller's arrays, col = its workspace column
__device__ static void generic_instance(const KParams &p, long long b, long long col);

__global__ void __launch_bounds__(64) lqmpc_generic_kernel(KParams p)
{
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    if (b >= p.Bsz) return;
    generic_instance(p, b, b);
}

// over a device-side list (p.perm, *p.count_dev entries: what a 16-lane-row kernel handed back): every thread owns one workspace
// column and walks the list with the stride of the grid
__global__ void __launch_bounds__(64) lqmpc_generic_list_kernel(KParams p)
{
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x, cols = (long long)gridDim.x * 64;
    const long long count = *p.count_dev;
    for (long long s = col; s < count; s += cols) generic_instance(p, p.perm[s], col);
}

__device__ static void generic_instance(const KParams &p, long long b, long long col)
{
    // (the workspace macro indexes by b: shift the base so that column `col` is used)
    Ctx c{p, gen_offsets(p.nx, p.nu, p.N), p.ws + (col - b), b, p.ws_stride, p.nx, p.nu, p.N, p.n};
    double *ws = c.ws; const long long stride = c.stride, Bsz = p.Bsz;
    const int nx = c.nx, nu = c.nu; const GenOff &o = c.o;
    const double *sh = p.sh, *Q = sh + p.so.Q, *R = sh + p.so.R;
    condense(c);
    int iters = 0, status = 0;
    if (p.mode == MODE_SOLVE) {
        for (int a = 0; a < nx; ++a) WS(o.x, a) = p.x0[(long long)a * Bsz + b];
        status = solve_qp(c, &iters);
        for (int k = 0; k < nu; ++k) p.u0[(long long)k * Bsz + b] = u_of(c, k);
        p.VN[b] = value_fn(c);
    } else if (p.mode == MODE_MAXVN) {
        double best = -1e308;
        for (int k = 0; k < p.K; ++k) {
            for (int a = 0; a < nx; ++a) WS(o.x, a) = sh[p.so.x0s + a * p.K + k];
            const int st = solve_qp(c, &iters);
            status = st > status ? st : status;
            const double v = value_fn(c);
            best = (v > best || v != v) ? v : best;
        }
        p.MV[b] = best;
    } else {
        for (int a = 0; a < nx; ++a) WS(o.x, a) = p.x0[(long long)a * Bsz + b];
        double cost = 0.0;                                                   // utils_class.py:261
        for (int a = 0; a < nx; ++a) for (int d = 0; d < nx; ++d) cost = __builtin_fma(WS(o.x, a) * Q[a * nx + d], WS(o.x, d), cost);
        if (p.X) for (int a = 0; a < nx; ++a) p.X[((long long)a * (p.T + 1)) * Bsz + b] = WS(o.x, a);
        for (int t = 0; t < p.T; ++t) {                                      // utils_class.py:266-283
            const int st = solve_qp(c, &iters);
            status = st > status ? st : status;
            for (int a = 0; a < nx; ++a) {                                   // line 277: plant step
                double s = 0.0;
                for (int d = 0; d < nx; ++d) {
                    const double at = p.true_per_instance ? p.At[(long long)(a * nx + d) * Bsz + b] : sh[p.so.At + a * nx + d];
                    s = __builtin_fma(at, WS(o.x, d), s);
                }
                for (int k = 0; k < nu; ++k) {
                    const double bt = p.true_per_instance ? p.Bt[(long long)(a * nu + k) * Bsz + b] : sh[p.so.Bt + a * nu + k];
                    s = __builtin_fma(bt, u_of(c, k), s);
                }
                WS(o.xt, a) = s;
            }
            for (int a = 0; a < nx; ++a) WS(o.x, a) = WS(o.xt, a);
            for (int a = 0; a < nx; ++a) for (int d = 0; d < nx; ++d) cost = __builtin_fma(WS(o.x, a) * Q[a * nx + d], WS(o.x, d), cost);   // 282
            for (int k = 0; k < nu; ++k) for (int j = 0; j < nu; ++j) cost = __builtin_fma(u_of(c, k) * R[k * nu + j], u_of(c, j), cost);  // 283
            if (p.X) for (int a = 0; a < nx; ++a) p.X[((long long)a * (p.T + 1) + t + 1) * Bsz + b] = WS(o.x, a);
            if (p.U) for (int k = 0; k < nu; ++k) p.U[((long long)k * p.T + t) * Bsz + b] = u_of(c, k);
        }
        p.JT[b] = cost;
    }
    if (p.status) p.status[b] = status;
    if (p.iters) p.iters[b] = iters;
}

void launch_generic_list(const KParams &p, int cols, hipStream_t stream)
{
    hipLaunchKernelGGL(lqmpc_generic_list_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, stream, p);
}

void launch_generic(const KParams &p, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.Bsz + 63) / 64);
    hipLaunchKernelGGL(lqmpc_generic_kernel, dim3(grid), dim3(64), 0, stream, p);
}

}  // namespace lqmpc
