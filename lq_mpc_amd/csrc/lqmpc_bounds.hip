// lqmpc_bounds.hip -- per-system coefficients of the reference's performance bound, batched on the GPU.
//
// What one instance computes (one instance per wavefront lane, run-time dimensions, per-instance matrices in an HBM
// workspace laid out instance-minor so that every access of a wavefront is one coalesced 512-byte row):
//   K        = dlqr(A, B, Q, R)                       /root/reference/utils_class.py:761, 840, 923 (control.dlqr)
//              P_inf by the structure-preserving doubling iteration in its symmetric form
//                 S = (H^-1 + G)^-1,  A+ = A H^-1 S A,  G+ = G + A H^-1 S G A',  H+ = H + A' S A,   G0 = B R^-1 B', H0 = Q
//              (two SPD inverses per step, quadratic convergence, H -> P_inf), K = (R + B'PB)^-1 B'PA
//   eps      = local_radius(F_u, -K, Q)               utils.py:548-564    (box rows of F_u: e_k / ub_k and e_k / lb_k)
//   gamma, rho_gamma = ex_stability_lq                utils.py:343-380    (spectral radius of A - BK by repeated squaring:
//                                                      log rho = sum_j 2^-j log |Y_j|_F with Y_{j+1} = (Y_j / |Y_j|)^2)
//   L_V, N_0 = ex_stability_bounds                    utils.py:567-584
//   omega_{N,1}, omega_{N,0.5}, eta, h, xi            utils.py:393-409, 469-538; utils_class.py:344-373
//   alpha, beta = energy_bound                        utils_class.py:308-342; utils.py:78-117, 186-334
//              |Phi|_2, |A|_2, |B|_2, |K|_2 from cyclic Jacobi sweeps on the small Gram matrices; |Gamma|_2 and lambda_min(hat H)
//              (Gamma' Gamma, hat H are N n_u x N n_u) from a Householder tridiagonalisation and bisection on the Sturm count; bar_u, bar_d_u in closed form for a box (utils.py:592-650 uses Gurobi)
//   bound    = (alpha V_expert + beta) / (1 - xi - eta)                                    utils_class.py:858-859
// Two quirks of the reference are kept because its golden data contain them (the constant 1.21 and the "+0.4" of
// utils.py:358, 364) and so is the Kronecker ordering of hat H (utils.py:316-319, np.kron(R, I_N) + Gamma' np.kron(Q, I_{N+1}) Gamma);
// one is not: the closed loop is A + B K as a matrix product (utils.py:356 multiplies elementwise, the same thing for n_u = 1).
#include "lqmpc_common.h"
#include "lqmpc_bounds.h"
#include "lqmpc_bounds_chip.h"

namespace lqmpc {

BoundsOff bounds_offsets(int nx, int nu, int N)
{
    const int n = N * nu;
    BoundsOff o;
    int c = 0;
    o.Ak = c; c += nx * nx;
    o.Gk = c; c += nx * nx;
    o.Hk = c; c += nx * nx;
    o.T1 = c; c += nx * nx;
    o.T2 = c; c += nx * nx;
    o.T3 = c; c += nx * nx;
    o.T4 = c; c += nx * nx;
    o.K = c;  c += nu * nx;
    o.U1 = c; c += nu * nu;
    o.U2 = c; c += nu * nx;
    o.Md = c; c += N * nx * nu;
    o.E = c;  c += n * n;
    o.V = c;  c += 4 * n;
    o.total = c;
    return o;
}

namespace {

struct Lane {
    double *ws;            // workspace, already offset to this instance
    long long stride;
    __device__ __forceinline__ double &operator()(int off, int e) const { return ws[(long long)(off + e) * stride]; }
};

// C (m x n) = op(A) (m x k) op(B) (k x n) [+ C if acc]; row-major with leading dimensions = the stored column counts
__device__ void mm(const Lane &w, int C, int A, bool ta, int B, bool tb, int m, int k, int n, bool acc)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double t = acc ? w(C, i * n + j) : 0.0;
            for (int l = 0; l < k; ++l) {
                const double a = ta ? w(A, l * m + i) : w(A, i * k + l);
                const double b = tb ? w(B, j * k + l) : w(B, l * n + j);
                t = __builtin_fma(a, b, t);
            }
            w(C, i * n + j) = t;
        }
}

__device__ void symmetrise(const Lane &w, int M, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            const double t = 0.5 * (w(M, i * n + j) + w(M, j * n + i));
            w(M, i * n + j) = t; w(M, j * n + i) = t;
        }
}

// in-place inverse of an SPD matrix by Gauss-Jordan elimination (no pivoting needed); false if a pivot is not positive
__device__ bool spd_inverse(const Lane &w, int M, int n)
{
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        const double d = w(M, k * n + k);
        ok = ok && (d > 0.0) && (d < 1e300);
        const double p = 1.0 / d;
        w(M, k * n + k) = 1.0;
        for (int j = 0; j < n; ++j) w(M, k * n + j) *= p;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = w(M, i * n + k);
            w(M, i * n + k) = 0.0;
            for (int j = 0; j < n; ++j) w(M, i * n + j) = __builtin_fma(-f, w(M, k * n + j), w(M, i * n + j));
        }
    }
    return ok;
}

// extreme eigenvalues of the symmetric n x n matrix at M (destroyed): cyclic Jacobi sweeps to machine precision
__device__ void sym_eig_extremes(const Lane &w, int M, int n, double &emax, double &emin)
{
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < n; ++p) {
            dg = __builtin_fma(w(M, p * n + p), w(M, p * n + p), dg);
            for (int q = p + 1; q < n; ++q) off = __builtin_fma(w(M, p * n + q), w(M, p * n + q), off);
        }
        if (!(off > 1e-32 * (dg + off))) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = w(M, p * n + q);
                if (fabs(apq) < 1e-300) continue;
                const double theta = (w(M, q * n + q) - w(M, p * n + p)) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(__builtin_fma(theta, theta, 1.0)));
                const double c = 1.0 / sqrt(__builtin_fma(t, t, 1.0)), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = w(M, k * n + p), akq = w(M, k * n + q);
                    w(M, k * n + p) = c * akp - s * akq;
                    w(M, k * n + q) = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = w(M, p * n + k), aqk = w(M, q * n + k);
                    w(M, p * n + k) = c * apk - s * aqk;
                    w(M, q * n + k) = s * apk + c * aqk;
                }
            }
    }
    emax = -1e308; emin = 1e308;
    for (int p = 0; p < n; ++p) {
        const double d = w(M, p * n + p);
        emax = fmax(emax, d); emin = fmin(emin, d);
    }
}

// extreme eigenvalues of the symmetric n x n matrix at M (destroyed) for the large matrices (Gamma'Gamma, hat H): Householder
// tridiagonalisation (~5 n^3 / 3 workspace accesses, against ~16 n^3 per Jacobi sweep and 6-10 sweeps) and bisection on the Sturm
// count of the tridiagonal matrix (d, e).  V: 4 n entries of scratch (v | p / w | d | e).
__device__ void sym_eig_extremes_tridiag(const Lane &w, int M, int n, int V, double &emax, double &emin)
{
    const int v = V, pw = V + n, d = V + 2 * n, e = V + 3 * n;
    if (n == 1) { emax = emin = w(M, 0); return; }
    for (int k = 0; k + 2 < n; ++k) {
        const int m = n - k - 1;                                     // order of the trailing block A22 = M[k+1.., k+1..]
        double nrm2 = 0.0;
        for (int r = 0; r < m; ++r) { const double x = w(M, (k + 1 + r) * n + k); nrm2 = __builtin_fma(x, x, nrm2); }
        const double x0 = w(M, (k + 1) * n + k);
        w(d, k) = w(M, k * n + k);
        const double tail2 = nrm2 - x0 * x0;
        if (!(tail2 > 0.0)) { w(e, k) = x0; continue; }              // nothing below the subdiagonal in this column
        const double alpha = (x0 > 0.0) ? -sqrt(nrm2) : sqrt(nrm2);
        const double r2 = 0.5 * (nrm2 - x0 * alpha), rinv = 1.0 / (2.0 * sqrt(r2));      // H = I - 2 v v',  |v| = 1
        for (int r = 0; r < m; ++r) w(v, r) = (w(M, (k + 1 + r) * n + k) - (r == 0 ? alpha : 0.0)) * rinv;
        w(e, k) = alpha;
        double K = 0.0;                                              // p = A22 v,  K = v'p
        for (int r = 0; r < m; ++r) {
            double acc = 0.0;
            for (int c = 0; c < m; ++c) acc = __builtin_fma(w(M, (k + 1 + r) * n + (k + 1 + c)), w(v, c), acc);
            w(pw, r) = acc;
            K = __builtin_fma(w(v, r), acc, K);
        }
        for (int r = 0; r < m; ++r) w(pw, r) = __builtin_fma(-K, w(v, r), w(pw, r));      // w = p - K v
        for (int r = 0; r < m; ++r) {                                // A22 -= 2 (v w' + w v')
            const double vr2 = 2.0 * w(v, r), wr2 = 2.0 * w(pw, r);
            for (int c = 0; c < m; ++c) {
                const int idx = (k + 1 + r) * n + (k + 1 + c);
                w(M, idx) = w(M, idx) - vr2 * w(pw, c) - wr2 * w(v, c);
            }
        }
    }
    w(d, n - 2) = w(M, (n - 2) * n + (n - 2));
    w(e, n - 2) = w(M, (n - 1) * n + (n - 2));
    w(d, n - 1) = w(M, (n - 1) * n + (n - 1));
    double lo = 1e308, hi = -1e308;                                  // Gershgorin
    for (int i = 0; i < n; ++i) {
        const double rad = (i > 0 ? fabs(w(e, i - 1)) : 0.0) + (i + 1 < n ? fabs(w(e, i)) : 0.0);
        lo = fmin(lo, w(d, i) - rad); hi = fmax(hi, w(d, i) + rad);
    }
    const double span = fmax(hi - lo, 1e-300), tiny = 1e-300 + 2.3e-16 * fmax(fabs(lo), fabs(hi));
    auto count_below = [&](double x) {                               // number of eigenvalues < x
        int cnt = 0;
        double q = w(d, 0) - x;
        if (q < 0.0) ++cnt;
        for (int i = 1; i < n; ++i) {
            if (fabs(q) < tiny) q = (q < 0.0) ? -tiny : tiny;
            const double ei = w(e, i - 1);
            q = w(d, i) - x - ei * ei / q;
            if (q < 0.0) ++cnt;
        }
        return cnt;
    };
    lo -= 1e-3 * span; hi += 1e-3 * span;
    double a = lo, b = hi;                                           // smallest: count(a) = 0, count(b) >= 1
    for (int it = 0; it < 120 && b - a > 4.5e-16 * fmax(fabs(a), fabs(b)); ++it) {
        const double mid = 0.5 * (a + b);
        if (count_below(mid) >= 1) b = mid; else a = mid;
    }
    emin = 0.5 * (a + b);
    a = lo; b = hi;                                                  // largest: count(a) <= n - 1, count(b) = n
    for (int it = 0; it < 120 && b - a > 4.5e-16 * fmax(fabs(a), fabs(b)); ++it) {
        const double mid = 0.5 * (a + b);
        if (count_below(mid) >= n) b = mid; else a = mid;
    }
    emax = 0.5 * (a + b);
}

// spectral norm of the m x k matrix at A (row-major) through the smaller Gram matrix, built at E (needs min(m,k)^2 entries)
__device__ double norm2(const Lane &w, int A, int m, int k, int E)
{
    double emax, emin;
    if (k <= m) mm(w, E, A, true, A, false, k, m, k, false);          // A'A
    else mm(w, E, A, false, A, true, m, k, m, false);                 // AA'
    sym_eig_extremes(w, E, k <= m ? k : m, emax, emin);
    return sqrt(fmax(emax, 0.0));
}

// spectral radius of the n x n matrix at X (destroyed; Y is scratch): |M^(2^J)|^(1/2^J) by repeated squaring of the normalised matrix
__device__ double spectral_radius(const Lane &w, int X, int Y, int n)
{
    double lg = 0.0, wgt = 1.0;
    for (int j = 0; j < 56; ++j) {
        double s = 0.0;
        for (int e = 0; e < n * n; ++e) s = __builtin_fma(w(X, e), w(X, e), s);
        s = sqrt(s);
        if (!(s > 0.0)) return 0.0;                                      // nilpotent
        lg = __builtin_fma(log(s), wgt, lg);
        wgt *= 0.5;
        const double inv = 1.0 / s;
        for (int e = 0; e < n * n; ++e) w(X, e) *= inv;
        mm(w, Y, X, false, X, false, n, n, n, false);
        const int t = X; X = Y; Y = t;
    }
    return exp(lg);
}

__device__ __forceinline__ double g_x(int power, int i, double eA, double fA)           // utils.py:78-95
{
    const double t = pow(eA + fA, (double)i) - pow(fA, (double)i);
    return power == 1 ? t : t * t;
}
__device__ __forceinline__ double g_u(int power, int i, double eA, double fA, double eB, double fB)   // utils.py:98-117
{
    const double t = (eB + fB) * g_x(1, i, eA, fA) + eB * pow(fA, (double)i);
    return power == 1 ? t : t * t;
}

}  // namespace

__global__ void __launch_bounds__(64) lqmpc_bounds_kernel(BoundsParams p)
{
    const long long b = p.b0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.b1) return;
    const int nx = p.nx, nu = p.nu, N = p.N, n = N * nu;
    const BoundsOff o = p.o;
    const Lane w{p.ws + (b - p.b0), p.stride};
    const double *sh = p.sh;
    const long long Bsz = p.Bsz;
    auto Ain = [&](int a, int c) { return p.A[(long long)(a * nx + c) * Bsz + b]; };
    auto Bin = [&](int a, int k) { return p.B[(long long)(a * nu + k) * Bsz + b]; };
    const double qmax = sh[p.osc + 0], qmin = sh[p.osc + 1], rmax = sh[p.osc + 2], rmin = sh[p.osc + 3];
    const double V_expert = sh[p.osc + 4], bar_u = sh[p.osc + 5], bar_du = sh[p.osc + 6];
    int status = 0;

    // ---------------- dlqr: P_inf by doubling ----------------
    for (int a = 0; a < nx; ++a)
        for (int c = 0; c < nx; ++c) {
            w(o.Ak, a * nx + c) = Ain(a, c);
            w(o.Hk, a * nx + c) = sh[p.oQ + a * nx + c];
            double t = 0.0;                                            // G0 = B R^-1 B'
            for (int k = 0; k < nu; ++k)
                for (int j = 0; j < nu; ++j) t = __builtin_fma(Bin(a, k) * sh[p.oRinv + k * nu + j], Bin(c, j), t);
            w(o.Gk, a * nx + c) = t;
        }
    symmetrise(w, o.Gk, nx);
    bool conv = false;
    for (int it = 0; it < 64 && !conv; ++it) {
        for (int e = 0; e < nx * nx; ++e) w(o.T1, e) = w(o.Hk, e);
        bool ok = spd_inverse(w, o.T1, nx);                             // T1 = H^-1
        for (int e = 0; e < nx * nx; ++e) w(o.T2, e) = w(o.T1, e) + w(o.Gk, e);
        symmetrise(w, o.T2, nx);
        ok = spd_inverse(w, o.T2, nx) && ok;                            // T2 = S = (H^-1 + G)^-1
        if (!ok) { status = 2; break; }
        mm(w, o.T3, o.T1, false, o.T2, false, nx, nx, nx, false);       // T3 = H^-1 S
        mm(w, o.T4, o.Ak, false, o.T3, false, nx, nx, nx, false);       // T4 = V = A H^-1 S = A (I + G H)^-1
        mm(w, o.T1, o.T4, false, o.Gk, false, nx, nx, nx, false);       // T1 = V G
        mm(w, o.Gk, o.T1, false, o.Ak, true, nx, nx, nx, true);         // G += V G A'
        symmetrise(w, o.Gk, nx);
        mm(w, o.T1, o.T2, false, o.Ak, false, nx, nx, nx, false);       // T1 = S A
        double dn = 0.0, hn = 0.0;
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nx; ++j) {
                double t = 0.0;
                for (int l = 0; l < nx; ++l) t = __builtin_fma(w(o.Ak, l * nx + i), w(o.T1, l * nx + j), t);   // (A' S A)_ij
                const double hnew = w(o.Hk, i * nx + j) + t;
                w(o.Hk, i * nx + j) = hnew;
                dn = __builtin_fma(t, t, dn); hn = __builtin_fma(hnew, hnew, hn);
            }
        symmetrise(w, o.Hk, nx);
        mm(w, o.T1, o.T4, false, o.Ak, false, nx, nx, nx, false);       // A+ = V A
        for (int e = 0; e < nx * nx; ++e) w(o.Ak, e) = w(o.T1, e);
        conv = dn <= 1e-34 * hn;                                         // |H+ - H|_F <= 1e-17 |H+|_F
        if (!(hn < 1e300)) { status = 2; break; }
    }
    if (!conv && status == 0) status = 1;
    // K = (R + B'PB)^-1 B'PA   (control.dlqr's gain, u = -K x)
    for (int k = 0; k < nu; ++k)
        for (int c = 0; c < nx; ++c) {                                  // U2 = B'P
            double t = 0.0;
            for (int a = 0; a < nx; ++a) t = __builtin_fma(Bin(a, k), w(o.Hk, a * nx + c), t);
            w(o.U2, k * nx + c) = t;
        }
    for (int k = 0; k < nu; ++k)
        for (int j = 0; j < nu; ++j) {
            double t = sh[p.oR + k * nu + j];
            for (int a = 0; a < nx; ++a) t = __builtin_fma(w(o.U2, k * nx + a), Bin(a, j), t);
            w(o.U1, k * nu + j) = t;
        }
    symmetrise(w, o.U1, nu);
    if (!spd_inverse(w, o.U1, nu)) status = 2;
    for (int k = 0; k < nu; ++k)
        for (int c = 0; c < nx; ++c) {                                  // T1 (nu x nx) = B'P A
            double t = 0.0;
            for (int a = 0; a < nx; ++a) t = __builtin_fma(w(o.U2, k * nx + a), Ain(a, c), t);
            w(o.T1, k * nx + c) = t;
        }
    for (int k = 0; k < nu; ++k)
        for (int c = 0; c < nx; ++c) {
            double t = 0.0;
            for (int j = 0; j < nu; ++j) t = __builtin_fma(w(o.U1, k * nu + j), w(o.T1, j * nx + c), t);
            w(o.K, k * nx + c) = t;
            if (p.K) p.K[(long long)(k * nx + c) * Bsz + b] = t;
        }
    if (p.Pinf)
        for (int e = 0; e < nx * nx; ++e) p.Pinf[(long long)e * Bsz + b] = w(o.Hk, e);

    // ---------------- local radius, stability numbers (the callers pass -K: utils_class.py:764, 843) ----------------
    double worst = 0.0;                                                  // max_i (F_u K)_i Q^-1 (F_u K)_i'
    for (int k = 0; k < nu; ++k) {
        double t = 0.0;
        for (int a = 0; a < nx; ++a)
            for (int c = 0; c < nx; ++c) t = __builtin_fma(w(o.K, k * nx + a) * sh[p.oQinv + a * nx + c], w(o.K, k * nx + c), t);
        const double ub = sh[p.oub + k], lb = sh[p.olb + k];
        worst = fmax(worst, fmax(t / (ub * ub), t / (lb * lb)));
    }
    const double eps = 1.0 / worst;
    if (p.eps) p.eps[b] = eps;
    const double nK = norm2(w, o.K, nu, nx, o.T1);
    for (int a = 0; a < nx; ++a)
        for (int c = 0; c < nx; ++c) {                                  // T1 = A - B K
            double t = Ain(a, c);
            for (int k = 0; k < nu; ++k) t = __builtin_fma(-Bin(a, k), w(o.K, k * nx + c), t);
            w(o.T1, a * nx + c) = t;
        }
    const double rho_cl = spectral_radius(w, o.T1, o.T2, nx);
    const double rho_K = (rho_cl + 0.4) * (rho_cl + 0.4);                // utils.py:358
    const double C_star = (1.0 + rmax * nK * nK / qmin) * fmax(1.0, qmax / qmin * 1.21);   // utils.py:364-368
    const double gamma = C_star / (1.0 - rho_K);
    const double rho_gamma = (gamma - 1.0) / gamma;

    // ---------------- |A|_2, |B|_2, the table M_d = A^d B, |Phi|_2, |Gamma|_2, lambda_min(hat H) ----------------
    for (int a = 0; a < nx; ++a) {
        for (int c = 0; c < nx; ++c) w(o.T1, a * nx + c) = Ain(a, c);
        for (int k = 0; k < nu; ++k) w(o.Md, a * nu + k) = Bin(a, k);
    }
    const double fA = norm2(w, o.T1, nx, nx, o.T2);
    const double fB = norm2(w, o.Md, nx, nu, o.T2);
    for (int d = 1; d < N; ++d)
        for (int a = 0; a < nx; ++a)
            for (int k = 0; k < nu; ++k) {
                double t = 0.0;
                for (int c = 0; c < nx; ++c) t = __builtin_fma(Ain(a, c), w(o.Md, ((d - 1) * nx + c) * nu + k), t);
                w(o.Md, (d * nx + a) * nu + k) = t;
            }
    // Phi'Phi = sum_{k=0..N} (A^k)'A^k: T1 = current power, T3 = the sum
    for (int a = 0; a < nx; ++a)
        for (int c = 0; c < nx; ++c) { w(o.T1, a * nx + c) = (a == c) ? 1.0 : 0.0; w(o.T3, a * nx + c) = (a == c) ? 1.0 : 0.0; }
    for (int k = 1; k <= N; ++k) {
        for (int a = 0; a < nx; ++a)
            for (int c = 0; c < nx; ++c) {
                double t = 0.0;
                for (int l = 0; l < nx; ++l) t = __builtin_fma(Ain(a, l), w(o.T1, l * nx + c), t);
                w(o.T2, a * nx + c) = t;
            }
        for (int e = 0; e < nx * nx; ++e) w(o.T1, e) = w(o.T2, e);
        mm(w, o.T3, o.T1, true, o.T1, false, nx, nx, nx, true);
    }
    double emax, emin;
    sym_eig_extremes(w, o.T3, nx, emax, emin);
    const double nPhi = sqrt(fmax(emax, 0.0));
    // Gamma (the reference's, with a leading zero block row): block (r, c) = M_{r-1-c} for r > c, r = 0..N, c = 0..N-1
    auto gam = [&](int rho, int i) -> double {
        const int r = rho / nx, a = rho % nx, c = i / nu, k = i % nu;
        return r > c ? w(o.Md, ((r - 1 - c) * nx + a) * nu + k) : 0.0;
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const int c1 = i / nu, k1 = i % nu, c2 = j / nu, k2 = j % nu;
            double t = 0.0;
            for (int r = c1 + 1; r <= N; ++r)                         // c1 >= c2
                for (int a = 0; a < nx; ++a)
                    t = __builtin_fma(w(o.Md, ((r - 1 - c1) * nx + a) * nu + k1), w(o.Md, ((r - 1 - c2) * nx + a) * nu + k2), t);
            w(o.E, i * n + j) = t; w(o.E, j * n + i) = t;
        }
    sym_eig_extremes_tridiag(w, o.E, n, o.V, emax, emin);
    const double nG = sqrt(fmax(emax, 0.0));
    // hat H = kron(R, I_N) + Gamma' kron(Q, I_{N+1}) Gamma with the index pairing of utils.py:316-319
    const int rows = (N + 1) * nx;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = (i % N == j % N) ? sh[p.oR + (i / N) * nu + (j / N)] : 0.0;
            for (int r1 = 0; r1 < rows; ++r1) {
                const double gi = gam(r1, i);
                if (gi == 0.0) continue;
                const int qa = r1 / (N + 1), s = r1 % (N + 1);
                double u = 0.0;
                for (int c = 0; c < nx; ++c) {
                    const double qv = sh[p.oQ + qa * nx + c];
                    if (qv != 0.0) u = __builtin_fma(qv, gam(c * (N + 1) + s, j), u);
                }
                t = __builtin_fma(gi, u, t);
            }
            w(o.E, i * n + j) = t;
        }
    for (int i = 0; i < n; ++i)                                        // the product above is symmetric when Q is; mirror the lower triangle
        for (int j = 0; j < i; ++j) w(o.E, j * n + i) = w(o.E, i * n + j);
    sym_eig_extremes_tridiag(w, o.E, n, o.V, emax, emin);
    const double min_H = emin;

    // ---------------- energy_decreasing: xi, eta (utils_class.py:344-373) ----------------
    const double eA = p.eA[b], eB = p.eB[b];
    const double MV = p.MV ? p.MV[b] : 0.0;
    const double L_V = fmax(gamma, MV / eps);                            // utils.py:575
    const double N_0 = ceil(fmax(0.0, MV / eps - gamma));                // utils.py:576
    const double G_A = (fA == 1.0) ? (double)(N - 1) : (1.0 - pow(fA, 2.0 * (N - 1))) / (1.0 - fA * fA);   // utils.py:393-409
    const double term = 1.0 + fA * fA * qmax / qmin;                     // utils.py:503
    const double fA2N = pow(fA, 2.0 * N - 2.0);
    const double omega_1 = qmax * (term * fA2N + G_A);                   // utils.py:510
    const double rg = pow(rho_gamma, (double)N - N_0);
    const double decay = qmax * fA2N * gamma * rg;
    const double omega_05 = sqrt(qmax * (L_V - 1.0) * G_A) + 0.5 * term * sqrt(decay);   // utils.py:514
    const double eta = (term - 1.0) * gamma * rg;                        // utils.py:517
    const double hh = eA * eA / qmin + eB * eB / rmin;                   // utils.py:538
    const double xi = hh * omega_1 + 2.0 * sqrt(hh) * omega_05;

    // ---------------- energy_bound: alpha, beta (utils_class.py:308-342) ----------------
    double nx2 = 0.0;
    for (int a = 0; a < nx; ++a) nx2 = __builtin_fma(sh[p.ox + a], sh[p.ox + a], nx2);
    double s_in = 0.0, s_out = 0.0;
    for (int i = 0; i <= N; ++i) {                                       // utils.py:296-302
        s_out += (s_in + g_x(2, i, eA, fA)) * (nx2 + i * bar_u);
        s_in += g_u(2, i, eA, fA, eB, fB);
    }
    const double E_psi = qmax * s_out;
    double bar_gx = 0.0, bar_gu = 0.0, run = 0.0;
    for (int i = 0; i < N; ++i) {                                        // utils.py:186-223
        bar_gx += g_x(1, i + 1, eA, fA);
        run += g_u(1, i, eA, fA, eB, fB);
        bar_gu += run;
    }
    const double theta_u = qmax * (2.0 * nG * bar_gu + bar_gu * bar_gu);                       // utils.py:253-255
    const double theta_xu = qmax * (nG * bar_gx + nPhi * bar_gu + bar_gx * bar_gu);            // utils.py:258-262
    const double bar_theta = sqrt(N * bar_u) * theta_u + sqrt(nx2) * theta_xu;                 // utils.py:313
    const double mn = fmin(sqrt(N * bar_du), bar_theta / min_H);
    const double E_u = rmax * mn * mn;                                                         // utils.py:325
    const double E_psi_u = qmax / rmax * (nG + bar_gu) * (nG + bar_gu) * E_u;                  // utils.py:331
    const double p0 = sh[p.op + 0], p1 = sh[p.op + 1], p2 = sh[p.op + 2];
    const double sp = sqrt(E_psi), su = sqrt(E_u), spu = sqrt(E_psi_u);
    const double alpha = fmax(p0 * sp + p2 * spu + p0 * sp * p2 * spu, p1 * su);               // utils_class.py:332-335
    const double beta = (1.0 + p0 * sp) * (spu / p2 + E_psi_u) + su / p1 + E_u + sp / p0 + E_psi;   // utils_class.py:338-340
    if (!(fabs(alpha) < 1e300) || !(fabs(beta) < 1e300) || !(fabs(xi) < 1e300) || !(fabs(eta) < 1e300)) status = status ? status : 3;

    if (p.alpha) p.alpha[b] = alpha;
    if (p.beta) p.beta[b] = beta;
    if (p.xi) p.xi[b] = xi;
    if (p.eta) p.eta[b] = eta;
    if (p.bound) p.bound[b] = (alpha * V_expert + beta) / (1.0 - xi - eta);                    // utils_class.py:858-859
    if (p.aux) {                                                                               // diagnostics: the norms behind the numbers
        double *a = p.aux;
        a[0 * Bsz + b] = gamma; a[1 * Bsz + b] = rho_cl; a[2 * Bsz + b] = fA; a[3 * Bsz + b] = fB;
        a[4 * Bsz + b] = nG; a[5 * Bsz + b] = nPhi; a[6 * Bsz + b] = min_H; a[7 * Bsz + b] = nK;
    }
    if (p.status) p.status[b] = status;
}

// ---------------- the on-chip kernels (lqmpc_bounds_chip.h), prebuilt for the reference's shapes and the headline shape ----------------
template <int NX, int NU, int N>
__global__ void __launch_bounds__(64) lqmpc_bounds_small_kernel(BoundsParams p) { bounds_small<NX, NU, N>(p); }

template <int NX, int NU, int N, int LPI>
__global__ void __launch_bounds__(64) lqmpc_bounds_big_kernel(BoundsParams p)
{
    __shared__ double lds_raw[BigT<NX, NU, N, LPI>::IPW * BigT<NX, NU, N, LPI>::INST];
    bounds_big<NX, NU, N, LPI>(p, lds_raw);
}

template <int NX, int NU, int N>
static void launch_chip_one(const BoundsParams &p, hipStream_t stream)
{
    constexpr int LPI = (N * NU <= 32) ? 16 : 64, IPW = 64 / LPI;
    const long long m = p.b1 - p.b0;
    hipLaunchKernelGGL((lqmpc_bounds_small_kernel<NX, NU, N>), dim3((unsigned)((m + 3) / 4)), dim3(64), 0, stream, p);
    hipLaunchKernelGGL((lqmpc_bounds_big_kernel<NX, NU, N, LPI>), dim3((unsigned)((m + IPW - 1) / IPW)), dim3(64), 0, stream, p);
}

bool launch_bounds_chip(const BoundsParams &p, hipStream_t stream)
{
#define LQMPC_BC(X_, U_, H_) if (p.nx == X_ && p.nu == U_ && p.N == H_) { launch_chip_one<X_, U_, H_>(p, stream); return true; }
    LQMPC_BC(2, 1, 6) LQMPC_BC(2, 1, 7) LQMPC_BC(2, 1, 8) LQMPC_BC(2, 1, 9) LQMPC_BC(2, 1, 10)      // the reference's horizon sweep
    LQMPC_BC(4, 2, 10)                                                                          // C3
#undef LQMPC_BC
    return false;
}

void launch_bounds(const BoundsParams &p, hipStream_t stream)
{
    const long long m = p.b1 - p.b0;
    hipLaunchKernelGGL(lqmpc_bounds_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, stream, p);
}

}  // namespace lqmpc
