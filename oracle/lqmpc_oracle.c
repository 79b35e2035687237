/*
 * lqmpc_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, fp64 restatement of the reference's hot path
 *     LQ_MPC_Controller.solve      /root/reference/utils_class.py:48-91
 *     LQ_MPC_Simulator.simulate    /root/reference/utils_class.py:245-285
 *     the M_V / J_T batch loops    /root/reference/utils_class.py:802-833, 886-916
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported baseline.  The shipped HIP
 * path (lq_mpc_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * Third-party arithmetic: the reference hands the QP to cvxpy's default QP
 * back-end (utils_class.py:84-88; cvxpy is un-vendored and un-pinned, absent
 * from this image).  The reference's committed outputs
 * (data_lq_mpc_multipleSys.npz <- error_{A,B}_f.npy, utils_class.py:944-958)
 * are the exact QP optimum to ~1e-12, so this oracle solves the box QP EXACTLY
 * with a primal active-set method (a different algorithm from the HIP path's
 * interior-point method, which makes the parity check independent).
 * Parity is PINNED: tests/test_oracle_golden.py reproduces the npz tables
 * true_cost_error, true_cost_horizon and V_expert from the .npy inputs.
 *
 * Problem statement followed (utils_class.py:59-91), n = N*nu, NO 1/2 factor:
 *   x_{i+1} = A x_i + B u_i                                     (line 64)
 *   cost = sum_{i=0}^{N-2} |x_{i+1} - xref_i|^2_Q               (lines 67-69)
 *        + |x_N - xref_{N-1}|^2_P                               (lines 70-72)
 *        + sum_{i=0}^{N-1} |u_i - uref_i|^2_R                   (line 75)
 *   s.t. F_u u_i <= 1  (box rows only: lb <= u_i <= ub)         (line 81)
 *   returns u_0 = u*[:,0],  V_N = cost* + x0' Q x0              (line 91)
 * Condensed:  X = Phi x0 + Gamma U,  Phi = [A; ...; A^N],
 *   Gamma_{r,c} = A^{r-c} B (r >= c)   (the reference's own Gamma, with an
 *   extra zero block-row, is utils.py:145-174),
 *   H = Gamma' Qbar Gamma + Rbar,  g = Gamma' Qbar (Phi x0 - Xref) - Rbar Uref,
 *   c = |Phi x0 - Xref|^2_Qbar + |Uref|^2_Rbar,  cost(U) = U'HU + 2 g'U + c.
 *
 * All matrices are row-major doubles.  U is time-major: U[i*nu + k] = u_i[k].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LQO_MAXN 256

/* ---------- small dense helpers ---------- */
static void matmul(int m, int k, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int t = 0; t < k; ++t) s += A[i * k + t] * B[t * n + j];
            C[i * n + j] = s;
        }
}

/* in-place lower Cholesky of the leading m x m of a matrix with row stride ld; 0 on success */
static int chol_lower(int m, double *K, int ld)
{
    for (int j = 0; j < m; ++j) {
        double d = K[j * ld + j];
        for (int k = 0; k < j; ++k) d -= K[j * ld + k] * K[j * ld + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        K[j * ld + j] = d;
        for (int i = j + 1; i < m; ++i) {
            double s = K[i * ld + j];
            for (int k = 0; k < j; ++k) s -= K[i * ld + k] * K[j * ld + k];
            K[i * ld + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(int m, const double *L, int ld, double *b)
{
    for (int i = 0; i < m; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * ld + k] * b[k];
        b[i] = s / L[i * ld + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s -= L[k * ld + i] * b[k];
        b[i] = s / L[i * ld + i];
    }
}

/* ---------- condensing (utils_class.py:62-75 restated in matrix form) ---------- */
/* H: n x n, F: n x nx with g = F x0 - gref.  M_k = A^k B. */
int lqo_condense(int nx, int nu, int N, const double *A, const double *B,
                 const double *Q, const double *R, const double *P,
                 double *H, double *F)
{
    const int n = N * nu;
    if (n > LQO_MAXN || nx > 64) return -2;
    double *M = (double *)malloc(sizeof(double) * (size_t)N * nx * nu);      /* M_k */
    double *Ap = (double *)malloc(sizeof(double) * (size_t)(N + 1) * nx * nx); /* A^k */
    double *W = (double *)malloc(sizeof(double) * (size_t)nx * (nu > nx ? nu : nx));
    if (!M || !Ap || !W) { free(M); free(Ap); free(W); return -3; }
    memcpy(M, B, sizeof(double) * nx * nu);
    for (int k = 1; k < N; ++k) matmul(nx, nx, nu, A, M + (size_t)(k - 1) * nx * nu, M + (size_t)k * nx * nu);
    memset(Ap, 0, sizeof(double) * nx * nx);
    for (int i = 0; i < nx; ++i) Ap[i * nx + i] = 1.0;
    for (int k = 1; k <= N; ++k) matmul(nx, nx, nx, A, Ap + (size_t)(k - 1) * nx * nx, Ap + (size_t)k * nx * nx);

    memset(H, 0, sizeof(double) * n * n);
    memset(F, 0, sizeof(double) * n * nx);
    for (int r = 0; r < N; ++r) {                 /* block row r of Gamma predicts x_{r+1} */
        const double *Qr = (r < N - 1) ? Q : P;   /* terminal weight on the last one (lines 67-72) */
        for (int bi = 0; bi <= r; ++bi) {
            /* W = Qr * M_{r-bi}  (nx x nu) */
            matmul(nx, nx, nu, Qr, M + (size_t)(r - bi) * nx * nu, W);
            for (int bj = 0; bj <= r; ++bj) {
                const double *Mj = M + (size_t)(r - bj) * nx * nu;
                for (int ui = 0; ui < nu; ++ui)
                    for (int uj = 0; uj < nu; ++uj) {
                        double s = 0.0;
                        for (int x = 0; x < nx; ++x) s += W[x * nu + ui] * Mj[x * nu + uj];
                        H[(bi * nu + ui) * n + (bj * nu + uj)] += s;
                    }
            }
            /* F block row bi += M_{r-bi}' Qr A^{r+1} = W' A^{r+1} */
            const double *Ar = Ap + (size_t)(r + 1) * nx * nx;
            for (int ui = 0; ui < nu; ++ui)
                for (int y = 0; y < nx; ++y) {
                    double s = 0.0;
                    for (int x = 0; x < nx; ++x) s += W[x * nu + ui] * Ar[x * nx + y];
                    F[(bi * nu + ui) * nx + y] += s;
                }
        }
    }
    for (int i = 0; i < N; ++i)
        for (int a = 0; a < nu; ++a)
            for (int b = 0; b < nu; ++b) H[(i * nu + a) * n + (i * nu + b)] += R[a * nu + b];
    free(M); free(Ap); free(W);
    return 0;
}

/* Linear term g and constant c for a given x0 and (nullable) references.
 * x_ref is (nx, N) row-major as in the reference (column i pairs with x_{i+1}, line 69),
 * u_ref is (nu, N).  g = Gamma' Qbar (Phi x0 - Xref) - Rbar Uref. */
int lqo_linear_terms(int nx, int nu, int N, const double *A, const double *B,
                     const double *Q, const double *R, const double *P,
                     const double *x0, const double *x_ref, const double *u_ref,
                     double *g, double *c_out)
{
    const int n = N * nu;
    if (n > LQO_MAXN || nx > 64) return -2;
    double d[64], lam[64], t[64];
    double *D = (double *)malloc(sizeof(double) * (size_t)N * nx);
    if (!D) return -3;
    double c = 0.0;
    /* free response minus reference, d_r = A^{r+1} x0 - xref_r */
    memcpy(d, x0, sizeof(double) * nx);
    for (int r = 0; r < N; ++r) {
        matmul(nx, nx, 1, A, d, t);
        memcpy(d, t, sizeof(double) * nx);
        const double *Qr = (r < N - 1) ? Q : P;
        for (int x = 0; x < nx; ++x) D[r * nx + x] = d[x] - (x_ref ? x_ref[x * N + r] : 0.0);
        matmul(nx, nx, 1, Qr, D + r * nx, t);
        for (int x = 0; x < nx; ++x) c += D[r * nx + x] * t[x];
    }
    /* costate recursion: lam_r = Qr d_r + A' lam_{r+1};  g_i = B' lam_i - R uref_i */
    memset(lam, 0, sizeof(lam));
    for (int r = N - 1; r >= 0; --r) {
        const double *Qr = (r < N - 1) ? Q : P;
        matmul(nx, nx, 1, Qr, D + r * nx, t);
        double nl[64];
        for (int x = 0; x < nx; ++x) {
            double s = t[x];
            for (int y = 0; y < nx; ++y) s += A[y * nx + x] * lam[y];
            nl[x] = s;
        }
        memcpy(lam, nl, sizeof(double) * nx);
        for (int k = 0; k < nu; ++k) {
            double s = 0.0;
            for (int x = 0; x < nx; ++x) s += B[x * nu + k] * lam[x];
            if (u_ref)
                for (int j = 0; j < nu; ++j) s -= R[k * nu + j] * u_ref[j * N + r];
            g[r * nu + k] = s;
        }
    }
    if (u_ref)
        for (int r = 0; r < N; ++r)
            for (int a = 0; a < nu; ++a)
                for (int b = 0; b < nu; ++b) c += u_ref[a * N + r] * R[a * nu + b] * u_ref[b * N + r];
    *c_out = c;
    free(D);
    return 0;
}

/* ---------- exact box QP:  min u'Hu + 2 g'u,  lb <= u <= ub  (primal active set) ---------- */
/* state: 0 free, -1 at lower, +1 at upper.  Returns iterations (>0) or <0 on failure. */
int lqo_boxqp(int n, const double *H, const double *g, const double *lb, const double *ub, double *u)
{
    if (n > LQO_MAXN) return -2;
    int st[LQO_MAXN], idx[LQO_MAXN];
    double *K = (double *)malloc(sizeof(double) * (size_t)n * n);
    double rhs[LQO_MAXN];
    if (!K) return -3;
    double scale = 0.0;
    for (int i = 0; i < n; ++i) { double a = fabs(g[i]); if (a > scale) scale = a; }
    for (int i = 0; i < n * n; ++i) { double a = fabs(H[i]); if (a > scale) scale = a; }
    if (scale == 0.0) scale = 1.0;

    /* start: unconstrained minimiser, clipped; clipped coordinates form the working set */
    memcpy(K, H, sizeof(double) * (size_t)n * n);
    if (chol_lower(n, K, n)) { free(K); return -4; }
    for (int i = 0; i < n; ++i) u[i] = -g[i];
    chol_solve(n, K, n, u);
    for (int i = 0; i < n; ++i) {
        st[i] = 0;
        if (u[i] <= lb[i]) { u[i] = lb[i]; st[i] = -1; }
        else if (u[i] >= ub[i]) { u[i] = ub[i]; st[i] = 1; }
    }
    int it, maxit = 20 * n + 50, rc = -5;
    for (it = 1; it <= maxit; ++it) {
        int m = 0;
        for (int i = 0; i < n; ++i) if (st[i] == 0) idx[m++] = i;
        double pmax = 0.0;
        if (m > 0) {
            for (int a = 0; a < m; ++a) {
                int i = idx[a];
                double s = -g[i];
                for (int j = 0; j < n; ++j) if (st[j] != 0) s -= H[i * n + j] * u[j];
                rhs[a] = s;
                for (int b = 0; b <= a; ++b) K[a * m + b] = H[i * n + idx[b]];
            }
            if (chol_lower(m, K, m)) { rc = -4; break; }
            chol_solve(m, K, m, rhs);            /* rhs = minimiser on the free face */
            double alpha = 1.0; int blk = -1, blk_side = 0;
            for (int a = 0; a < m; ++a) {
                int i = idx[a];
                double p = rhs[a] - u[i];
                if (fabs(p) > pmax) pmax = fabs(p);
                if (p < 0.0 && rhs[a] < lb[i]) { double t = (lb[i] - u[i]) / p; if (t < alpha) { alpha = t; blk = i; blk_side = -1; } }
                if (p > 0.0 && rhs[a] > ub[i]) { double t = (ub[i] - u[i]) / p; if (t < alpha) { alpha = t; blk = i; blk_side = 1; } }
            }
            if (blk >= 0) {
                for (int a = 0; a < m; ++a) { int i = idx[a]; u[i] += alpha * (rhs[a] - u[i]); }
                u[blk] = blk_side < 0 ? lb[blk] : ub[blk];
                st[blk] = blk_side;
                continue;
            }
            for (int a = 0; a < m; ++a) u[idx[a]] = rhs[a];   /* full step to the face minimiser */
        }
        /* multipliers on the working set: grad = H u + g; need grad >= 0 at lb, <= 0 at ub */
        int worst = -1; double wv = 1e-13 * scale;
        for (int i = 0; i < n; ++i) {
            double s = g[i];
            for (int j = 0; j < n; ++j) s += H[i * n + j] * u[j];
            if (st[i] == -1 && -s > wv) { wv = -s; worst = i; }
            if (st[i] == 1 && s > wv) { wv = s; worst = i; }
        }
        if (worst < 0) { rc = it; break; }
        st[worst] = 0;
    }
    free(K);
    return rc;
}

/* ---------- one open-loop solve (utils_class.py:48-91) ---------- */
/* lb, ub: (nu,) box on every u_i.  U_out (n) may be NULL.  Returns active-set iterations or <0. */
int lqo_solve(int nx, int nu, int N, const double *A, const double *B,
              const double *Q, const double *R, const double *P,
              const double *lb, const double *ub,
              const double *x0, const double *x_ref, const double *u_ref,
              double *u0_out, double *VN_out, double *U_out)
{
    const int n = N * nu;
    if (n > LQO_MAXN) return -2;
    double *H = (double *)malloc(sizeof(double) * ((size_t)n * n + (size_t)n * nx));
    if (!H) return -3;
    double *F = H + (size_t)n * n;
    double g[LQO_MAXN], U[LQO_MAXN], LB[LQO_MAXN], UB[LQO_MAXN], c;
    int rc = lqo_condense(nx, nu, N, A, B, Q, R, P, H, F);
    if (!rc) rc = lqo_linear_terms(nx, nu, N, A, B, Q, R, P, x0, x_ref, u_ref, g, &c);
    if (rc) { free(H); return rc; }
    for (int i = 0; i < N; ++i) for (int k = 0; k < nu; ++k) { LB[i * nu + k] = lb[k]; UB[i * nu + k] = ub[k]; }
    rc = lqo_boxqp(n, H, g, LB, UB, U);
    if (rc > 0) {
        double cost = c;
        for (int i = 0; i < n; ++i) {
            double s = 2.0 * g[i];
            for (int j = 0; j < n; ++j) s += H[i * n + j] * U[j];
            cost += U[i] * s;
        }
        double q0 = 0.0;
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) q0 += x0[a] * Q[a * nx + b] * x0[b];
        *VN_out = cost + q0;                                  /* line 91 */
        for (int k = 0; k < nu; ++k) u0_out[k] = U[k];
        if (U_out) memcpy(U_out, U, sizeof(double) * n);
    }
    free(H);
    return rc;
}

/* ---------- closed-loop rollout (utils_class.py:245-285) ---------- */
/* Model (A,B) is condensed once (H is constant over the rollout); the plant is (A_true,B_true).
 * X_out: (nx, T+1) row-major, U_out: (nu, T) row-major, as the reference returns them; may be NULL. */
int lqo_simulate(int T, int nx, int nu, int N, const double *A, const double *B,
                 const double *Q, const double *R, const double *P,
                 const double *lb, const double *ub, const double *x0,
                 const double *A_true, const double *B_true,
                 const double *x_ref, const double *u_ref,
                 double *JT_out, double *X_out, double *U_out)
{
    const int n = N * nu;
    if (n > LQO_MAXN || nx > 64) return -2;
    double *H = (double *)malloc(sizeof(double) * ((size_t)n * n + (size_t)n * nx));
    if (!H) return -3;
    double *F = H + (size_t)n * n;
    double g[LQO_MAXN], gref[LQO_MAXN], U[LQO_MAXN], LB[LQO_MAXN], UB[LQO_MAXN], x[64], xn[64], zero[64], c;
    int rc = lqo_condense(nx, nu, N, A, B, Q, R, P, H, F);
    if (rc) { free(H); return rc; }
    /* gref = g(x0 = 0) carries the reference terms; g(x) = F x + gref */
    memset(zero, 0, sizeof(zero));
    rc = lqo_linear_terms(nx, nu, N, A, B, Q, R, P, zero, x_ref, u_ref, gref, &c);
    if (rc) { free(H); return rc; }
    for (int i = 0; i < N; ++i) for (int k = 0; k < nu; ++k) { LB[i * nu + k] = lb[k]; UB[i * nu + k] = ub[k]; }
    memcpy(x, x0, sizeof(double) * nx);
    double cost = 0.0;                                         /* line 261 */
    for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) cost += x[a] * Q[a * nx + b] * x[b];
    if (X_out) for (int a = 0; a < nx; ++a) X_out[a * (T + 1)] = x[a];
    int total_it = 0;
    for (int t = 0; t < T; ++t) {                              /* lines 266-283 */
        for (int i = 0; i < n; ++i) {
            double s = gref[i];
            for (int a = 0; a < nx; ++a) s += F[i * nx + a] * x[a];
            g[i] = s;
        }
        rc = lqo_boxqp(n, H, g, LB, UB, U);
        if (rc < 0) { free(H); return rc; }
        total_it += rc;
        for (int a = 0; a < nx; ++a) {                         /* line 277: plant step */
            double s = 0.0;
            for (int b = 0; b < nx; ++b) s += A_true[a * nx + b] * x[b];
            for (int k = 0; k < nu; ++k) s += B_true[a * nu + k] * U[k];
            xn[a] = s;
        }
        memcpy(x, xn, sizeof(double) * nx);
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) cost += x[a] * Q[a * nx + b] * x[b];   /* 282 */
        for (int a = 0; a < nu; ++a) for (int b = 0; b < nu; ++b) cost += U[a] * R[a * nu + b] * U[b];   /* 283 */
        if (X_out) for (int a = 0; a < nx; ++a) X_out[a * (T + 1) + t + 1] = x[a];
        if (U_out) for (int k = 0; k < nu; ++k) U_out[k * T + t] = U[k];
    }
    *JT_out = cost;
    free(H);
    return total_it > 0 ? total_it : 1;
}

/* ---------- batched entry points (same SoA layout as include/lqmpc.h) ----------
 * Per-instance arrays are instance-minor: A[(r*nx+c)*Bsz + b], B[(r*nu+k)*Bsz + b], x0[a*Bsz + b].
 * Q, R, P, lb, ub and the (nullable) references are shared by the batch.
 * Used by tests (parity) and by bench.py's cpu_baseline leg (threads = OpenMP threads). */
static void gather_inst(int rows, int cols, const double *S, long Bsz, long b, double *out)
{
    for (int e = 0; e < rows * cols; ++e) out[e] = S[(long)e * Bsz + b];
}

int lqo_solve_batch(long Bsz, int nx, int nu, int N, const double *A, const double *B,
                    const double *Q, const double *R, const double *P,
                    const double *lb, const double *ub, const double *x0,
                    const double *x_ref, const double *u_ref,
                    double *u0, double *VN, int *iters, int threads)
{
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad)
#endif
    for (long b = 0; b < Bsz; ++b) {
        double Ab[64 * 64], Bb[64 * 64], xb[64], u0b[64], v;
        gather_inst(nx, nx, A, Bsz, b, Ab);
        gather_inst(nx, nu, B, Bsz, b, Bb);
        gather_inst(nx, 1, x0, Bsz, b, xb);
        int rc = lqo_solve(nx, nu, N, Ab, Bb, Q, R, P, lb, ub, xb, x_ref, u_ref, u0b, &v, NULL);
        if (rc < 0) { bad += 1; v = NAN; for (int k = 0; k < nu; ++k) u0b[k] = NAN; }
        for (int k = 0; k < nu; ++k) u0[(long)k * Bsz + b] = u0b[k];
        VN[b] = v;
        if (iters) iters[b] = rc;
    }
    return bad ? -1 : 0;
}

/* A_true/B_true: shared (true_shared != 0: plain (nx,nx)/(nx,nu)) or per-instance SoA.
 * X (nullable): [(a*(T+1)+t)*Bsz + b];  U (nullable): [(k*T+t)*Bsz + b]. */
int lqo_rollout_batch(long Bsz, int T, int nx, int nu, int N, const double *A, const double *B,
                      const double *Q, const double *R, const double *P,
                      const double *lb, const double *ub, const double *x0,
                      const double *A_true, const double *B_true, int true_shared,
                      const double *x_ref, const double *u_ref,
                      double *JT, double *X, double *U, int threads)
{
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
#endif
    for (long b = 0; b < Bsz; ++b) {
        double Ab[64 * 64], Bb[64 * 64], At[64 * 64], Bt[64 * 64], xb[64], j;
        double *Xb = X ? (double *)malloc(sizeof(double) * nx * (T + 1)) : NULL;
        double *Ub = U ? (double *)malloc(sizeof(double) * nu * T) : NULL;
        gather_inst(nx, nx, A, Bsz, b, Ab);
        gather_inst(nx, nu, B, Bsz, b, Bb);
        gather_inst(nx, 1, x0, Bsz, b, xb);
        if (true_shared) { memcpy(At, A_true, sizeof(double) * nx * nx); memcpy(Bt, B_true, sizeof(double) * nx * nu); }
        else { gather_inst(nx, nx, A_true, Bsz, b, At); gather_inst(nx, nu, B_true, Bsz, b, Bt); }
        int rc = lqo_simulate(T, nx, nu, N, Ab, Bb, Q, R, P, lb, ub, xb, At, Bt, x_ref, u_ref, &j, Xb, Ub);
        if (rc < 0) { bad += 1; j = NAN; }
        JT[b] = j;
        if (Xb) { for (int e = 0; e < nx * (T + 1); ++e) X[(long)e * Bsz + b] = Xb[e]; free(Xb); }
        if (Ub) { for (int e = 0; e < nu * T; ++e) U[(long)e * Bsz + b] = Ub[e]; free(Ub); }
    }
    return bad ? -1 : 0;
}

/* M_V[b] = max_k V_N(x0s[:,k]) over K shared initial states (utils_class.py:816-824, 899-907).
 * x0s: (nx, K) row-major, shared by the batch (the reference's x0_vec). */
int lqo_max_vn_batch(long Bsz, int K, int nx, int nu, int N, const double *A, const double *B,
                     const double *Q, const double *R, const double *P,
                     const double *lb, const double *ub, const double *x0s,
                     const double *x_ref, const double *u_ref, double *MV, int threads)
{
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
#endif
    for (long b = 0; b < Bsz; ++b) {
        double Ab[64 * 64], Bb[64 * 64], xb[64], u0b[64], v, best = -INFINITY;
        gather_inst(nx, nx, A, Bsz, b, Ab);
        gather_inst(nx, nu, B, Bsz, b, Bb);
        for (int k = 0; k < K; ++k) {
            for (int a = 0; a < nx; ++a) xb[a] = x0s[a * K + k];
            int rc = lqo_solve(nx, nu, N, Ab, Bb, Q, R, P, lb, ub, xb, x_ref, u_ref, u0b, &v, NULL);
            if (rc < 0) { bad += 1; v = NAN; }
            if (v > best || v != v) best = v;
        }
        MV[b] = best;
    }
    return bad ? -1 : 0;
}

int lqo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
