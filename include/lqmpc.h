/*
 * lqmpc.h -- C ABI of the MI355X batched LQ-MPC box-QP solver (liblqmpc_hip.so).
 *
 * The reference (lcrekko/lq_mpc) has no FFI: its operator boundary for this path is two
 * Python classes.  Each entry point below names the reference interface it replaces:
 *
 *   lqmpc_solve_batch    <- LQ_MPC_Controller.solve          /root/reference/utils_class.py:48-91
 *                           (one call per instance there; Bsz instances per call here)
 *   lqmpc_rollout_batch  <- LQ_MPC_Simulator.simulate        /root/reference/utils_class.py:245-285
 *   lqmpc_sweep_batch    <- one (error level | horizon) of data_generation: both of the following in one call
 *   lqmpc_bounds_batch   <- the rest of one pass of data_generation: control.dlqr + LQ_RDP_Calculator.energy_decreasing +
 *                           energy_bound + the bound               /root/reference/utils_class.py:837-859, 920-942
 *                                                                  (utils_class.py:308-373; utils.py:78-117, 186-584)
 *   lqmpc_max_vn_batch   <- the M_V loops of
 *                           LQ_RDP_Behavior_Multiple.data_generation
 *                                                            /root/reference/utils_class.py:813-824, 896-907
 *                           and LQ_RDP_Behavior.OL_energy_bound  utils_class.py:439-466
 *
 * Problem solved per instance (utils_class.py:59-91; n = N*nu; no 1/2 factor):
 *   min  sum_{i=0}^{N-2} |x_{i+1}-xref_i|^2_Q + |x_N-xref_{N-1}|^2_P + sum_{i=0}^{N-1} |u_i-uref_i|^2_R
 *   s.t. x_{i+1} = A x_i + B u_i,   lb <= u_i <= ub     (F_u u_i <= 1 with box rows, line 81)
 *   u_0 = u*[:,0],  V_N = cost* + x0' Q x0               (line 91)
 *
 * Data layout.  Per-instance arrays are instance-minor ("SoA", the layout of the reference's
 * error_{A,B}_f.npy files, utils_class.py:749-750):
 *     A[(r*nx + c)*Bsz + b]   B[(r*nu + k)*Bsz + b]   x0[a*Bsz + b]
 *     u0[k*Bsz + b]  VN[b]  JT[b]  MV[b]  status[b]  iters[b]
 *     X[(a*(T+1) + t)*Bsz + b]   U[(k*T + t)*Bsz + b]            (optional trajectories)
 * Shared by the whole batch (plain row-major, always HOST pointers, copied by the call):
 *     Q (nx,nx)  R (nu,nu)  P (nx,nx)  lb (nu)  ub (nu)
 *     x_ref (nx,N) / u_ref (nu,N) as the reference passes them; NULL means zeros
 *     A_true (nx,nx), B_true (nx,nu) when true_per_instance == 0
 *     x0s (nx,K): the K initial states of lqmpc_max_vn_batch
 * All floating point is IEEE fp64, as in the reference.
 *
 * Two flavours of every batched call:
 *   lqmpc_*_batch      per-instance pointers are HOST memory; the call copies in, runs, copies
 *                      out and returns when the results are in the caller's buffers.
 *   lqmpc_*_batch_dev  per-instance pointers are DEVICE memory on the handle's GPU; the call
 *                      enqueues on the handle's stream and returns (use lqmpc_sync).
 *
 * Errors: every call returns 0 on success or a negative lqmpc_error; nothing is thrown
 * across the ABI.  lqmpc_last_error() gives a thread-local message.  Per instance,
 * status[b] is 0 converged, 1 iteration cap reached, 2 non-finite data; iters[b] counts the KKT-system
 * factorisations spent on instance b (interior-point iterations + active-set iterations).  The QP is always
 * feasible and strictly convex (R > 0, non-empty box), so "infeasible" cannot occur.
 *
 * Threading: a handle is bound to one device and one stream and is not thread-safe;
 * distinct handles may be used from distinct threads.
 */
#ifndef LQMPC_H
#define LQMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lqmpc_handle lqmpc_handle;

enum lqmpc_error {
    LQMPC_OK = 0,
    LQMPC_ERR_BAD_ARG = -1,      /* NULL pointer, non-positive size, empty box, dims over the limits */
    LQMPC_ERR_HIP = -2,          /* a HIP runtime call failed (message in lqmpc_last_error) */
    LQMPC_ERR_NO_DEVICE = -3,    /* no usable GPU */
    LQMPC_ERR_ALLOC = -4,        /* device or host allocation failed */
    LQMPC_ERR_UNSUPPORTED = -5   /* requested kernel variant not built for these dims */
};

/* Which kernel family runs the batch. */
enum lqmpc_kernel {
    LQMPC_KERNEL_AUTO = 0,        /* register-resident specialisation when built for (nx,nu,N); else the workgroup
                                     kernel when it applies; else generic */
    LQMPC_KERNEL_GENERIC = 1,     /* any dims up to the limits; one instance per lane, workspace in HBM */
    LQMPC_KERNEL_SPECIALIZED = 2, /* fail with LQMPC_ERR_UNSUPPORTED if no specialisation exists */
    LQMPC_KERNEL_WORKGROUP = 3    /* one instance per 256-thread workgroup, matrices in LDS (32 < N*nu <= 128, nx <= 16,
                                     nu <= LQMPC_MAX_NU, LDS image within 160 KiB); LQMPC_ERR_UNSUPPORTED otherwise */
};

/* ABI note (library 0.2.0): the struct starts with its own size.  lqmpc_default_options / lqmpc_get_options fill it in;
 * lqmpc_set_options rejects a struct whose struct_size is not the library's sizeof(lqmpc_options) -- a caller built against another
 * version of this header gets LQMPC_ERR_BAD_ARG instead of the library reading past its struct.  Always start from
 * lqmpc_default_options() or lqmpc_get_options(). */
typedef struct lqmpc_options {
    uint32_t struct_size; /* sizeof(lqmpc_options) of the header the caller was built against */
    uint32_t reserved;    /* 0 */
    double eps;        /* relative tolerance on complementarity gap and dual residual (default 1e-12) */
    double tau;        /* fraction-to-the-boundary of the interior-point step (default 0.999) */
    double z0_scale;   /* initial multipliers = z0_scale * |q|_inf (default 0.1) */
    int32_t max_iter;  /* interior-point iteration cap per QP; also caps the <= 8 active-set warm-start iterations of the packed
                          kernel (default 50) */
    int32_t polish;    /* 1: finish with an exact solve on the identified active set (default 1) */
    int32_t kernel;    /* enum lqmpc_kernel (default AUTO) */
    int32_t presolve;  /* unconstrained-minimiser shortcut: the minimiser v = G x + v_r (G = -P^-1 Fq, built once
                          per instance) is tested against the box before any iteration; a QP whose minimiser is
                          interior is finished there, exactly.  -1 auto (= on), 0 off, 1 on.  Specialised kernels only; the generic kernel ignores it.  (default -1) */
    int32_t order;     /* processing order of a rollout batch.  1: a probe launch computes a difficulty key per instance
                          (largest stage gradient of the free response over the horizon, in units of what one input
                          can counter), the batch is bucket-sorted by its logarithm and the rollout walks it hardest-first, so the
                          instances that share a wavefront leave the constrained regime together; results are written
                          back to their original positions and do not depend on the order.  0: natural order.
                          -1 auto (1 for specialised rollouts with presolve, T >= 4 and a batch large enough to pay for the probe: 8192 instances in the 16-lane-row layout, 1024 in the packed one).  (default -1) */
    int32_t warm_start; /* primal-dual active-set warm start: before the interior-point loop, the QP is solved exactly on
                          the face guessed from the unconstrained minimiser (rows outside the box sit on their bound)
                          and the guess is corrected from the KKT signs, up to 8 times; a fixed point is the exact
                          optimum, otherwise the interior-point loop runs.  -1 auto (= presolve), 0 off, 1 on.
                          Specialised kernels only.  (default -1) */
    /* ---- layout / tuning selectors (the tests force every path through these; defaults are the measured best) ---- */
    int32_t layout;     /* which specialised family serves a shape that has both: -1 auto (lqmpc_api.hip: use_r16), 0 the
                          packed register-resident kernel (and its two-tier launch for sorted rollouts), 1 the 16-lane-row
                          kernel.  (default -1) */
    int32_t r16_maxit;  /* active-set iterations the 16-lane-row kernel spends on one QP before it hands the instance back to
                          the packed kernel's interior-point path (second launch over a device-side list); 0 hands every
                          constrained QP back.  In [0, 64].  (default 12) */
    int32_t r16_build;  /* build of the 16-lane-row kernel: -1 auto (the one-wave latency build up to 4 096 instances, the
                          two-waves throughput build above), 0 throughput build, 1 latency build.  (default -1) */
    int32_t nwide;      /* packed family, sorted rollouts: how many of the hardest instances get the 16-lane-row layout in
                          the two-tier launch; -1 auto (min(Bsz/8, 4096)), 0 none.  (default -1) */
    int32_t jit;        /* shapes (nx, nu, N) without a prebuilt instantiation: compile the 16-lane-row kernel for them at run time
                          (hiprtc; nx <= 8, nu <= 4, N*nu <= 48; about two seconds per shape and entry point on first use, then
                          cached) instead of falling back to the generic kernel.  -1 auto (= on), 0 off, 1 on.  (default -1) */
    int32_t reserved2;  /* 0 */
} lqmpc_options;

/* Limits of this build. */
#define LQMPC_MAX_NX 16
#define LQMPC_MAX_NU 8
#define LQMPC_MAX_N  64
#define LQMPC_MAX_NVAR 128   /* n = N*nu */

const char *lqmpc_version(void);
const char *lqmpc_last_error(void);
/* Number of GPUs the HIP runtime sees (0 if none; never fails). */
int lqmpc_device_count(void);

/* Handle life cycle.  lqmpc_create makes its own stream; lqmpc_create_on_stream borrows a
 * hipStream_t (passed as void*, may be NULL for the default stream). */
int lqmpc_create(int device, lqmpc_handle **out);
int lqmpc_create_on_stream(int device, void *hip_stream, lqmpc_handle **out);
int lqmpc_destroy(lqmpc_handle *h);
int lqmpc_sync(lqmpc_handle *h);

void lqmpc_default_options(lqmpc_options *opt);
int lqmpc_set_options(lqmpc_handle *h, const lqmpc_options *opt);
int lqmpc_get_options(const lqmpc_handle *h, lqmpc_options *opt);

/* Run-time compiled kernels (options.jit).  lqmpc_jit_cache_dir: directory where code objects are kept across processes (NULL or
 * "" = in memory only; process-wide).  lqmpc_jit_compile: compile (or find in that directory) every kernel of one shape now -- it needs
 * no GPU, so a build machine can fill the cache; returns the number of code objects available (5) or a negative lqmpc_error, with the
 * compiler's message in `log` if given. */
int lqmpc_jit_cache_dir(const char *dir);
int lqmpc_jit_compile(int nx, int nu, int N, char *log, int log_len);
/* ... and the two on-chip kernels of lqmpc_bounds_batch for a shape (returns 2). */
int lqmpc_jit_compile_bounds(int nx, int nu, int N, char *log, int log_len);

/* 1 if a register-resident specialisation for (nx,nu,N) is compiled in, else 0. */
int lqmpc_has_specialization(int nx, int nu, int N);
/* Name of the kernel the last batched call on this handle launched (for logs/profiles). */
const char *lqmpc_last_kernel(const lqmpc_handle *h);

/* Pre-size the handle's device scratch so later calls do no allocation (e.g. before timing). */
int lqmpc_reserve(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T);

/* ---- LQ_MPC_Controller.solve, batched (utils_class.py:48-91) ---- */
int lqmpc_solve_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz,
                      const double *A, const double *B,
                      const double *Q, const double *R, const double *P,
                      const double *lb, const double *ub,
                      const double *x0, const double *x_ref, const double *u_ref,
                      double *u0, double *VN, int32_t *status, int32_t *iters);
int lqmpc_solve_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz,
                          const double *dA, const double *dB,
                          const double *Q, const double *R, const double *P,
                          const double *lb, const double *ub,
                          const double *dx0, const double *x_ref, const double *u_ref,
                          double *du0, double *dVN, int32_t *dstatus, int32_t *diters);

/* ---- LQ_MPC_Simulator.simulate, batched (utils_class.py:245-285) ----
 * The model (A,B) drives the QP, the plant (A_true,B_true) drives the state.
 * true_per_instance == 0: A_true/B_true are single HOST matrices shared by the batch
 * (the reference's usage, utils_class.py:830-832); != 0: SoA arrays like A/B (host for the
 * host flavour, device for the _dev flavour).  X, U, status, iters may be NULL. */
int lqmpc_rollout_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T,
                        const double *A, const double *B,
                        const double *Q, const double *R, const double *P,
                        const double *lb, const double *ub, const double *x0,
                        const double *A_true, const double *B_true, int true_per_instance,
                        const double *x_ref, const double *u_ref,
                        double *JT, double *X, double *U, int32_t *status, int32_t *iters);
int lqmpc_rollout_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T,
                            const double *dA, const double *dB,
                            const double *Q, const double *R, const double *P,
                            const double *lb, const double *ub, const double *dx0,
                            const double *A_true, const double *B_true, int true_per_instance,
                            const double *x_ref, const double *u_ref,
                            double *dJT, double *dX, double *dU, int32_t *dstatus, int32_t *diters);

/* ---- M_V = max_k V_N(x0s[:,k]) per instance (utils_class.py:816-824, 899-907) ---- */
int lqmpc_max_vn_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int K,
                       const double *A, const double *B,
                       const double *Q, const double *R, const double *P,
                       const double *lb, const double *ub, const double *x0s,
                       const double *x_ref, const double *u_ref,
                       double *MV, int32_t *status, int32_t *iters);
int lqmpc_max_vn_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int K,
                           const double *dA, const double *dB,
                           const double *Q, const double *R, const double *P,
                           const double *lb, const double *ub, const double *x0s,
                           const double *x_ref, const double *u_ref,
                           double *dMV, int32_t *dstatus, int32_t *diters);

/* ---- one horizon of the reference's sweep in one call: M_V over the K states x0s AND the closed-loop cost J_T from x0,
 *      for the same (A,B,N) per instance (LQ_RDP_Behavior_Multiple.data_generation, utils_class.py:813-833, 896-916).
 *      One kernel launch (condensing once per instance) where the 16-lane-row layout serves the shape and the batch has
 *      at most 16 384 instances; otherwise lqmpc_max_vn_batch_dev followed by lqmpc_rollout_batch_dev on the stream.
 *      status = the worse of the two parts, iters = their sum. ---- */
int lqmpc_sweep_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, int K,
                      const double *A, const double *B,
                      const double *Q, const double *R, const double *P,
                      const double *lb, const double *ub, const double *x0, const double *x0s,
                      const double *A_true, const double *B_true, int true_per_instance,
                      const double *x_ref, const double *u_ref,
                      double *JT, double *MV, int32_t *status, int32_t *iters);
int lqmpc_sweep_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, int K,
                          const double *dA, const double *dB,
                          const double *Q, const double *R, const double *P,
                          const double *lb, const double *ub, const double *dx0, const double *x0s,
                          const double *A_true, const double *B_true, int true_per_instance,
                          const double *x_ref, const double *u_ref,
                          double *dJT, double *dMV, int32_t *dstatus, int32_t *diters);

/* ---- the coefficients of the reference's performance bound, per system (utils_class.py:837-859, 920-942) ----
 * For every model (A, B) of the batch: K = control.dlqr(A, B, Q, R) (utils_class.py:840); with the gain -K, the error levels
 * e_A[b], e_B[b] and the energy bar M_V[b] (lqmpc_max_vn_batch's output; NULL = 0): xi, eta of
 * LQ_RDP_Calculator.energy_decreasing (utils_class.py:344-373) and eps = local_radius (utils.py:548-564); with the state x
 * (n_x, shared) and the parameter triple p (3, positive): alpha, beta of energy_bound (utils_class.py:308-342); and
 * bound = (alpha V_expert + beta) / (1 - xi - eta) (utils_class.py:858-859).  The box (lb, ub) stands for the rows of F_u
 * (e_k / ub_k, e_k / lb_k; bounds must be non-zero); bar_u and bar_d_u (two Gurobi QPs in utils.py:592-650) are closed forms for a box.
 * Outputs are per instance, any may be NULL: K[(k*nx + a)*Bsz + b] (dlqr's sign: u = -K x), alpha, beta, xi, eta, bound, eps [b],
 * aux[j*Bsz + b] for j < 8 = gamma, rho(A - BK), |A|_2, |B|_2, |Gamma|_2, |Phi|_2, lambda_min(hat H), |K|_2 (diagnostics),
 * status[b] = 0 ok, 1 the Riccati doubling did not settle in 64 steps, 2 not stabilisable / non-finite data (no gain), 3 the gain
 * and eps are valid but the bound's formulas leave the reals (gamma < 0 when rho(A - BK) + 0.4 > 1, utils.py:358-371).
 * Q, R (symmetric positive definite), lb, ub, x, p are HOST pointers in both flavours. */
int lqmpc_bounds_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz,
                       const double *A, const double *B, const double *Q, const double *R,
                       const double *lb, const double *ub,
                       const double *e_A, const double *e_B, const double *MV,
                       const double *x, const double *p, double V_expert,
                       double *K, double *alpha, double *beta, double *xi, double *eta, double *bound, double *eps,
                       double *aux, int32_t *status);
int lqmpc_bounds_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz,
                           const double *dA, const double *dB, const double *Q, const double *R,
                           const double *lb, const double *ub,
                           const double *de_A, const double *de_B, const double *dMV,
                           const double *x, const double *p, double V_expert,
                           double *dK, double *dalpha, double *dbeta, double *dxi, double *deta, double *dbound, double *deps,
                           double *daux, int32_t *dstatus);

/* ---- timing on the handle's stream (hipEvents), for bench.py's roofline ----
 * begin/end bracket any number of *_dev calls; end waits for the stream and returns the
 * elapsed milliseconds between the two events. */
int lqmpc_timer_begin(lqmpc_handle *h);
int lqmpc_timer_end(lqmpc_handle *h, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* LQMPC_H */
