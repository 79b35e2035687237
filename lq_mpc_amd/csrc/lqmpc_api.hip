// lqmpc_api.hip -- host side of the C ABI declared in include/lqmpc.h: handle, argument checks,
// shared-block packing, workspace management, kernel dispatch, host<->device staging.
#include "lqmpc_common.h"
#include "lqmpc_bounds.h"
#include "../../include/lqmpc.h"


#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace lqmpc {
long long generic_ws_entries(int nx, int nu, int N);
void launch_generic(const KParams &p, hipStream_t stream);
// lqmpc_spec.hip: returns false when no specialisation is built for (nx,nu,N)
bool spec_available(int nx, int nu, int N);
bool launch_spec(const KParams &p, hipStream_t stream, const char **name);
void launch_order_scatter(const KParams &p, int *perm, hipStream_t stream);
bool spec_tiered_available(int nx, int nu, int N);
// lqmpc_r16.hip: rollouts with one instance per 16-lane row (n <= 32)
bool r16_available(int nx, int nu, int N);
int r16_lanes(int nx, int nu, int N);       // 16, 64 (one instance per wavefront: n > 32) or 0
bool launch_r16(const KParams &p, hipStream_t stream, const char **name);
// lqmpc_wg.hip: one instance per workgroup, 32 < n <= 128
bool wg_supported(const KParams &p, const double *lb, const double *ub);
bool launch_wg(const KParams &p, hipStream_t stream, const char **name);
// lqmpc_jit.hip: the 16-lane-row kernel (and the probe) of a shape without a prebuilt instantiation, compiled at run time
bool jit_r16_shape(int nx, int nu, int N, int *lpi);
bool jit_available(int device, int nx, int nu, int N, int mode, std::string *why);
bool launch_jit(int device, const KParams &p, hipStream_t stream, const char **name, std::string *why);
bool launch_jit_bounds(int device, const BoundsParams &p, hipStream_t stream, std::string *why);
// lqmpc_generic.hip: the generic kernel over a device-side list (p.perm, p.count_dev) with `cols` workspace columns
void launch_generic_list(const KParams &p, int cols, hipStream_t stream);
}  // namespace lqmpc

using lqmpc::KParams;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(LQMPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct HostBuf {                     // pinned host memory (hipHostMalloc) + the event of the last copy that read / wrote it
    void *p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool ev_valid = false;
};

struct lqmpc_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    lqmpc_options opt;
    DevBuf shared, ws;
    DevBuf key, perm, rec;           // difficulty ordering of rollout batches (its counters live in `fail`)
    DevBuf fail;                     // [count (2) | counters of the difficulty order | list] of the instances handed to the packed kernel
    bool fail_cleared = false;       // build_order zeroed count and counters in this call already
    bool hist_ready = false;         // both sets of order counters are zero / being zeroed by the probe launches: no fill needed
    int hist_turn = 0;               // which set the next probe counts into
    DevBuf st2, it2;                 // lqmpc_sweep_batch_dev without a fused kernel: status / iters of the max-V_N pass
    DevBuf stage[12];                // host-flavour staging (inputs and outputs)
    DevBuf arena;                    // host-flavour staging of small calls: one packed block, one copy each way
    HostBuf arena_pin;
    HostBuf shared_pin[2];           // source of the shared block's upload
    int shared_turn = 0;
    std::vector<double> shared_host; // last uploaded shared block
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const char *last_kernel = "none";
    bool use_wg = false;             // set by prepare(): this call runs on the workgroup kernel
    bool use_jit = false;            // set by prepare(): this call runs on a run-time compiled 16-lane-row kernel (lqmpc_jit.hip)
};

constexpr int JIT_FALLBACK_COLS = 1024;   // workspace columns of the generic kernel when it only serves a hand-back list

constexpr size_t HIST_INTS = (size_t)lqmpc::ORDER_CELLS * lqmpc::ORDER_PAD;
constexpr size_t FAIL_HDR = 16 + 2 * HIST_INTS;    // ints in front of the hand-back list: its count, the order's two alternating sets of counters

static int ensure_host(HostBuf &b, size_t bytes)
{
    if (!b.ev) {
        hipError_t e = hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
        if (e != hipSuccess) { b.ev = nullptr; return fail(LQMPC_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    }
    if (bytes <= b.cap) return 0;
    if (b.p) {
        if (b.ev_valid) (void)hipEventSynchronize(b.ev);
        (void)hipHostFree(b.p);
        b.p = nullptr; b.cap = 0; b.ev_valid = false;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) { b.p = nullptr; return fail(LQMPC_ERR_ALLOC, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    b.cap = want;
    return 0;
}

static int ensure(lqmpc_handle *h, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return fail(LQMPC_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    b.cap = want;
    return 0;
}

// The buffer that holds the hand-back count, the order's counters and the hand-back list.  A re-allocated buffer holds undefined
// counters (hipMalloc may even hand the old address back), so the no-fill path of build_order is switched off by CAPACITY.
static int ensure_fail(lqmpc_handle *h, size_t bytes)
{
    const size_t cap_before = h->fail.cap;
    const int rc = ensure(h, h->fail, bytes);
    if (h->fail.cap != cap_before || rc) h->hist_ready = false;
    return rc;
}

extern "C" {

const char *lqmpc_version(void) { return "lqmpc-mi355x 0.2.0 (gfx950, fp64)"; }
const char *lqmpc_last_error(void) { return g_err.c_str(); }

int lqmpc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void lqmpc_default_options(lqmpc_options *opt)
{
    if (!opt) return;
    opt->struct_size = (uint32_t)sizeof(lqmpc_options);
    opt->reserved = 0;
    opt->eps = 1e-12;
    opt->tau = 0.999;
    opt->z0_scale = 0.1;
    opt->max_iter = 50;
    opt->polish = 1;
    opt->kernel = LQMPC_KERNEL_AUTO;
    opt->presolve = -1;
    opt->order = -1;
    opt->warm_start = -1;
    opt->layout = -1;
    opt->r16_maxit = 12;
    opt->r16_build = -1;
    opt->nwide = -1;
    opt->jit = -1;
    opt->reserved2 = 0;
}

int lqmpc_create_on_stream(int device, void *hip_stream, lqmpc_handle **out)
{
    if (!out) return fail(LQMPC_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(LQMPC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(LQMPC_ERR_BAD_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    lqmpc_handle *h = new lqmpc_handle();
    h->device = device;
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    lqmpc_default_options(&h->opt);
    hipError_t e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { delete h; return fail(LQMPC_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    *out = h;
    return 0;
}

int lqmpc_create(int device, lqmpc_handle **out)
{
    int rc = lqmpc_create_on_stream(device, nullptr, out);
    if (rc) return rc;
    hipStream_t s;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { lqmpc_destroy(*out); *out = nullptr; return fail(LQMPC_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    (*out)->stream = s;
    (*out)->own_stream = true;
    return 0;
}

int lqmpc_destroy(lqmpc_handle *h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->shared.p) (void)hipFree(h->shared.p);
    if (h->ws.p) (void)hipFree(h->ws.p);
    for (DevBuf *b : {&h->key, &h->perm, &h->rec, &h->fail, &h->st2, &h->it2}) if (b->p) (void)hipFree(b->p);
    for (auto &b : h->stage) if (b.p) (void)hipFree(b.p);
    if (h->arena.p) (void)hipFree(h->arena.p);
    for (HostBuf *b : {&h->arena_pin, &h->shared_pin[0], &h->shared_pin[1]}) {
        if (b->p) (void)hipHostFree(b->p);
        if (b->ev) (void)hipEventDestroy(b->ev);
    }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int lqmpc_sync(lqmpc_handle *h)
{
    if (!h) return fail(LQMPC_ERR_BAD_ARG, "handle is NULL");
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int lqmpc_set_options(lqmpc_handle *h, const lqmpc_options *opt)
{
    if (!h || !opt) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    if (opt->struct_size != sizeof(lqmpc_options))
        return fail(LQMPC_ERR_BAD_ARG, "lqmpc_options.struct_size does not match this library: built against another lqmpc.h? "
                                       "(start from lqmpc_default_options / lqmpc_get_options)");
    if (!(opt->eps > 0.0 && opt->eps < 1.0)) return fail(LQMPC_ERR_BAD_ARG, "eps must be in (0,1)");
    if (!(opt->tau > 0.0 && opt->tau < 1.0)) return fail(LQMPC_ERR_BAD_ARG, "tau must be in (0,1)");
    if (!(opt->z0_scale > 0.0)) return fail(LQMPC_ERR_BAD_ARG, "z0_scale must be positive");
    if (opt->max_iter < 1 || opt->max_iter > 1000) return fail(LQMPC_ERR_BAD_ARG, "max_iter must be in [1,1000]");
    if (opt->kernel < LQMPC_KERNEL_AUTO || opt->kernel > LQMPC_KERNEL_WORKGROUP) return fail(LQMPC_ERR_BAD_ARG, "unknown kernel selector");
    if (opt->presolve < -1 || opt->presolve > 1) return fail(LQMPC_ERR_BAD_ARG, "presolve must be -1, 0 or 1");
    if (opt->order < -1 || opt->order > 1) return fail(LQMPC_ERR_BAD_ARG, "order must be -1, 0 or 1");
    if (opt->warm_start < -1 || opt->warm_start > 1) return fail(LQMPC_ERR_BAD_ARG, "warm_start must be -1, 0 or 1");
    if (opt->layout < -1 || opt->layout > 1) return fail(LQMPC_ERR_BAD_ARG, "layout must be -1, 0 or 1");
    if (opt->r16_maxit < 0 || opt->r16_maxit > 64) return fail(LQMPC_ERR_BAD_ARG, "r16_maxit must be in [0,64]");
    if (opt->r16_build < -1 || opt->r16_build > 1) return fail(LQMPC_ERR_BAD_ARG, "r16_build must be -1, 0 or 1");
    if (opt->nwide < -1) return fail(LQMPC_ERR_BAD_ARG, "nwide must be -1 (auto) or a count");
    if (opt->jit < -1 || opt->jit > 1) return fail(LQMPC_ERR_BAD_ARG, "jit must be -1, 0 or 1");
    h->opt = *opt;
    return 0;
}

int lqmpc_get_options(const lqmpc_handle *h, lqmpc_options *opt)
{
    if (!h || !opt) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    *opt = h->opt;
    return 0;
}

int lqmpc_has_specialization(int nx, int nu, int N) { return lqmpc::spec_available(nx, nu, N) ? 1 : 0; }

const char *lqmpc_last_kernel(const lqmpc_handle *h) { return h ? h->last_kernel : "none"; }

}  // extern "C"

// ------------------------------------------------------------------------------------------
static int check_dims(int nx, int nu, int N, int64_t Bsz)
{
    if (nx < 1 || nu < 1 || N < 1 || Bsz < 1) return fail(LQMPC_ERR_BAD_ARG, "nx, nu, N and Bsz must be positive");
    if (nx > LQMPC_MAX_NX || nu > LQMPC_MAX_NU || N > LQMPC_MAX_N || N * nu > LQMPC_MAX_NVAR) {
        char buf[160];
        snprintf(buf, sizeof buf, "dims over the build limits (nx<=%d, nu<=%d, N<=%d, N*nu<=%d)", LQMPC_MAX_NX, LQMPC_MAX_NU,
                 LQMPC_MAX_N, LQMPC_MAX_NVAR);
        return fail(LQMPC_ERR_BAD_ARG, buf);
    }
    return 0;
}

// small host-side dense helpers for the batch-shared weights (lqmpc_bounds_batch): SPD inverse, extreme eigenvalues
static bool host_spd_inverse(std::vector<double> &M, int n)
{
    for (int k = 0; k < n; ++k) {
        const double d = M[k * n + k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double p = 1.0 / d;
        M[k * n + k] = 1.0;
        for (int j = 0; j < n; ++j) M[k * n + j] *= p;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = M[i * n + k];
            M[i * n + k] = 0.0;
            for (int j = 0; j < n; ++j) M[i * n + j] -= f * M[k * n + j];
        }
    }
    return true;
}

static bool host_sym_eig_extremes(const double *Min, int n, double &emax, double &emin)
{
    std::vector<double> M(Min, Min + n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (std::fabs(M[i * n + j] - M[j * n + i]) > 1e-12 * (std::fabs(M[i * n + i]) + std::fabs(M[j * n + j]))) return false;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < n; ++p) {
            dg += M[p * n + p] * M[p * n + p];
            for (int q = p + 1; q < n; ++q) off += M[p * n + q] * M[p * n + q];
        }
        if (!(off > 1e-32 * (dg + off))) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = M[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (M[q * n + q] - M[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = M[k * n + p], akq = M[k * n + q];
                    M[k * n + p] = c * akp - s * akq; M[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = M[p * n + k], aqk = M[q * n + k];
                    M[p * n + k] = c * apk - s * aqk; M[q * n + k] = s * apk + c * aqk;
                }
            }
    }
    emax = -1e308; emin = 1e308;
    for (int p = 0; p < n; ++p) { emax = std::max(emax, M[p * n + p]); emin = std::min(emin, M[p * n + p]); }
    return std::isfinite(emax) && std::isfinite(emin);
}

// First gain K_0 of the N-stage problem on (A, B): S_N = P, K_j = (R + B'S B)^-1 B'S A, S <- Q + A'S (A - B K_j).  Row-major, nu x nx.
// (Host, once per call of a rollout with a shared plant: the difficulty order's roll uses it; 10 x O(nx^3) flops.)
static bool host_first_gain(int nx, int nu, int N, const double *A, const double *B, const double *Q, const double *R, const double *P, double *K)
{
    std::vector<double> S(P, P + nx * nx), SA(nx * nx), SB(nx * nu), Re(nu * nu), F(nu * nx), Acl(nx * nx), T(nx * nx);
    for (int j = 0; j < N; ++j) {
        for (int a = 0; a < nx; ++a) {
            for (int c = 0; c < nx; ++c) { double t = 0.0; for (int k = 0; k < nx; ++k) t += S[a * nx + k] * A[k * nx + c]; SA[a * nx + c] = t; }
            for (int c = 0; c < nu; ++c) { double t = 0.0; for (int k = 0; k < nx; ++k) t += S[a * nx + k] * B[k * nu + c]; SB[a * nu + c] = t; }
        }
        for (int r = 0; r < nu; ++r) {
            for (int c = 0; c < nu; ++c) { double t = R[r * nu + c]; for (int k = 0; k < nx; ++k) t += B[k * nu + r] * SB[k * nu + c]; Re[r * nu + c] = t; }
            for (int c = 0; c < nx; ++c) { double t = 0.0; for (int k = 0; k < nx; ++k) t += B[k * nu + r] * SA[k * nx + c]; F[r * nx + c] = t; }
        }
        // K = Re^-1 F by Gaussian elimination with partial pivoting (nu <= 8)
        for (int col = 0; col < nu; ++col) {
            int piv = col;
            for (int r = col + 1; r < nu; ++r) if (std::fabs(Re[r * nu + col]) > std::fabs(Re[piv * nu + col])) piv = r;
            if (!(std::fabs(Re[piv * nu + col]) > 0.0)) return false;
            if (piv != col) {
                for (int c = 0; c < nu; ++c) std::swap(Re[piv * nu + c], Re[col * nu + c]);
                for (int c = 0; c < nx; ++c) std::swap(F[piv * nx + c], F[col * nx + c]);
            }
            const double inv = 1.0 / Re[col * nu + col];
            for (int r = 0; r < nu; ++r) {
                if (r == col) continue;
                const double f = Re[r * nu + col] * inv;
                for (int c = 0; c < nu; ++c) Re[r * nu + c] -= f * Re[col * nu + c];
                for (int c = 0; c < nx; ++c) F[r * nx + c] -= f * F[col * nx + c];
            }
        }
        for (int r = 0; r < nu; ++r) { const double inv = 1.0 / Re[r * nu + r]; for (int c = 0; c < nx; ++c) K[r * nx + c] = F[r * nx + c] * inv; }
        for (int a = 0; a < nx; ++a)
            for (int c = 0; c < nx; ++c) { double t = A[a * nx + c]; for (int k = 0; k < nu; ++k) t -= B[a * nu + k] * K[k * nx + c]; Acl[a * nx + c] = t; }
        for (int a = 0; a < nx; ++a)
            for (int c = 0; c < nx; ++c) { double t = 0.0; for (int k = 0; k < nx; ++k) t += S[a * nx + k] * Acl[k * nx + c]; T[a * nx + c] = t; }
        for (int a = 0; a < nx; ++a)
            for (int c = 0; c < nx; ++c) { double t = Q[a * nx + c]; for (int k = 0; k < nx; ++k) t += A[k * nx + a] * T[k * nx + c]; S[a * nx + c] = t; }
    }
    for (int e = 0; e < nu * nx; ++e) if (!std::isfinite(K[e])) return false;
    return true;
}

static bool use_spec(const lqmpc_handle *h, int nx, int nu, int N)
{
    return h->opt.kernel != LQMPC_KERNEL_GENERIC && h->opt.kernel != LQMPC_KERNEL_WORKGROUP && lqmpc::spec_available(nx, nu, N);
}

struct Call {
    int nx, nu, N, T, K, mode, true_per_instance;
    int64_t Bsz;
    const double *Q, *R, *P, *lb, *ub, *x_ref, *u_ref, *At_sh, *Bt_sh, *x0s;
};

// Pack the batch-shared data, upload it (skipped when identical to the last upload), size the
// workspace, fill the kernel parameter block.
static int prepare(lqmpc_handle *h, const Call &c, KParams &p)
{
    h->fail_cleared = false;
    int rc = check_dims(c.nx, c.nu, c.N, c.Bsz);
    if (rc) return rc;
    if (!c.Q || !c.R || !c.P || !c.lb || !c.ub) return fail(LQMPC_ERR_BAD_ARG, "Q, R, P, lb, ub must not be NULL");
    for (int k = 0; k < c.nu; ++k)
        if (!(c.ub[k] > c.lb[k]) || !std::isfinite(c.lb[k]) || !std::isfinite(c.ub[k]))
            return fail(LQMPC_ERR_BAD_ARG, "every input needs a finite box with lb < ub");
    HIP_TRY(hipSetDevice(h->device));
    const int nx = c.nx, nu = c.nu, N = c.N;
    std::vector<double> sh;
    auto put = [&](const double *src, int count) {
        int off = (int)sh.size();
        sh.resize(sh.size() + count, 0.0);
        if (src) memcpy(sh.data() + off, src, sizeof(double) * count);
        return off;
    };
    memset(&p, 0, sizeof p);
    p.so.Q = put(c.Q, nx * nx);
    p.so.R = put(c.R, nu * nu);
    p.so.P = put(c.P, nx * nx);
    p.so.lb = put(c.lb, nu);
    p.so.ub = put(c.ub, nu);
    p.so.xref = put(c.x_ref, nx * N);
    p.so.uref = put(c.u_ref, nu * N);
    p.so.At = put(c.true_per_instance ? nullptr : c.At_sh, nx * nx);
    p.so.Bt = put(c.true_per_instance ? nullptr : c.Bt_sh, nx * nu);
    p.so.x0s = put(c.x0s, c.x0s ? nx * c.K : 1);
    // the difficulty order's key for rollouts on a shared plant with zero references and a centred box (lqmpc_probe.h): a clipped roll
    // of the PLANT under the first gain of the N-stage problem on it -- the closed loop every instance of the batch runs in
    bool lin = c.x_ref || c.u_ref;
    for (int k = 0; k < nu; ++k) lin = lin || (c.ub[k] + c.lb[k] != 0.0);
    // (only where an order can be built at all: small batches skip the host's Riccati steps)
    const bool may_order = h->opt.order > 0 || (h->opt.order < 0 && c.T >= 4 && c.Bsz >= 1024);
    bool roll = may_order && !lin && !c.true_per_instance && c.At_sh && c.Bt_sh && nu <= 8 && (c.mode == lqmpc::MODE_ROLLOUT || c.mode == lqmpc::MODE_SWEEP);
    std::vector<double> Kg((size_t)nu * nx, 0.0);
    if (roll) roll = host_first_gain(nx, nu, N, c.At_sh, c.Bt_sh, c.Q, c.R, c.P, Kg.data());
    p.so.Kg = put(roll ? Kg.data() : nullptr, nu * nx);
    p.order_roll = roll ? 1 : 0;
    rc = ensure(h, h->shared, sh.size() * sizeof(double));
    if (rc) return rc;
    if (sh != h->shared_host) {
        // The device copy may still be read by an earlier launch on this stream: the copy is ordered behind it by the stream.  No
        // host-side wait: the source is pinned memory owned by the handle (two buffers taken in turn, so that the copy of this call
        // cannot be overwritten by the next call's packing while it is still in flight).
        const size_t bytes = sh.size() * sizeof(double);
        HostBuf &hb = h->shared_pin[h->shared_turn ^= 1];
        int rc2 = ensure_host(hb, bytes);
        if (rc2) return rc2;
        if (hb.ev_valid) HIP_TRY(hipEventSynchronize(hb.ev));       // (two calls back: long done, costs nothing)
        memcpy(hb.p, sh.data(), bytes);
        HIP_TRY(hipMemcpyAsync(h->shared.p, hb.p, bytes, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(hb.ev, h->stream));
        hb.ev_valid = true;
        h->shared_host.swap(sh);
    }
    p.nx = nx; p.nu = nu; p.N = N; p.n = N * nu;
    p.T = c.T; p.K = c.K; p.mode = c.mode;
    p.true_per_instance = c.true_per_instance;
    p.has_ref = (c.x_ref || c.u_ref) ? 1 : 0;
    p.has_lin = p.has_ref;
    for (int k = 0; k < nu; ++k) p.has_lin |= (c.ub[k] + c.lb[k] != 0.0) ? 1 : 0;
    p.max_iter = h->opt.max_iter; p.polish = h->opt.polish;
    p.presolve = h->opt.presolve < 0 ? 1 : h->opt.presolve;
    p.warm_start = h->opt.warm_start < 0 ? p.presolve : h->opt.warm_start;
    p.eps = h->opt.eps; p.tau = h->opt.tau; p.z0_scale = h->opt.z0_scale;
    p.Bsz = c.Bsz;
    p.sh = (const double *)h->shared.p;
    p.r16_maxit = h->opt.r16_maxit;                            // a small cap exercises the hand-back path (tests)
    p.r16_build = h->opt.r16_build;
    if (h->opt.kernel == LQMPC_KERNEL_SPECIALIZED && !lqmpc::spec_available(nx, nu, N))
        return fail(LQMPC_ERR_UNSUPPORTED, "no register-resident specialisation built for these dims");
    // a shape without a prebuilt instantiation inside the 16-lane-row domain: compile its kernel now (first use: ~2 s; then cached).
    // The algorithm is the presolve + warm-started active set, so the option combinations that switch those off stay on the
    // generic / workgroup kernels, as do the shapes whose compile fails (no hiprtc on the machine: reported by last_error once).
    h->use_jit = false;
    if (h->opt.kernel == LQMPC_KERNEL_AUTO && h->opt.jit != 0 && !lqmpc::spec_available(nx, nu, N) && p.presolve && p.warm_start &&
        c.Bsz <= INT32_MAX && lqmpc::jit_r16_shape(nx, nu, N, nullptr)) {
        std::string why;
        h->use_jit = lqmpc::jit_available(h->device, nx, nu, N, c.mode, &why);
        if (!h->use_jit) g_err = "run-time compile unavailable, using the generic kernels: " + why;
    }
    h->use_wg = !use_spec(h, nx, nu, N) && !h->use_jit && (h->opt.kernel == LQMPC_KERNEL_AUTO || h->opt.kernel == LQMPC_KERNEL_WORKGROUP) &&
                lqmpc::wg_supported(p, c.lb, c.ub);
    if (h->opt.kernel == LQMPC_KERNEL_WORKGROUP && !h->use_wg)
        return fail(LQMPC_ERR_UNSUPPORTED, "the workgroup kernel needs 32 < N*nu <= 128, nx <= 16, nu <= 8 and an LDS image within 160 KiB");
    if (h->use_jit) {
        // the generic kernel only ever sees the instances the 16-lane-row kernel hands back: a bounded workspace, walked by a grid-stride loop
        p.ws_stride = JIT_FALLBACK_COLS;
        rc = ensure(h, h->ws, (size_t)lqmpc::generic_ws_entries(nx, nu, N) * (size_t)p.ws_stride * sizeof(double));
        if (rc) return rc;
        p.ws = (double *)h->ws.p;
    } else if (!use_spec(h, nx, nu, N) && !h->use_wg) {
        p.ws_stride = (c.Bsz + 63) / 64 * 64;
        const size_t bytes = (size_t)lqmpc::generic_ws_entries(nx, nu, N) * (size_t)p.ws_stride * sizeof(double);
        rc = ensure(h, h->ws, bytes);
        if (rc) return rc;
        p.ws = (double *)h->ws.p;
    }
    return 0;
}

// Difficulty ordering (options.order): probe launch -> (bucket, position) per instance -> scatter, hardest first.
// On success p.perm points at the permutation (slot -> instance).
// The order only has to group similar instances: a bucket sort over the logarithm of the probe's key (lqmpc_probe_kernel),
// two launches instead of the eight of a full radix / merge sort (41 us of a 0.51 ms C3 launch).  Which instance gets which
// position inside a bucket depends on the order of the atomics, i.e. the permutation is not reproducible run to run; the
// results are, because no result depends on the instances that share a wavefront.
static int build_order(lqmpc_handle *h, KParams &p)
{
    const size_t B = (size_t)p.Bsz;
    if (B > (size_t)INT32_MAX) return fail(LQMPC_ERR_BAD_ARG, "ordering supports up to 2^31-1 instances");
    int rc = ensure(h, h->key, B * sizeof(double));
    if (!rc) rc = ensure(h, h->perm, B * sizeof(int));
    if (!rc) rc = ensure_fail(h, ((size_t)p.Bsz + FAIL_HDR) * sizeof(int));
    const size_t rec_doubles = (size_t)(p.nx * p.nx + p.nx * p.nu + p.nx);
    if (!rc) rc = ensure(h, h->rec, B * rec_doubles * sizeof(double));
    if (rc) return rc;
    // The hand-back count and the order's counters.  First call (or a new buffer): one fill of everything.  After that the probe
    // launch itself zeroes the count and the set of counters the NEXT call will use (the sets alternate), so a call costs no fill
    // launch (4 us of a 0.39 ms C3 call).  Any failure below leaves hist_ready false: the next call fills again.
    if (!h->hist_ready) {
        HIP_TRY(hipMemsetAsync(h->fail.p, 0, FAIL_HDR * sizeof(int), h->stream));
        h->hist_turn = 0;
    }
    h->hist_ready = false;
    h->fail_cleared = true;
    KParams q = p;
    q.mode = lqmpc::MODE_PROBE;
    q.perm = nullptr;
    q.key = (double *)h->key.p;
    q.hist = (int *)h->fail.p + 16 + (size_t)h->hist_turn * HIST_INTS;
    q.hist_next = (int *)h->fail.p + 16 + (size_t)(h->hist_turn ^ 1) * HIST_INTS;
    q.fail_count = (int *)h->fail.p;
    q.stage = (double *)h->rec.p;
    const char *name = nullptr;
    if (h->use_jit) {
        std::string why;
        if (!lqmpc::launch_jit(h->device, q, h->stream, &name, &why)) return fail(LQMPC_ERR_UNSUPPORTED, "probe launch failed: " + why);
    } else if (!lqmpc::launch_spec(q, h->stream, &name)) return fail(LQMPC_ERR_UNSUPPORTED, "probe launch failed");
    lqmpc::launch_order_scatter(q, (int *)h->perm.p, h->stream);
    HIP_TRY(hipGetLastError());
    h->hist_turn ^= 1;
    h->hist_ready = true;
    p.perm = (const int *)h->perm.p;
    p.rec = (const double *)h->rec.p;
    return 0;
}

static int launch(lqmpc_handle *h, const KParams &p)
{
    const char *name = "lqmpc_generic_kernel";
    if (use_spec(h, p.nx, p.nu, p.N)) {
        if (!lqmpc::launch_spec(p, h->stream, &name)) return fail(LQMPC_ERR_UNSUPPORTED, "specialisation launch failed");
    } else if (h->use_wg) {
        if (!lqmpc::launch_wg(p, h->stream, &name)) return fail(LQMPC_ERR_HIP, "workgroup kernel launch failed");
    } else {
        lqmpc::launch_generic(p, h->stream);
    }
    h->last_kernel = name;
    HIP_TRY(hipGetLastError());
    return 0;
}

// Which layout (measured at C3 / C2 shapes).  Rollouts: the 16-lane-row kernel (two waves per SIMD,
// P and W packed in LDS) for every batch size -- 0.77 ms against 0.97 ms for the two-tier launch of the packed kernel at
// C3's 65 536 instances, and it also fills the GPU where the packed kernel (16 or 32 instances per wavefront) cannot.
// One-shot entry points and the fused sweep (built for one wave per SIMD): up to `limit` instances.
// options.layout = 0/1 forces the choice (0 gives the packed kernel and its two-tier launch).
static bool use_r16(const lqmpc_handle *h, const KParams &p, int64_t Bsz, int64_t limit)
{
    const int force = h->opt.layout;
    const bool ok = h->opt.kernel == LQMPC_KERNEL_AUTO && p.presolve && p.warm_start && lqmpc::r16_available(p.nx, p.nu, p.N) &&
                    Bsz <= INT32_MAX;
    if (ok && lqmpc::r16_lanes(p.nx, p.nu, p.N) == 64) return force != 0;   // vs one wave per instance in the packed family too: always
    return ok && (force >= 0 ? force == 1 : Bsz <= limit);
}

// 16-lane-row kernel on the whole batch, then the packed kernel over whatever it handed back (device-side list)
// the hand-back list of this call: count zeroed (by build_order's fill if it ran), list behind the order's counters
static int prepare_hand_back(lqmpc_handle *h, KParams &p)
{
    int rc = ensure_fail(h, ((size_t)p.Bsz + FAIL_HDR) * sizeof(int));
    if (rc) return rc;
    if (!h->fail_cleared) HIP_TRY(hipMemsetAsync(h->fail.p, 0, 2 * sizeof(int), h->stream));
    h->fail_cleared = false;
    p.fail_count = (int *)h->fail.p;
    p.fail_list = (int *)h->fail.p + FAIL_HDR;
    return 0;
}

// status / iters of the instances on a device-side list: worse status, summed iterations (grid-stride over the list)
__global__ void lqmpc_merge_listed_kernel(int *st, int *it, const int *st2, const int *it2, const int *list, const int *count)
{
    const int n = *count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = list[i];
        if (st) st[b] = st[b] > st2[b] ? st[b] : st2[b];
        if (it) it[b] += it2[b];
    }
}

static int launch_r16_with_hand_back(lqmpc_handle *h, KParams &p)
{
    int rc = prepare_hand_back(h, p);
    if (rc) return rc;
    const char *name = nullptr;
    if (h->use_jit) {
        std::string why;
        if (!lqmpc::launch_jit(h->device, p, h->stream, &name, &why)) return fail(LQMPC_ERR_HIP, "run-time compiled kernel: " + why);
    } else if (!lqmpc::launch_r16(p, h->stream, &name)) return fail(LQMPC_ERR_UNSUPPORTED, "r16 launch failed");
    HIP_TRY(hipGetLastError());
    KParams f = p;
    f.perm = p.fail_list; f.count_dev = p.fail_count; f.fail_list = nullptr; f.fail_count = nullptr; f.nwide = 0;
    const char *name2 = nullptr;
    // second pass over the hand-back list: the packed kernel (interior point + polish) where the shape has one, the generic kernel otherwise
    auto second = [&](const KParams &g) -> bool {
        if (h->use_jit) { lqmpc::launch_generic_list(g, JIT_FALLBACK_COLS, h->stream); return true; }
        return lqmpc::launch_spec(g, h->stream, &name2);
    };
    if (p.mode == lqmpc::MODE_SWEEP) {         // the packed kernel has no fused mode: max V_N, then the rollout, over the list
        // the two passes write status / iters of the listed instances: the max-V_N pass into side buffers, merged below
        // (status = the worse of the two parts, iters = their sum -- the contract of lqmpc_sweep_batch, as on the two-launch path)
        if (p.status) { rc = ensure(h, h->st2, (size_t)p.Bsz * sizeof(int32_t)); if (rc) return rc; }
        if (p.iters) { rc = ensure(h, h->it2, (size_t)p.Bsz * sizeof(int32_t)); if (rc) return rc; }
        f.mode = lqmpc::MODE_MAXVN;
        f.status = p.status ? (int *)h->st2.p : nullptr;
        f.iters = p.iters ? (int *)h->it2.p : nullptr;
        if (!second(f)) return fail(LQMPC_ERR_UNSUPPORTED, "hand-back launch failed");
        f.mode = lqmpc::MODE_ROLLOUT;
        f.status = p.status; f.iters = p.iters;
    }
    if (!second(f)) return fail(LQMPC_ERR_UNSUPPORTED, "hand-back launch failed");
    HIP_TRY(hipGetLastError());
    if (p.mode == lqmpc::MODE_SWEEP && (p.status || p.iters)) {
        const unsigned blocks = (unsigned)(p.Bsz < 65536 ? (p.Bsz + 255) / 256 : 256);
        hipLaunchKernelGGL(lqmpc_merge_listed_kernel, dim3(blocks), dim3(256), 0, h->stream, p.status, p.iters,
                           (const int *)h->st2.p, (const int *)h->it2.p, (const int *)p.fail_list, (const int *)p.fail_count);
        HIP_TRY(hipGetLastError());
    }
    h->last_kernel = name;
    return 0;
}

__global__ void lqmpc_merge_status_kernel(int *st, int *it, const int *st2, const int *it2, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st) st[i] = st[i] > st2[i] ? st[i] : st2[i];
    if (it) it[i] += it2[i];
}

extern "C" {

int lqmpc_reserve(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T)
{
    if (!h) return fail(LQMPC_ERR_BAD_ARG, "handle is NULL");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    if (T < 0) return fail(LQMPC_ERR_BAD_ARG, "T must be >= 0");
    HIP_TRY(hipSetDevice(h->device));
    rc = ensure(h, h->shared, 8192 * sizeof(double));
    if (rc) return rc;
    const bool fast = use_spec(h, nx, nu, N) || (h->opt.kernel == LQMPC_KERNEL_AUTO && h->opt.jit != 0 && lqmpc::jit_r16_shape(nx, nu, N, nullptr));
    if (fast) {
        // the hand-back list / order counters of the 16-lane-row kernels, and -- for rollouts long and large enough to be walked in
        // difficulty order (T >= 4: lqmpc_rollout_batch_dev) -- the key, permutation and record buffers of the order
        rc = ensure_fail(h, ((size_t)Bsz + FAIL_HDR) * sizeof(int));
        if (!rc && T >= 4 && Bsz >= 1024) {
            rc = ensure(h, h->key, (size_t)Bsz * sizeof(double));
            if (!rc) rc = ensure(h, h->perm, (size_t)Bsz * sizeof(int));
            if (!rc) rc = ensure(h, h->rec, (size_t)Bsz * (size_t)(nx * nx + nx * nu + nx) * sizeof(double));
        }
    }
    if (!rc && !use_spec(h, nx, nu, N)) {
        const bool wg_only = !fast && N * nu > 32;          // (the workgroup kernel needs no workspace)
        const size_t cols = fast ? (size_t)JIT_FALLBACK_COLS : (size_t)(Bsz + 63) / 64 * 64;
        if (!wg_only) rc = ensure(h, h->ws, (size_t)lqmpc::generic_ws_entries(nx, nu, N) * cols * sizeof(double));
    }
    return rc;
}

int lqmpc_solve_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, const double *dA, const double *dB,
                          const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                          const double *dx0, const double *x_ref, const double *u_ref, double *du0, double *dVN,
                          int32_t *dstatus, int32_t *diters)
{
    if (!h || !dA || !dB || !dx0 || !du0 || !dVN) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    Call c{nx, nu, N, 0, 0, lqmpc::MODE_SOLVE, 0, Bsz, Q, R, P, lb, ub, x_ref, u_ref, nullptr, nullptr, nullptr};
    KParams p;
    int rc = prepare(h, c, p);
    if (rc) return rc;
    p.A = dA; p.B = dB; p.x0 = dx0; p.u0 = du0; p.VN = dVN; p.status = dstatus; p.iters = diters;
    if (h->use_jit || (use_spec(h, nx, nu, N) && use_r16(h, p, Bsz, INT32_MAX))) return launch_r16_with_hand_back(h, p);
    return launch(h, p);
}

int lqmpc_rollout_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, const double *dA, const double *dB,
                            const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                            const double *dx0, const double *A_true, const double *B_true, int true_per_instance,
                            const double *x_ref, const double *u_ref, double *dJT, double *dX, double *dU,
                            int32_t *dstatus, int32_t *diters)
{
    if (!h || !dA || !dB || !dx0 || !dJT || !A_true || !B_true) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    if (T < 1 || T > 100000) return fail(LQMPC_ERR_BAD_ARG, "T must be in [1,100000]");
    Call c{nx, nu, N, T, 0, lqmpc::MODE_ROLLOUT, true_per_instance ? 1 : 0, Bsz, Q, R, P, lb, ub, x_ref, u_ref,
           A_true, B_true, nullptr};
    KParams p;
    int rc = prepare(h, c, p);
    if (rc) return rc;
    p.A = dA; p.B = dB; p.x0 = dx0; p.JT = dJT; p.X = dX; p.U = dU; p.status = dstatus; p.iters = diters;
    if (true_per_instance) { p.At = A_true; p.Bt = B_true; }
    const bool spec = use_spec(h, nx, nu, N);
    const int order = h->opt.order < 0 ? ((spec && p.presolve && T >= 4 && Bsz >= 1024) ? 1 : 0) : (spec ? h->opt.order : 0);
    if (h->use_jit || (spec && use_r16(h, p, Bsz, INT32_MAX))) {
        // (measured at C3: the sorted walk pays for its probe and scatter launches from about 8 192 instances: 0.26 against 0.31 ms
        // at 16 384, 0.25 against 0.19 ms at 4 096)
        const int r16_order = h->opt.order < 0 ? ((T >= 4 && Bsz >= 8192) ? 1 : 0) : h->opt.order;
        if (r16_order) {
            rc = build_order(h, p);
            if (rc) return rc;
        }
        return launch_r16_with_hand_back(h, p);
    }
    if (order) {
        rc = build_order(h, p);
        if (rc) return rc;
        // the hardest instances (first in the order) in the 16-lane-row layout: see lqmpc_spec_tiered_kernel
        if (lqmpc::spec_tiered_available(nx, nu, N) && p.presolve && p.warm_start && Bsz <= INT32_MAX) {
            long long nw = h->opt.nwide >= 0 ? h->opt.nwide : (Bsz / 8 < 4096 ? Bsz / 8 : 4096) / 4 * 4;   // measured: ~one 16-lane-row wave per SIMD
            p.nwide = nw < 0 ? 0 : (nw > Bsz ? Bsz : nw);
        }
        if (p.nwide > 0) {
            rc = prepare_hand_back(h, p);
            if (rc) return rc;
            rc = launch(h, p);
            if (rc) return rc;
            const char *name = h->last_kernel;
            // second pass: whatever the wide tier handed back (status 3), packed, from the start of the rollout
            KParams f = p;
            f.perm = p.fail_list; f.count_dev = p.fail_count; f.fail_list = nullptr; f.fail_count = nullptr; f.nwide = 0;
            rc = launch(h, f);
            h->last_kernel = name;
            return rc;
        }
    }
    return launch(h, p);
}

int lqmpc_max_vn_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int K, const double *dA, const double *dB,
                           const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                           const double *x0s, const double *x_ref, const double *u_ref, double *dMV, int32_t *dstatus,
                           int32_t *diters)
{
    if (!h || !dA || !dB || !x0s || !dMV) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    if (K < 1 || K > 1024) return fail(LQMPC_ERR_BAD_ARG, "K must be in [1,1024]");
    Call c{nx, nu, N, 0, K, lqmpc::MODE_MAXVN, 0, Bsz, Q, R, P, lb, ub, x_ref, u_ref, nullptr, nullptr, x0s};
    KParams p;
    int rc = prepare(h, c, p);
    if (rc) return rc;
    p.A = dA; p.B = dB; p.MV = dMV; p.status = dstatus; p.iters = diters;
    // (n <= 10 and a large batch: the packed kernel's one lane per instance wins the K-state loop, 0.24 against 0.36 ms at C2 x 65 536)
    if (h->use_jit || (use_spec(h, nx, nu, N) && use_r16(h, p, Bsz, N * nu <= 10 ? 32768 : INT32_MAX))) return launch_r16_with_hand_back(h, p);
    return launch(h, p);
}

int lqmpc_sweep_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, int K, const double *dA, const double *dB,
                          const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                          const double *dx0, const double *x0s, const double *A_true, const double *B_true,
                          int true_per_instance, const double *x_ref, const double *u_ref, double *dJT, double *dMV,
                          int32_t *dstatus, int32_t *diters)
{
    if (!h || !dA || !dB || !dx0 || !x0s || !dJT || !dMV || !A_true || !B_true) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    if (T < 1 || T > 100000) return fail(LQMPC_ERR_BAD_ARG, "T must be in [1,100000]");
    if (K < 1 || K > 1024) return fail(LQMPC_ERR_BAD_ARG, "K must be in [1,1024]");
    Call c{nx, nu, N, T, K, lqmpc::MODE_SWEEP, true_per_instance ? 1 : 0, Bsz, Q, R, P, lb, ub, x_ref, u_ref, A_true, B_true, x0s};
    KParams p;
    int rc = prepare(h, c, p);
    if (rc) return rc;
    if (h->use_jit || (use_spec(h, nx, nu, N) && use_r16(h, p, Bsz, INT32_MAX))) {
        // one launch: condensing and W once per instance, K open-loop QPs, then the closed loop
        p.A = dA; p.B = dB; p.x0 = dx0; p.JT = dJT; p.MV = dMV; p.status = dstatus; p.iters = diters;
        if (true_per_instance) { p.At = A_true; p.Bt = B_true; }
        const int order = h->opt.order < 0 ? ((T >= 4 && Bsz >= 8192) ? 1 : 0) : h->opt.order;
        if (order) {
            rc = build_order(h, p);
            if (rc) return rc;
        }
        return launch_r16_with_hand_back(h, p);
    }
    // no fused kernel for this shape / size: the two launches, status = worse of the two, iters = their sum
    int32_t *st2 = nullptr, *it2 = nullptr;
    if (dstatus) { rc = ensure(h, h->st2, (size_t)Bsz * sizeof(int32_t)); if (rc) return rc; st2 = (int32_t *)h->st2.p; }
    if (diters) { rc = ensure(h, h->it2, (size_t)Bsz * sizeof(int32_t)); if (rc) return rc; it2 = (int32_t *)h->it2.p; }
    rc = lqmpc_max_vn_batch_dev(h, nx, nu, N, Bsz, K, dA, dB, Q, R, P, lb, ub, x0s, x_ref, u_ref, dMV, st2, it2);
    if (rc) return rc;
    rc = lqmpc_rollout_batch_dev(h, nx, nu, N, Bsz, T, dA, dB, Q, R, P, lb, ub, dx0, A_true, B_true, true_per_instance, x_ref, u_ref,
                                 dJT, nullptr, nullptr, dstatus, diters);
    if (rc) return rc;
    if (dstatus || diters) {
        hipLaunchKernelGGL(lqmpc_merge_status_kernel, dim3((unsigned)((Bsz + 255) / 256)), dim3(256), 0, h->stream, dstatus, diters,
                           st2, it2, (long long)Bsz);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

}  // extern "C"

// ---- host-buffer flavours: stage in, run, stage out ----
namespace {
// Small calls (the reference's own call shape is ONE instance per solve(), utils_class.py:269) are dominated by the number of
// copies, not their size: below PACK_LIMIT bytes in total every array of the call is packed into one pinned host block and one
// device block -- one copy in, one copy out.  Larger calls copy array by array straight from / to the caller's buffers.
constexpr size_t PACK_LIMIT = 256 * 1024;
struct Stager {
    lqmpc_handle *h;
    int slot = 0;
    int rc = 0;
    bool packed = false;
    size_t off = 0, in_end = 0;
    struct Out { void *host; size_t off, bytes; } outs[12];
    int nout = 0;
    // total: an upper bound of the bytes of all in() / out() arrays of the call
    void begin(size_t total)
    {
        total += 24 * 256;
        if (total > PACK_LIMIT) return;
        rc = ensure(h, h->arena, total);
        if (!rc) rc = ensure_host(h->arena_pin, total);
        if (rc) return;
        if (h->arena_pin.ev_valid) { (void)hipEventSynchronize(h->arena_pin.ev); h->arena_pin.ev_valid = false; }
        packed = true;
    }
    size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; }
    // after the last in(): the one copy of the inputs
    void upload()
    {
        if (rc || !packed) return;
        if (nout == 0) in_end = off;                              // (the first out() froze it otherwise)
        if (in_end == 0) return;
        hipError_t e = hipMemcpyAsync(h->arena.p, h->arena_pin.p, in_end, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) rc = fail(LQMPC_ERR_HIP, std::string("H2D: ") + hipGetErrorString(e));
    }
    // after the launches: the one copy of the outputs, the wait, and the hand-over to the caller's arrays
    int finish()
    {
        if (rc) return rc;
        if (packed && nout > 0) {
            hipError_t e = hipMemcpyAsync((char *)h->arena_pin.p + in_end, (char *)h->arena.p + in_end, off - in_end, hipMemcpyDeviceToHost, h->stream);
            if (e != hipSuccess) return fail(LQMPC_ERR_HIP, std::string("D2H: ") + hipGetErrorString(e));
        }
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return fail(LQMPC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        if (packed)
            for (int k = 0; k < nout; ++k) memcpy(outs[k].host, (char *)h->arena_pin.p + outs[k].off, outs[k].bytes);
        return 0;
    }
    template <typename T>
    T *in(const T *host, size_t count)
    {
        if (rc || !host) return nullptr;
        if (packed) {
            const size_t o = take(count * sizeof(T));
            memcpy((char *)h->arena_pin.p + o, host, count * sizeof(T));
            return (T *)((char *)h->arena.p + o);
        }
        DevBuf &b = h->stage[slot++];
        rc = ensure(h, b, count * sizeof(T));
        if (rc) return nullptr;
        hipError_t e = hipMemcpyAsync(b.p, host, count * sizeof(T), hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { rc = fail(LQMPC_ERR_HIP, std::string("H2D: ") + hipGetErrorString(e)); return nullptr; }
        return (T *)b.p;
    }
    template <typename T>
    T *out(T *host, size_t count)
    {
        if (rc || !host) return nullptr;
        if (packed) {
            if (nout == 0 && in_end == 0) in_end = off;           // (a call without inputs to upload)
            const size_t o = take(count * sizeof(T));
            outs[nout++] = Out{host, o, count * sizeof(T)};
            return (T *)((char *)h->arena.p + o);
        }
        DevBuf &b = h->stage[slot++];
        rc = ensure(h, b, count * sizeof(T));
        return rc ? nullptr : (T *)b.p;
    }
    template <typename T>
    void back(T *host, const T *dev, size_t count)
    {
        if (rc || !host || packed) return;                        // packed: finish() copies everything back at once
        hipError_t e = hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) rc = fail(LQMPC_ERR_HIP, std::string("D2H: ") + hipGetErrorString(e));
    }
};
}  // namespace

extern "C" {

int lqmpc_solve_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, const double *A, const double *B,
                      const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                      const double *x0, const double *x_ref, const double *u_ref, double *u0, double *VN,
                      int32_t *status, int32_t *iters)
{
    if (!h || !A || !B || !x0 || !u0 || !VN) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Stager s{h};
    const size_t b = (size_t)Bsz;
    s.begin(b * 8 * (size_t)(nx * nx + nx * nu + nx + nu + 2));
    const double *dA = s.in(A, b * nx * nx), *dB = s.in(B, b * nx * nu), *dx0 = s.in(x0, b * nx);
    double *du0 = s.out(u0, b * nu), *dVN = s.out(VN, b);
    int32_t *dst = s.out(status, b), *dit = s.out(iters, b);
    s.upload();
    if (s.rc) return s.rc;
    rc = lqmpc_solve_batch_dev(h, nx, nu, N, Bsz, dA, dB, Q, R, P, lb, ub, dx0, x_ref, u_ref, du0, dVN, dst, dit);
    if (rc) return rc;
    s.back(u0, du0, b * nu); s.back(VN, dVN, b); s.back(status, dst, b); s.back(iters, dit, b);
    return s.finish();
}

int lqmpc_rollout_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, const double *A, const double *B,
                        const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                        const double *x0, const double *A_true, const double *B_true, int true_per_instance,
                        const double *x_ref, const double *u_ref, double *JT, double *X, double *U, int32_t *status,
                        int32_t *iters)
{
    if (!h || !A || !B || !x0 || !JT || !A_true || !B_true) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    if (T < 1 || T > 100000) return fail(LQMPC_ERR_BAD_ARG, "T must be in [1,100000]");
    HIP_TRY(hipSetDevice(h->device));
    Stager s{h};
    const size_t b = (size_t)Bsz;
    s.begin(b * 8 * (size_t)(2 * (nx * nx + nx * nu) + nx + 2 + (X ? nx * (T + 1) : 0) + (U ? nu * T : 0)));
    const double *dA = s.in(A, b * nx * nx), *dB = s.in(B, b * nx * nu), *dx0 = s.in(x0, b * nx);
    const double *dAt = A_true, *dBt = B_true;
    if (true_per_instance) { dAt = s.in(A_true, b * nx * nx); dBt = s.in(B_true, b * nx * nu); }
    double *dJT = s.out(JT, b), *dX = s.out(X, b * nx * (T + 1)), *dU = s.out(U, b * nu * T);
    int32_t *dst = s.out(status, b), *dit = s.out(iters, b);
    s.upload();
    if (s.rc) return s.rc;
    rc = lqmpc_rollout_batch_dev(h, nx, nu, N, Bsz, T, dA, dB, Q, R, P, lb, ub, dx0, dAt, dBt, true_per_instance, x_ref,
                                 u_ref, dJT, dX, dU, dst, dit);
    if (rc) return rc;
    s.back(JT, dJT, b); s.back(X, dX, b * nx * (T + 1)); s.back(U, dU, b * nu * T);
    s.back(status, dst, b); s.back(iters, dit, b);
    return s.finish();
}

int lqmpc_max_vn_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int K, const double *A, const double *B,
                       const double *Q, const double *R, const double *P, const double *lb, const double *ub,
                       const double *x0s, const double *x_ref, const double *u_ref, double *MV, int32_t *status,
                       int32_t *iters)
{
    if (!h || !A || !B || !x0s || !MV) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Stager s{h};
    const size_t b = (size_t)Bsz;
    s.begin(b * 8 * (size_t)(nx * nx + nx * nu + 2));
    const double *dA = s.in(A, b * nx * nx), *dB = s.in(B, b * nx * nu);
    double *dMV = s.out(MV, b);
    int32_t *dst = s.out(status, b), *dit = s.out(iters, b);
    s.upload();
    if (s.rc) return s.rc;
    rc = lqmpc_max_vn_batch_dev(h, nx, nu, N, Bsz, K, dA, dB, Q, R, P, lb, ub, x0s, x_ref, u_ref, dMV, dst, dit);
    if (rc) return rc;
    s.back(MV, dMV, b); s.back(status, dst, b); s.back(iters, dit, b);
    return s.finish();
}

int lqmpc_sweep_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, int T, int K, const double *A, const double *B,
                      const double *Q, const double *R, const double *P, const double *lb, const double *ub, const double *x0,
                      const double *x0s, const double *A_true, const double *B_true, int true_per_instance, const double *x_ref,
                      const double *u_ref, double *JT, double *MV, int32_t *status, int32_t *iters)
{
    if (!h || !A || !B || !x0 || !x0s || !JT || !MV || !A_true || !B_true) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    Stager s{h};
    const size_t b = (size_t)Bsz;
    s.begin(b * 8 * (size_t)(2 * (nx * nx + nx * nu) + nx + 3));
    const double *dA = s.in(A, b * nx * nx), *dB = s.in(B, b * nx * nu), *dx0 = s.in(x0, b * nx);
    const double *dAt = A_true, *dBt = B_true;
    if (true_per_instance) { dAt = s.in(A_true, b * nx * nx); dBt = s.in(B_true, b * nx * nu); }
    double *dJT = s.out(JT, b), *dMV = s.out(MV, b);
    int32_t *dst = s.out(status, b), *dit = s.out(iters, b);
    s.upload();
    if (s.rc) return s.rc;
    rc = lqmpc_sweep_batch_dev(h, nx, nu, N, Bsz, T, K, dA, dB, Q, R, P, lb, ub, dx0, x0s, dAt, dBt, true_per_instance, x_ref, u_ref,
                               dJT, dMV, dst, dit);
    if (rc) return rc;
    s.back(JT, dJT, b); s.back(MV, dMV, b); s.back(status, dst, b); s.back(iters, dit, b);
    return s.finish();
}

// ---- the bound coefficients of data_generation: lqmpc_bounds.hip ----
int lqmpc_bounds_batch_dev(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, const double *dA, const double *dB,
                           const double *Q, const double *R, const double *lb, const double *ub, const double *de_A,
                           const double *de_B, const double *dMV, const double *x, const double *p3, double V_expert,
                           double *dK, double *dalpha, double *dbeta, double *dxi, double *deta, double *dbound, double *deps,
                           double *daux, int32_t *dstatus)
{
    if (!h || !dA || !dB || !Q || !R || !lb || !ub || !de_A || !de_B || !x || !p3) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    for (int k = 0; k < nu; ++k)
        if (!(ub[k] > lb[k]) || !std::isfinite(lb[k]) || !std::isfinite(ub[k]) || lb[k] == 0.0 || ub[k] == 0.0)
            return fail(LQMPC_ERR_BAD_ARG, "every input needs a finite box with lb < ub and non-zero bounds (rows of F_u)");
    for (int k = 0; k < 3; ++k)
        if (!(p3[k] > 0.0)) return fail(LQMPC_ERR_BAD_ARG, "p must be positive");
    HIP_TRY(hipSetDevice(h->device));
    // batch-shared block: Q | R | Q^-1 | R^-1 | lb | ub | x | p | (qmax qmin rmax rmin V_expert bar_u bar_d_u)
    std::vector<double> sh;
    auto put = [&](const double *src, int count) { int off = (int)sh.size(); sh.insert(sh.end(), src, src + count); return off; };
    lqmpc::BoundsParams bp;
    memset(&bp, 0, sizeof bp);
    bp.oQ = put(Q, nx * nx); bp.oR = put(R, nu * nu);
    std::vector<double> Qi(Q, Q + nx * nx), Ri(R, R + nu * nu);
    double qmax, qmin, rmax, rmin;
    if (!host_sym_eig_extremes(Q, nx, qmax, qmin) || !host_sym_eig_extremes(R, nu, rmax, rmin) || !(qmin > 0.0) || !(rmin > 0.0) ||
        !host_spd_inverse(Qi, nx) || !host_spd_inverse(Ri, nu))
        return fail(LQMPC_ERR_BAD_ARG, "Q and R must be symmetric positive definite");
    bp.oQinv = put(Qi.data(), nx * nx); bp.oRinv = put(Ri.data(), nu * nu);
    bp.olb = put(lb, nu); bp.oub = put(ub, nu); bp.ox = put(x, nx); bp.op = put(p3, 3);
    double bar_u = 0.0, bar_du = 0.0;                    // max |u|^2 and max |u1 - u2|^2 over the box (utils.py:592-650 solves two QPs for these)
    for (int k = 0; k < nu; ++k) { bar_u += std::max(lb[k] * lb[k], ub[k] * ub[k]); bar_du += (ub[k] - lb[k]) * (ub[k] - lb[k]); }
    const double sc[7] = {qmax, qmin, rmax, rmin, V_expert, bar_u, bar_du};
    bp.osc = put(sc, 7);
    // for the on-chip kernels: identity / zero weights (Gamma'Gamma is the condensed Hessian with them), and whether Q, R are multiples of I
    std::vector<double> eye((size_t)nx * nx, 0.0), zer((size_t)nu * nu, 0.0);
    for (int a = 0; a < nx; ++a) eye[(size_t)a * nx + a] = 1.0;
    bp.oI = put(eye.data(), nx * nx); bp.oZ = put(zer.data(), nu * nu);
    auto scalar_mult = [](const double *M, int n) { for (int a = 0; a < n; ++a) for (int c = 0; c < n; ++c) if (M[a * n + c] != (a == c ? M[0] : 0.0)) return false; return true; };
    bp.q_scalar = scalar_mult(Q, nx) ? 1 : 0; bp.r_scalar = scalar_mult(R, nu) ? 1 : 0;
    bp.qs = Q[0]; bp.rs = R[0];
    rc = ensure(h, h->shared, std::max(sh.size(), (size_t)8192) * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->shared.p, sh.data(), sh.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->shared_host.clear();                              // the solver's cached copy of the shared block is gone
    bp.o = lqmpc::bounds_offsets(nx, nu, N);
    bp.nx = nx; bp.nu = nu; bp.N = N; bp.Bsz = Bsz;
    bp.A = dA; bp.B = dB; bp.eA = de_A; bp.eB = de_B; bp.MV = dMV; bp.sh = (const double *)h->shared.p;
    bp.K = dK; bp.alpha = dalpha; bp.beta = dbeta; bp.xi = dxi; bp.eta = deta; bp.bound = dbound; bp.eps = deps; bp.aux = daux;
    bp.status = dstatus;
    // the on-chip kernels (registers + LDS, two launches): prebuilt for the reference's shapes, compiled at run time for the others;
    // the HBM-workspace kernel below stays for what neither serves (kernel = generic forces it)
    if (h->opt.kernel != LQMPC_KERNEL_GENERIC) {
        rc = ensure(h, h->ws, (size_t)Bsz * 10 * sizeof(double));
        if (rc) return rc;
        bp.rec = (double *)h->ws.p; bp.b0 = 0; bp.b1 = Bsz;
        bool done = lqmpc::launch_bounds_chip(bp, h->stream);
        if (!done && h->opt.jit != 0) {
            std::string why;
            done = lqmpc::launch_jit_bounds(h->device, bp, h->stream, &why);
            if (!done) g_err = "on-chip bounds kernels unavailable, using the workspace kernel: " + why;
        }
        if (done) {
            HIP_TRY(hipGetLastError());
            h->last_kernel = "lqmpc_bounds_small_kernel + lqmpc_bounds_big_kernel";
            return 0;
        }
    }
    // workspace: at most ~1 GiB at a time, the batch in chunks of whole wavefronts
    const size_t per = (size_t)bp.o.total * sizeof(double);
    long long chunk = (long long)(((size_t)1 << 30) / per) / 64 * 64;
    if (chunk < 64) chunk = 64;
    if (chunk > (Bsz + 63) / 64 * 64) chunk = (Bsz + 63) / 64 * 64;
    rc = ensure(h, h->ws, per * (size_t)chunk);
    if (rc) return rc;
    bp.ws = (double *)h->ws.p; bp.stride = chunk;
    for (long long b0 = 0; b0 < Bsz; b0 += chunk) {
        bp.b0 = b0; bp.b1 = std::min<long long>(Bsz, b0 + chunk);
        lqmpc::launch_bounds(bp, h->stream);
        HIP_TRY(hipGetLastError());
    }
    h->last_kernel = "lqmpc_bounds_kernel";
    return 0;
}

int lqmpc_bounds_batch(lqmpc_handle *h, int nx, int nu, int N, int64_t Bsz, const double *A, const double *B,
                       const double *Q, const double *R, const double *lb, const double *ub, const double *e_A,
                       const double *e_B, const double *MV, const double *x, const double *p3, double V_expert,
                       double *K, double *alpha, double *beta, double *xi, double *eta, double *bound, double *eps,
                       double *aux, int32_t *status)
{
    if (!h || !A || !B || !e_A || !e_B) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    int rc = check_dims(nx, nu, N, Bsz);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    // (the workspace of the kernel is h->ws; staging uses its own buffers.  Twelve staging slots are in use at most.)
    Stager s{h};
    const size_t b = (size_t)Bsz;
    const double *dA = s.in(A, b * nx * nx), *dB = s.in(B, b * nx * nu), *deA = s.in(e_A, b), *deB = s.in(e_B, b), *dMV = s.in(MV, b);
    double *dK = s.out(K, b * nu * nx), *dal = s.out(alpha, b), *dbe = s.out(beta, b), *dxi = s.out(xi, b), *det = s.out(eta, b);
    double *dbd = s.out(bound, b), *dep = s.out(eps, b);
    if (s.rc) return s.rc;
    // aux (8 doubles per instance) and status share no staging slot with the above: reuse the ordering buffers of the solver
    double *dax = nullptr; int32_t *dst = nullptr;
    if (aux) { rc = ensure(h, h->rec, b * 8 * sizeof(double)); if (rc) return rc; dax = (double *)h->rec.p; }
    if (status) { rc = ensure(h, h->st2, b * sizeof(int32_t)); if (rc) return rc; dst = (int32_t *)h->st2.p; }
    rc = lqmpc_bounds_batch_dev(h, nx, nu, N, Bsz, dA, dB, Q, R, lb, ub, deA, deB, dMV, x, p3, V_expert, dK, dal, dbe, dxi, det, dbd, dep,
                                dax, dst);
    if (rc) return rc;
    s.back(K, dK, b * nu * nx); s.back(alpha, dal, b); s.back(beta, dbe, b); s.back(xi, dxi, b); s.back(eta, det, b);
    s.back(bound, dbd, b); s.back(eps, dep, b); s.back(aux, dax, b * 8); s.back(status, dst, b);
    return s.finish();
}

int lqmpc_timer_begin(lqmpc_handle *h)
{
    if (!h) return fail(LQMPC_ERR_BAD_ARG, "handle is NULL");
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return 0;
}

int lqmpc_timer_end(lqmpc_handle *h, float *ms)
{
    if (!h || !ms) return fail(LQMPC_ERR_BAD_ARG, "NULL argument");
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

}  // extern "C"
