// lqmpc_bounds.h -- parameter block of lqmpc_bounds_kernel (lqmpc_bounds.hip), shared with the host side in lqmpc_api.hip.
#pragma once
#ifndef LQMPC_JIT
#include <hip/hip_runtime.h>
#endif

namespace lqmpc {

// workspace offsets (doubles per instance): the doubling iterates and their temporaries, the gain, M_d = A^d B, one N n_u x N n_u matrix
struct BoundsOff {
    int Ak, Gk, Hk, T1, T2, T3, T4, K, U1, U2, Md, E, V, total;
};
#ifndef LQMPC_JIT
BoundsOff bounds_offsets(int nx, int nu, int N);
#endif

struct BoundsParams {
    int nx, nu, N;
    long long Bsz;                 // instances in the caller's arrays (their stride)
    long long b0, b1;              // this launch handles instances [b0, b1)
    long long stride;              // workspace: instances per entry row
    const double *A, *B;           // per instance, instance-minor (device)
    const double *eA, *eB, *MV;    // per instance (device); MV may be null (xi, eta, bound then use M_V = 0)
    const double *sh;              // shared block (device): Q | R | Q^-1 | R^-1 | lb | ub | x | p | scalars
    int oQ, oR, oQinv, oRinv, olb, oub, ox, op, osc;   // scalars: qmax qmin rmax rmin V_expert bar_u bar_d_u
    BoundsOff o;
    double *ws;
    double *K, *Pinf, *alpha, *beta, *xi, *eta, *bound, *eps, *aux;   // outputs (device, instance-minor; any may be null)
    int *status;
    // the on-chip kernels (lqmpc_bounds_chip.h)
    double *rec;                   // BOUNDS_REC doubles per instance, instance-minor: what bounds_small leaves for bounds_big
    int oI, oZ;                    // shared block: an nx x nx identity, an nu x nu zero (Gamma'Gamma = the condensed Hessian with unit weights)
    int q_scalar, r_scalar;        // Q = qs I / R = rs I: hat H = kron(R, I_N) + qs Gamma'Gamma, and its smallest eigenvalue rs + qs lambda_min(Gamma'Gamma)
    double qs, rs;
};

#ifndef LQMPC_JIT
void launch_bounds(const BoundsParams &p, hipStream_t stream);
// the on-chip pair of kernels for [b0, b1) where the shape has a prebuilt instantiation; false otherwise
bool launch_bounds_chip(const BoundsParams &p, hipStream_t stream);
#endif

}  // namespace lqmpc
