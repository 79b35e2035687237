// lqmpc_bounds.h -- parameter block of lqmpc_bounds_kernel (lqmpc_bounds.hip), shared with the host side in lqmpc_api.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace lqmpc {

// workspace offsets (doubles per instance): the doubling iterates and their temporaries, the gain, M_d = A^d B, one N n_u x N n_u matrix
struct BoundsOff {
    int Ak, Gk, Hk, T1, T2, T3, T4, K, U1, U2, Md, E, V, total;
};
BoundsOff bounds_offsets(int nx, int nu, int N);

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
};

void launch_bounds(const BoundsParams &p, hipStream_t stream);

}  // namespace lqmpc
