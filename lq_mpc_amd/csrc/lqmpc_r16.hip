// lqmpc_r16.hip -- solve / rollout / max-V_N for n = N*nu <= 32 with one instance per 16-lane row of a wavefront
// (four per wave), matrix row i of an instance in lane i % 16 (slot i / 16).  Device code: lqmpc_r16_body.h.
//
// Why another layout.  The packed register-resident kernel (lqmpc_spec.hip) factors the masked n x n matrix in
// every active-set iteration: ~4000 instructions per iteration and wave of 16 instances, one wave per SIMD (512
// registers, 40 KiB of LDS), so a sorted launch lasts as long as the wave holding the instances that stay
// constrained for all T steps (C3: 78 iterations, 1.1 ms), and a batch below ~16 000 instances leaves most SIMDs
// idle.  Here an iteration costs ~1000 instructions per wave of 4 instances because it only ever solves the
// SMALLER side of the active-set system:
//   with A the active set (signs s), F the free set, r_A = v_unc,A - s h_A and W = P^-1,
//     |A| <= |F| (dual side):    W_AA lam = r_A,          v_F = v_unc,F - W_FA lam,          gradient on A = -lam
//     |A| >  |F| (primal side):  P_FF dlt = P_FA r_A,     v_F = v_unc,F + dlt,               gradient on A = P_AF dlt - P_AA r_A
//   both are the same iterate of the primal-dual active-set method the other kernels use; min(|A|, |F|) <= 16.
// P and W live in LDS with full rows (7.8 KB per instance at n = 20: every access is row base + constant), the
// gathered system is solved in registers by Gauss-Jordan elimination with DPP row broadcasts (v_fmac_f64_dpp
// row_newbcast: the pivot row reaches the 16 lanes of an instance without leaving the vector registers).  W itself
// comes from P by the same elimination.
//
// Used for whole batches of every size (all entry points; the host picks, lqmpc_api.hip: use_r16) and as the wide
// tier of the packed family's sorted rollouts (lqmpc_spec_tiered_kernel, options.layout = 0).
// An instance whose active set does not settle within the iteration cap is reported with status 3 internally; the
// host re-runs exactly those instances on the packed kernel (interior point + active-set finishing) in a second
// launch over a device-side list.
#include <cstdio>
#include "lqmpc_r16_body.h"

namespace lqmpc {

// Register budget: two waves per SIMD (256 registers) -- the n <= 10 shapes as they are, the larger 16-lane shapes and the
// one-instance-per-wavefront mapping in the low-register build (OCC = 2: rolled set-up loops, packed W, constants in LDS).
// The fully unrolled one-wave build of those 16-lane shapes lives in lqmpc_r16_lat.hip and serves small batches.
template <int NX, int NU, int N, int MODE, int LPI>
struct R16Build {
    static constexpr int OCC = ((LPI == 16 && N * NU > 10) || LPI == 64) ? 2 : 1;
    // two waves per SIMD only where their LDS fits as well ((2,1,30): 33 KB per wavefront -> one wave, the whole register file)
    static constexpr long long LDS_BYTES = (long long)(64 / LPI) * R16<NX, NU, N, LPI, (OCC == 2)>::INST * 8;
    static constexpr int WAVES = ((OCC == 2 || (N * NU <= 10 && LPI == 16)) && LDS_BYTES * 8 <= 160 * 1024) ? 2 : 1;
};

template <int NX, int NU, int N, int MODE, int LPI>
__global__ void __launch_bounds__(64, (R16Build<NX, NU, N, MODE, LPI>::WAVES)) lqmpc_r16_kernel(KParams p)
{
    using C = R16<NX, NU, N, LPI, (R16Build<NX, NU, N, MODE, LPI>::OCC == 2)>;
    __shared__ double lds_raw[C::IPW * C::INST];
    r16_body<NX, NU, N, MODE, LPI, R16Build<NX, NU, N, MODE, LPI>::OCC>(p, lds_raw, (long long)blockIdx.x * C::IPW, p.Bsz);
}

struct R16Entry {
    int nx, nu, N, lpi;
    const char *name;
    void (*launch)(const KParams &, hipStream_t);
};

template <int NX, int NU, int N, int LPI>
static void launch_r16_one(const KParams &p, hipStream_t stream)
{
    constexpr int IPW = 64 / LPI;
    const dim3 grid((unsigned)((p.Bsz + IPW - 1) / IPW));
    if (p.mode == MODE_SOLVE) hipLaunchKernelGGL((lqmpc_r16_kernel<NX, NU, N, MODE_SOLVE, LPI>), grid, dim3(64), 0, stream, p);
    else if (p.mode == MODE_MAXVN) hipLaunchKernelGGL((lqmpc_r16_kernel<NX, NU, N, MODE_MAXVN, LPI>), grid, dim3(64), 0, stream, p);
    else if (p.mode == MODE_SWEEP) hipLaunchKernelGGL((lqmpc_r16_kernel<NX, NU, N, MODE_SWEEP, LPI>), grid, dim3(64), 0, stream, p);
    else {
#ifdef LQMPC_R16_PROF
        long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_r16_prof), z, sizeof z);
#endif
        hipLaunchKernelGGL((lqmpc_r16_kernel<NX, NU, N, MODE_ROLLOUT, LPI>), grid, dim3(64), 0, stream, p);
#ifdef LQMPC_R16_PROF
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_r16_prof), sizeof z);
        fprintf(stderr, "r16 prof (block %d ticks): set-up on the matrix core %lld v_r %lld | rollout %lld\n", PROFBLK, z[5], z[6], z[7]);
        fprintf(stderr, "r16 prof (grid): waves %lld  set-up %.0f  total %.0f ticks/wave | free wave-steps %lld at %.0f ticks | busy wave-steps %lld at %.0f ticks, "
                "%.2f wave-iterations each (steps 1..T-1); %lld wave-iterations on the general (primal-capable) path in all\n", z[15], (double)z[13] / z[15], (double)z[14] / z[15], z[10], (double)z[8] / (z[10] ? z[10] : 1),
                z[11], (double)z[9] / (z[11] ? z[11] : 1), (double)z[12] / (z[11] ? z[11] : 1), z[16]);
        fprintf(stderr, "r16 prof (grid, ticks/wave): set-up on the matrix core %.0f v_r %.0f\n", (double)z[22] / z[15], (double)z[23] / z[15]);
        fprintf(stderr, "r16 prof (grid): %lld dual-only wave-iterations at %.0f ticks, %lld general at %.0f ticks\n", z[26], (double)z[24] / (z[26] ? z[26] : 1), z[16],
                (double)z[25] / (z[16] ? z[16] : 1));
        fprintf(stderr, "r16 prof (grid): step 0: %.0f ticks/wave, %.2f wave-iterations/wave, %.2f of them general; %lld of %lld wavefronts ever took the general (P-reading) path\n",
                (double)z[27] / z[15], (double)z[28] / z[15], (double)z[29] / z[15], z[30], z[15]);
#endif
    }
}

#define R16E(NX, NU, N) {NX, NU, N, 16, "lqmpc_r16_kernel<" #NX "," #NU "," #N ">", launch_r16_one<NX, NU, N, 16>}
#define R64E(NX, NU, N) {NX, NU, N, 64, "lqmpc_r64_kernel<" #NX "," #NU "," #N ">", launch_r16_one<NX, NU, N, 64>}
static const R16Entry g_r16[] = {
    R16E(4, 2, 10),     // C3 (headline)
    R16E(2, 1, 10),     // C2
    R16E(2, 1, 5), R16E(2, 1, 6), R16E(2, 1, 7), R16E(2, 1, 8), R16E(2, 1, 9),   // C1 and the reference's horizon sweep
    R16E(2, 1, 20), R16E(2, 1, 30),   // mpc_test.py (N_open = 20), V_expert (N_opc = 30)
    R64E(4, 2, 20),     // C4 (n = 40): one instance per wavefront (LPI = 64), same algorithm, v_readlane broadcasts
};

bool launch_r16_lat(const KParams &p, hipStream_t stream);   // lqmpc_r16_lat.hip

static const R16Entry *find_r16(int nx, int nu, int N)
{
    for (const R16Entry &e : g_r16)
        if (e.nx == nx && e.nu == nu && e.N == N) return &e;
    return nullptr;
}

bool r16_available(int nx, int nu, int N) { return find_r16(nx, nu, N) != nullptr; }
int r16_lanes(int nx, int nu, int N) { const R16Entry *e = find_r16(nx, nu, N); return e ? e->lpi : 0; }

bool launch_r16(const KParams &p, hipStream_t stream, const char **name)
{
    const R16Entry *e = find_r16(p.nx, p.nu, p.N);
    if (!e) return false;
    // a batch of at most one wave per SIMD: the latency build where the shape has one (lqmpc_r16_lat.hip)
    const bool lat = e->lpi == 16 && (p.r16_build >= 0 ? p.r16_build == 1 : p.Bsz <= 4096);
    if (!(lat && launch_r16_lat(p, stream))) e->launch(p, stream);
    if (name) *name = e->name;
    return true;
}

}  // namespace lqmpc
