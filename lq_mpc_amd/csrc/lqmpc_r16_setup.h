// lqmpc_r16_setup.h -- the per-instance set-up of the 16-lane-row kernels on the fp64 matrix core:
//   W = P^-1 (packed or full rows, LDS), P = 2 (Gamma'Qbar Gamma + Rbar) (LDS), G = -P^-1 Fq (registers, row layout)
// for one model (A, B) and the batch-shared weights (utils_class.py:59-75 is the cost being condensed).
//
// Everything is a product of 4x4 matrices held one element per lane ("register matrix", zero-padded): v_mfma_f64_4x4x4_4b_f64
// computes D = A_op B_op + C for four independent 4x4x4 blocks; block g lives in the lanes with (lane >> 2) & 3 == g, and
// A_op[i][k] is taken from lane 16k + 4g + i, B_op[k][j] from lane 16k + 4g + j, D[i][j] lands in lane 16i + 4g + j (measured:
// tools/ubench/mfma4_layout.hip).  So with the convention "entry [x][y] of block g in lane 16x + 4g + y" for a, b, c and the result:
//     mm4(a, b, c) = a' b + c,
// no LDS round trip, no barrier and no cross-lane instruction between dependent products.  With LPI = 16 the four blocks are the
// four instances of the wavefront (block g = the instance whose rows live in lanes 16g .. 16g+15 everywhere else in the kernel: the
// set-up addresses its inputs and its LDS by g, and hands its results over through LDS); with LPI = 64 (one instance per
// wavefront) they are four different tiles of the one instance.  A 16-lane DPP row holds row x of all four blocks, so DPP row
// broadcasts cannot serve here: the small inverse of Re is done with products as well (adjugate J Re J', det I = Re adj).
//
// Instead of condensing P and inverting it (Gauss-Jordan with a 20..40-pivot dependent chain: rounds 1-2), W comes from the
// Riccati / innovations form of the same quadratic:  with Sg_N = P_T and, for j = N-1 .. 0,
//     Re_j = R + B'Sg_{j+1}B,  K_j = Re_j^-1 B'Sg_{j+1}A,  Acl_j = A - B K_j,  Sg_j = Q + A'Sg_{j+1}Acl_j,
// the change of variables w_j = u_j + K_j x_j decouples the cost:  U'HU = sum_j w_j'Re_j w_j,  U = T w with T unit lower block
// triangular,  T_kj = -rho_k(j) B,  rho_k(j) = K_k Acl_{k-1} .. Acl_{j+1}  (k > j).  Hence
//     W = (2H)^-1 = 1/2 T D^-1 T' = sum_M T(:, M) (D_M^-1 / 2) T(:, M)'       (one update per column tile M of 4 / NU stages, on 4x4 tiles),
//     G rows of stage k = -K_k Acl_{k-1} .. Acl_0 = -rho_k(-1)                    (what the rows rho_k have become after stage 0),
// and the rows rho_k advance inside the same backward sweep (rho_k(j-1) = rho_k(j) Acl_j), so nothing is stored per stage.
// Sizes: state matrices of up to 8 x 8 are 2 x 2 tiles (TX = 2: every product of the recursion becomes TX^2..TX^3 tile products);
// stage blocks of 1, 2 or 4 inputs tile the rows exactly, 3 inputs are padded to 4 with a dummy input (zero column of B, unit
// weight: decoupled, its rows are never stored).  The inverse of Re for 4 inputs goes by 2 x 2 blocks (Schur complement), again
// with products and element-wise reciprocals only.
// P keeps its Lyapunov / Toeplitz form (round 2): block (bi, bj) = B' Lt_{bi+1} A^(bi-bj) B, Lt_N = P_T, Lt_k = Q + A'Lt_{k+1}A,
// assembled tile by tile along the block diagonals so that the powers A^d B stream through one register -- and only when a
// wavefront first needs it (r16_build_P: the primal side of an iteration).
#pragma once
#include "lqmpc_wg_linalg.h"

namespace lqmpc {

constexpr int SETUP_MAX_NX = 8, SETUP_MAX_NU = 4;   // what this set-up serves (the run-time compile asks: lqmpc_jit.hip)

// (a)' b + c on register matrices (four independent blocks per wavefront)
__device__ __forceinline__ double mm4(double a, double b, double c = 0.0) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int NX, int NU, int N, int LPI>
struct SetupT {
    static_assert(NX >= 1 && NX <= SETUP_MAX_NX, "state matrices are at most 2 x 2 tiles of 4 x 4");
    static_assert(NU >= 1 && NU <= SETUP_MAX_NU, "stage blocks of 1..4 rows");
    static constexpr int TX = (NX + 3) / 4;                   // tiles per side of a state matrix
    static constexpr int NUP = (NU == 3) ? 4 : NU;            // rows of a stage block in the tiled (padded) row space
    static constexpr int n = N * NU, npad = N * NUP;
    static constexpr int NT = (npad + 3) / 4;                 // 4-row tiles of the npad x npad matrices
    static constexpr int SPT = 4 / NUP;                       // stages per tile
    static constexpr int NTM = (LPI == 16) ? NT : (NT + 3) / 4;   // tile registers: LPI = 64 deals tile I to block I % 4, register I / 4
    static constexpr bool EXACT_TILES = (NUP == NU) && (n % 4 == 0);   // the 4-row tiles cover exactly the rows of the problem
    static constexpr int tile_of_stage(int j) { return (j * NUP) / 4; }
    static constexpr int off_of_stage(int j) { return (j * NUP) % 4; }
    // has tile I rows of a stage beyond j (i.e. is its rho non-zero when stage j is processed)?
    static constexpr bool live(int I, int j) { return ((I + 1) * SPT < N ? (I + 1) * SPT : N) - 1 > j; }
    // row of the problem for a row of the tiled space (-1: a dummy input's row, or beyond the horizon)
    __device__ static __forceinline__ int row_of(int rp) { return (rp < npad && rp % NUP < NU) ? (rp / NUP) * NU + rp % NUP : -1; }
};

// Inverse of the 2 x 2 block at rows / columns o, o+1 of the register matrix M (symmetric positive definite there), zero elsewhere:
// adj(M) = J M J' with J = [0 1; -1 0];  M adj(M) = det I;  det into all four lanes of the block by a product with ones.
// Not positive definite: NaN (it spreads through every later product into G, and the solver hands the instance back).
__device__ __forceinline__ double inv2(double M, int r, int c, int o)
{
    const bool in = r >= o && r < o + 2 && c >= o && c < o + 2;
    const double Jt = (r == o && c == o + 1) ? -1.0 : ((r == o + 1 && c == o) ? 1.0 : 0.0);
    const double ones = in ? 1.0 : 0.0;
    const double blk = in ? M : 0.0;
    const double adj = mm4(mm4(blk, Jt), Jt);
    const double det = mm4(ones, mm4(blk, adj));
    const double e00ok = (r == o && c == o && !(M > 0.0)) ? __builtin_nan("") : 1.0;
    const double idet = (in && !(det > 0.0)) ? __builtin_nan("") : frcp(in ? det : 1.0);
    return in ? adj * idet * e00ok : 0.0;
}

// Inverse of the leading NUP x NUP block of the register matrix Re (symmetric positive definite), zero elsewhere.
template <int NUP>
__device__ __forceinline__ double small_inverse(double Re, int r, int c)
{
    if constexpr (NUP == 1) {
        const double inv = (Re > 0.0) ? frcp(Re) : __builtin_nan("");
        return (r == 0 && c == 0) ? inv : 0.0;
    } else if constexpr (NUP == 2) {
        return inv2(Re, r, c, 0);
    } else {
        // [E F; F' H]^-1 by 2 x 2 blocks, S = H - F'E^-1 F:  [E^-1 + X S X', -X; -X', S^-1] with X = E^-1 F S^-1
        const double Fb = (r < 2 && c >= 2) ? Re : 0.0;
        const double Ei = inv2(Re, r, c, 0);
        const double EiF = mm4(Ei, Fb);                                // E^-1 F      (rows 0-1, columns 2-3)
        const double Sc = mm4(-Fb, EiF, (r >= 2 && c >= 2) ? Re : 0.0); // H - F'E^-1 F
        const double Si = inv2(Sc, r, c, 2);
        const double FtEi = mm4(Fb, Ei);                               // F'E^-1      (rows 2-3, columns 0-1)
        const double X = mm4(FtEi, Si);                                // E^-1 F S^-1
        const double Xt = mm4(Si, FtEi);                               // S^-1 F'E^-1
        return mm4(Xt, FtEi, Ei) - X - Xt + Si;
    }
}

// what the set-up reads of the kernel parameters, by value (a reference to KParams passed to the non-inlined r16_build_P would
// make the whole block address-taken: every later p.xxx becomes a scratch load instead of a scalar load of the kernel argument)
struct SetupArgs {
    const double *rec, *A, *B, *sh;
    long long Bsz;
    int oQ, oP, oR;
};
__device__ __forceinline__ SetupArgs setup_args(const KParams &p) { return SetupArgs{p.rec, p.A, p.B, p.sh, p.Bsz, p.so.Q, p.so.P, p.so.R}; }
// ... and of the cold-start guess (r16_setup_mfma<.., ROLL = true>): the first state, the box, whether the guess applies at all.
// (Kept out of SetupArgs: past 64 bytes the by-value argument of r16_build_P goes through scratch.)
struct RollArgs {
    const double *x0;
    int oLb, oUb, roll;
};
__device__ __forceinline__ RollArgs roll_args(const KParams &p) { return RollArgs{p.x0, p.so.lb, p.so.ub, p.has_lin == 0}; }

// Zero-padded loads of one instance's model (instance-minor arrays or the probe's instance-major records) and of the shared weights.
// Both layouts are "base + element index x stride" (records: stride 1 from rec + bg REC; arrays: stride Bsz from A + bg, B + bg), so an
// element costs one 64-bit multiply-add whichever layout the call uses (the per-element choice between two address computations
// was a sixth of the set-up's vector instructions).
template <int NX, int NU>
struct ModelLd {
    const SetupArgs &p;
    long long bg;
    const double *bA, *bB;
    long long se;
    static constexpr int REC = NX * NX + NX * NU + NX;
    __device__ __forceinline__ ModelLd(const SetupArgs &p_, long long bg_) : p(p_), bg(bg_)
    {
        bA = p.rec ? p.rec + bg * REC : p.A + bg;
        bB = p.rec ? p.rec + bg * REC + NX * NX : p.B + bg;
        se = p.rec ? 1 : p.Bsz;
    }
    __device__ __forceinline__ double A(int a, int k) const
    {
        const bool v = a < NX && k < NX;
        const double x = bA[(long long)(v ? a * NX + k : 0) * se];
        return v ? x : 0.0;
    }
    __device__ __forceinline__ double B(int a, int k) const
    {
        const bool v = a < NX && k >= 0 && k < NU;
        const double x = bB[(long long)(v ? a * NU + k : 0) * se];
        return v ? x : 0.0;
    }
    __device__ __forceinline__ double S(int o, int a, int k, int dim) const
    {
        const bool v = a >= 0 && k >= 0 && a < dim && k < dim;
        const double x = p.sh[o + (v ? a * dim + k : 0)];
        return v ? x : 0.0;
    }
};

// one 4x4 tile of a symmetric matrix to LDS: lane (r, c) holds entry (4I + r, 4J + c) of the tiled row space; lower triangle only
// (packed), or both triangles of full rows; predicated stores (upper triangle, dummy rows) go to the dummy slot
template <class T, bool PACKED>
__device__ __forceinline__ void store_tile(wg::ldsd *M, int dummy, int I, int J, double val, bool on, int r, int c)
{
    constexpr int LDW = T::n + 1;
    if constexpr (PACKED && T::EXACT_TILES) {
        // every row of every tile is a row of the problem: the packed index is r (r + 1) / 2 + c (once per lane) + r (4 I) + a constant of
        // the tile, and only the diagonal tiles have an upper triangle to skip (two instructions per tile instead of seven)
        const int idx = (r * (r + 1) / 2 + c) + r * (4 * I) + (8 * I * I + 2 * I + 4 * J);
        const bool low = on && (J < I || c <= r);
        M[low ? idx : dummy] = val;
        return;
    }
    const int row = T::row_of(4 * I + r), col = T::row_of(4 * J + c);
    const bool low = on && row >= 0 && col >= 0 && col <= row;
    if constexpr (PACKED) M[low ? row * (row + 1) / 2 + col : dummy] = val;
    else {
        M[low ? row * LDW + col : dummy] = val;
        M[(low && col < row) ? col * LDW + row : dummy] = val;
    }
}

// P = 2 (H + Rbar) of one instance to LDS, in its Lyapunov / Toeplitz form: tile (I, I - D) = sum_p (Lt_{k+1} B placed at stage k's
// columns)' [A^(D SPT + p) B | A^(D SPT + p - 1) B | ..],  k = I SPT + p.  Only the primal side of an active-set iteration reads P
// (|A| > n / 2), so the kernels call this the first time a wavefront gets there (a quarter of the wavefronts of C3's default mix,
// one in a hundred of C4's) instead of in every set-up.  Ends with a barrier.  NOT inlined: inside the iteration loop its ~80 live
// registers would be charged to every iteration (the sweep / C4 builds spill); as a call its frame costs only where it runs.
template <int NX, int NU, int N, int LPI, bool PACKED>
__device__ __attribute__((noinline)) void r16_build_P(const SetupArgs p, long long bg, wg::ldsd *Lg, int oP, int oD)
{
    using T = SetupT<NX, NU, N, LPI>;
    constexpr int NT = T::NT, SPT = T::SPT, NTM = T::NTM, TX = T::TX, NUP = T::NUP;
    const int lane = threadIdx.x, r = lane >> 4, c = lane & 3, g = (lane >> 2) & 3;
    wg::ldsd *Pp = Lg + oP;
    const int dP = oD - oP;
    const ModelLd<NX, NU> ld(p, bg);
    double A[TX][TX], At[TX][TX], Q[TX][TX], Lt[TX][TX];
    double Bpl[SPT][TX];                                              // B in column block q of a tile
#pragma unroll
    for (int a = 0; a < TX; ++a) {
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            A[a][b] = ld.A(4 * a + r, 4 * b + c); At[a][b] = ld.A(4 * b + c, 4 * a + r);
            Q[a][b] = ld.S(p.oQ, 4 * a + r, 4 * b + c, NX); Lt[a][b] = ld.S(p.oP, 4 * a + r, 4 * b + c, NX);
        }
#pragma unroll
        for (int qq = 0; qq < SPT; ++qq) Bpl[qq][a] = ld.B(4 * a + r, c - qq * NUP);
    }
    double ap[NTM][SPT][TX];                                          // a-operands of my tiles' stages
#pragma unroll
    for (int m = 0; m < NTM; ++m)
#pragma unroll
        for (int pp = 0; pp < SPT; ++pp)
#pragma unroll
            for (int a = 0; a < TX; ++a) ap[m][pp][a] = 0.0;
    sfor<0, N>([&](auto kc) {
        constexpr int k = N - 1 - decltype(kc)::value;
        constexpr int I = k / SPT, pp = k % SPT;
#pragma unroll
        for (int a = 0; a < TX; ++a) {
            double v = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < TX; ++k2) v = mm4(Lt[k2][a], Bpl[pp][k2], v);
            if constexpr (LPI == 16) ap[I][pp][a] = v;
            else ap[I / 4][pp][a] = (g == I % 4) ? v : ap[I / 4][pp][a];
        }
        if constexpr (k > 0) {
            double LA[TX][TX];
#pragma unroll
            for (int a = 0; a < TX; ++a)
#pragma unroll
                for (int b = 0; b < TX; ++b) {
                    double v = 0.0;
#pragma unroll
                    for (int k2 = 0; k2 < TX; ++k2) v = mm4(Lt[k2][a], A[k2][b], v);
                    LA[a][b] = v;
                }
#pragma unroll
            for (int a = 0; a < TX; ++a)
#pragma unroll
                for (int b = 0; b < TX; ++b) {
                    double v = Q[a][b];
#pragma unroll
                    for (int k2 = 0; k2 < TX; ++k2) v = mm4(A[k2][a], LA[k2][b], v);
                    Lt[a][b] = v;
                }
        }
    });
    double Rd = 0.0;                                                  // R on the diagonal blocks of a diagonal tile
#pragma unroll
    for (int qq = 0; qq < SPT; ++qq) {
        const int rr = r - qq * NUP, cc = c - qq * NUP;
        const double x = ld.S(p.oR, rr, cc, NU);
        Rd = (rr >= 0 && rr < NU && cc >= 0 && cc < NU) ? x : Rd;
    }
    double X[TX];                                                     // X_d = [A^d B | A^(d-1) B | ..] (columns of negative powers: zero)
#pragma unroll
    for (int a = 0; a < TX; ++a) X[a] = Bpl[0][a];
    sfor<0, NT>([&](auto Dc) {
        constexpr int D = decltype(Dc)::value;
        double acc[NTM];
#pragma unroll
        for (int m = 0; m < NTM; ++m) acc[m] = (D == 0) ? Rd : 0.0;
        sfor<0, SPT>([&](auto pc) {
            constexpr int pp = decltype(pc)::value;
            constexpr int d = D * SPT + pp;
            if constexpr (d > 0) {
                double Xn[TX];
#pragma unroll
                for (int a = 0; a < TX; ++a) {
                    double v = (d <= SPT - 1) ? Bpl[d <= SPT - 1 ? d : 0][a] : 0.0;
#pragma unroll
                    for (int k2 = 0; k2 < TX; ++k2) v = mm4(At[k2][a], X[k2], v);
                    Xn[a] = v;
                }
#pragma unroll
                for (int a = 0; a < TX; ++a) X[a] = Xn[a];
            }
            if constexpr (LPI == 16) {
                sfor<D, NT>([&](auto Ic) {
                    constexpr int I = decltype(Ic)::value;
#pragma unroll
                    for (int k2 = 0; k2 < TX; ++k2) acc[I] = mm4(ap[I][pp][k2], X[k2], acc[I]);
                });
            } else {
                sfor<D / 4, NTM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
#pragma unroll
                    for (int k2 = 0; k2 < TX; ++k2) acc[m] = mm4(ap[m][pp][k2], X[k2], acc[m]);
                });
            }
        });
        if constexpr (LPI == 16) {
            sfor<D, NT>([&](auto Ic) { constexpr int I = decltype(Ic)::value; store_tile<T, PACKED>(Pp, dP, I, I - D, 2.0 * acc[I], true, r, c); });
        } else {
            sfor<D / 4, NTM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const int I = 4 * m + g;
                store_tile<T, PACKED>(Pp, dP, I, I - D, 2.0 * acc[m], I >= D && I < NT, r, c);
            });
        }
    });
    __syncthreads();
}

// The set-up.  Lg: LDS of block g's instance (LPI = 16) or of the one instance (LPI = 64); oW: where W goes in it (packed lower
// triangle: index row (row + 1) / 2 + col; or full rows of stride n + 1, both triangles); oG: n NX doubles of scratch for G (may alias
// P / W: they are written after G has been read back); oD: a dummy slot for predicated stores.
// bg: the instance block g works on.  Lq: where the lane's OWN instance (q = lane / LPI) reads its rows of G back.
// On return: W in LDS (a barrier has been passed), G[s][a] = row (i + LPI s) of G in registers.  P: r16_build_P, on demand.
//
// ROLL (four instances per wavefront, short horizons): a first guess of the active set at the first state x0, for the cold start of
// the primal-dual active-set iteration.  The model is rolled forward under the gains of the sweep with the inputs clipped to the box,
//     u_j = clip(-K_j x_j),  x_{j+1} = A x_j + B u_j,   guess = the (stage, input) pairs that clipped, by side
// -- i.e. forward substitution through T with clipping.  It predicts the optimal face far better than "the rows of the unconstrained
// minimiser outside the box": primal-dual iterations at step 0 of C3's default mix 2.19 -> 1.40 per QP, hard mix 4.40 -> 2.51
// (tools/proto/cold_start_guess.py, warm_start_guess.py; later steps keep the shifted face of the previous step, which is better
// still).  On the matrix core like the rest: K_j' is kept per stage (N TX register matrices), a stage is three dependent products
// and a clip for the four instances at once.  Zero references and a centred box only: otherwise -K_j x_j is not the
// unconstrained input (ra.roll).  coldL / coldU: bit (j NU + k) per clipped pair, in the lanes of the instance (q = lane / 16) as everywhere
// outside the set-up; both zero: no guess.
template <int NX, int NU, int N, int LPI, bool PACKED, int RB, bool ROLL = false>
__device__ __forceinline__ void r16_setup_mfma(const SetupArgs &p, long long bg, wg::ldsd *Lg, wg::ldsd *Lq, int oW, int oG, int oD,
                                              double (&G)[RB][NX], const RollArgs &ra, unsigned &coldL, unsigned &coldU)
{
    static_assert(!ROLL || (LPI == 16 && N * NU <= 32), "the cold-start roll serves the four-instance mapping");
    using T = SetupT<NX, NU, N, LPI>;
    constexpr int n = T::n, NT = T::NT, SPT = T::SPT, NTM = T::NTM, TX = T::TX, NUP = T::NUP;
    const int lane = threadIdx.x, r = lane >> 4, c = lane & 3, g = (lane >> 2) & 3;
    wg::ldsd *Wp = Lg + oW, *Gs = Lg + oG;
    const int dW = oD - oW, dG = oD - oG;
    // ---- the model and the weights as register matrices (tiles [a][b]: rows 4a.., columns 4b..) ----
    const ModelLd<NX, NU> ld(p, bg);
    double A[TX][TX], Q[TX][TX], S[TX][TX];
    double Bp[TX], nBt[TX];                                           // B (n_x x 4, columns 0..NU-1) by row tiles; -(B tile)' as register matrices
    double Bpl[SPT][TX], nBpl[SPT][TX];                               // B (and -B) in column block q of a tile
#pragma unroll
    for (int a = 0; a < TX; ++a) {
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            A[a][b] = ld.A(4 * a + r, 4 * b + c);
            Q[a][b] = ld.S(p.oQ, 4 * a + r, 4 * b + c, NX); S[a][b] = ld.S(p.oP, 4 * a + r, 4 * b + c, NX);
        }
        Bp[a] = ld.B(4 * a + r, c); nBt[a] = -ld.B(4 * a + c, r);
#pragma unroll
        for (int qq = 0; qq < SPT; ++qq) { Bpl[qq][a] = ld.B(4 * a + r, c - qq * NUP); nBpl[qq][a] = -Bpl[qq][a]; }
    }
    const double Rp = (r == c && r >= NU && r < NUP) ? 1.0 : ld.S(p.oR, r, c, NU);   // (a dummy input weighs 1)

    // ---- backward sweep: Riccati recursion, the rows rho_k, W by one update per column tile ----
    // LPI = 16: W by 4 x 4 tiles, Wacc[I][J], J <= I.  LPI = 64 (tile I in block I % 4 of register I / 4): the four tiles of a
    // register are, lane for lane, the A / B operand of v_mfma_f64_16x16x4_f64 for rows 16m .. 16m+15 (lane 16x + 4g + y holds
    // [column x of the column tile][row 4g + y]) -- so W is accumulated by 16 x 16 blocks, Wb[m(m+1)/2 + m'], one instruction per
    // block and column tile, and no operand has to be replicated over the blocks.
    double Wacc[(LPI == 16) ? NTM : 1][(LPI == 16) ? NT : 1];
    wg::d4_t Wb[(LPI == 64) ? NTM * (NTM + 1) / 2 : 1];
    double rho[NTM][TX];                          // rho' of the rows of tile I (n_x x 4: column = row of the tile), by row tiles
    // T(:, M)' and (T(:, M) D_M^-1 / 2)' of the current column tile M on the rows of every tile from M on: register-matrix row
    // q NUP + u <-> column u of stage M SPT + q; filled stage by stage (the accumulator operand), used once per column tile
    double TtA[NTM], TDA[NTM], Dh = 0.0;
    double KT[ROLL ? N : 1][TX];                  // K_j' (n_x x 4, by row tiles) of every stage, for the roll after the sweep
#pragma unroll
    for (int m = 0; m < NTM; ++m) {
        TDA[m] = 0.0; TtA[m] = 0.0;
#pragma unroll
        for (int a = 0; a < TX; ++a) rho[m][a] = 0.0;
        if constexpr (LPI == 16) {
#pragma unroll
            for (int J = 0; J < NT; ++J) Wacc[m][J] = 0.0;
        }
    }
    if constexpr (LPI == 64) {
#pragma unroll
        for (int e = 0; e < NTM * (NTM + 1) / 2; ++e) Wb[e] = wg::d4_t{0.0, 0.0, 0.0, 0.0};
    }
    // o' = Acl' o (+ add) on an n_x x 4 column of tiles
    auto advance = [&](double (&o)[TX], const double (&Acl)[TX][TX], const double (&add)[TX], bool with_add) {
        double t[TX];
#pragma unroll
        for (int a = 0; a < TX; ++a) {
            double v = with_add ? add[a] : 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) v = mm4(Acl[k][a], o[k], v);
            t[a] = v;
        }
#pragma unroll
        for (int a = 0; a < TX; ++a) o[a] = t[a];
    };
    // (x)' o + cc with x, o columns of tiles: a 4 x 4 register matrix
    auto dotc = [&](const double (&x)[TX], const double (&o)[TX], double cc) -> double {
        double v = cc;
#pragma unroll
        for (int k = 0; k < TX; ++k) v = mm4(x[k], o[k], v);
        return v;
    };
    sfor<0, N>([&](auto jc) {
        constexpr int j = N - 1 - decltype(jc)::value;
        constexpr int Ij = T::tile_of_stage(j), off = T::off_of_stage(j), q = off / NUP;
        constexpr bool first_of_tile = (j == N - 1) || (q == SPT - 1);        // (backward: the highest stage of column tile Ij comes first)
        double SA[TX][TX], SB[TX], F[TX], K[TX], Acl[TX][TX], Ktp[TX];
#pragma unroll
        for (int a = 0; a < TX; ++a) {
#pragma unroll
            for (int b = 0; b < TX; ++b) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < TX; ++k) v = mm4(S[k][a], A[k][b], v);
                SA[a][b] = v;
            }
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) w = mm4(S[k][a], Bp[k], w);
            SB[a] = w;
        }
        double Re = Rp;
#pragma unroll
        for (int k = 0; k < TX; ++k) Re = mm4(Bp[k], SB[k], Re);
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) v = mm4(Bp[k], SA[k][b], v);
            F[b] = v;
        }
        const double R0 = small_inverse<NUP>(Re, r, c);
#pragma unroll
        for (int b = 0; b < TX; ++b) K[b] = mm4(R0, F[b]);
        if constexpr (ROLL && off != 0) {                             // (off == 0: K' is Ktp below, as it stands)
#pragma unroll
            for (int b = 0; b < TX; ++b) KT[j][b] = mm4(F[b], R0);
        }
#pragma unroll
        for (int a = 0; a < TX; ++a)
#pragma unroll
            for (int b = 0; b < TX; ++b) Acl[a][b] = mm4(nBt[a], K[b], A[a][b]);
        if constexpr (j > 0) {
            double Zm[TX][TX];
#pragma unroll
            for (int a = 0; a < TX; ++a)
#pragma unroll
                for (int b = 0; b < TX; ++b) {
                    double v = 0.0;
#pragma unroll
                    for (int k = 0; k < TX; ++k) v = mm4(S[k][a], Acl[k][b], v);
                    Zm[a][b] = v;
                }
#pragma unroll
            for (int a = 0; a < TX; ++a)
#pragma unroll
                for (int b = 0; b < TX; ++b) {
                    double v = Q[a][b];
#pragma unroll
                    for (int k = 0; k < TX; ++k) v = mm4(A[k][a], Zm[k][b], v);
                    S[a][b] = v;
                }
        }
        const double SH = (r < NUP && c == off + r) ? 1.0 : 0.0;       // as a right factor: moves columns 0.. to off..; its transpose as a left factor: rows
        const double Iq = (r == c && r >= off && r < off + NUP) ? 1.0 : 0.0;   // the identity block of T(:, j), at the tile position of stage j
        const double R0h = 0.5 * R0;
        const double R1h = (off == 0) ? R0h : mm4(R0h, SH);           // Re^-1 / 2, columns moved to stage j's
        const double Rqq = (off == 0) ? R0h : mm4(SH, R1h);           // ... and rows
#pragma unroll
        for (int a = 0; a < TX; ++a) {
            Ktp[a] = mm4(K[a], SH);                                   // K' in the columns of stage j
            if constexpr (ROLL && off == 0) KT[j][a] = Ktp[a];
        }
        // T(:, M)' stage by stage; (T(:, M) D_M^-1 / 2)' = Dh T(:, M)' once per column tile, Dh = the blocks Re_j^-1 / 2 of its stages on the
        // diagonal (one product per tile and column tile instead of one per tile and stage)
        Dh = (first_of_tile ? 0.0 : Dh) + Rqq;
        if constexpr (LPI == 16) {
            sfor<Ij, NT>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                const double cT = (first_of_tile ? 0.0 : TtA[I]) + (I == Ij ? Iq : 0.0);
                if constexpr (T::live(I, j)) TtA[I] = dotc(nBpl[q], rho[I], cT);
                else TtA[I] = cT;                                   // (I == Ij and no row of a later stage in the tile yet)
            });
            if constexpr (q == 0) {
                sfor<Ij, NT>([&](auto Ic) { constexpr int I = decltype(Ic)::value; TDA[I] = mm4(Dh, TtA[I]); });
                sfor<Ij, NT>([&](auto Ic) {
                    constexpr int I = decltype(Ic)::value;
                    sfor<Ij, I + 1>([&](auto Jc) {
                        constexpr int J = decltype(Jc)::value;
                        Wacc[I][J] = mm4(TDA[I], TtA[J], Wacc[I][J]);
                    });
                });
            }
            sfor<Ij, NT>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                if constexpr (T::live(I, j)) advance(rho[I], Acl, Ktp, I == Ij);
                else {
#pragma unroll
                    for (int a = 0; a < TX; ++a) rho[I][a] = Ktp[a];
                }
            });
        } else {
            constexpr int m0 = Ij / 4;
            sfor<m0, NTM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const bool mine = (4 * m + g == Ij);
                TtA[m] = dotc(nBpl[q], rho[m], (first_of_tile ? 0.0 : TtA[m]) + (mine ? Iq : 0.0));
            });
            if constexpr (q == 0) {
                sfor<m0, NTM>([&](auto mc) { constexpr int m = decltype(mc)::value; TDA[m] = mm4(Dh, TtA[m]); });
                sfor<m0, NTM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    sfor<m0, m + 1>([&](auto nc) {
                        constexpr int m2 = decltype(nc)::value;
                        Wb[m * (m + 1) / 2 + m2] = __builtin_amdgcn_mfma_f64_16x16x4f64(TDA[m], TtA[m2], Wb[m * (m + 1) / 2 + m2], 0, 0, 0);
                    });
                });
            }
            sfor<m0, NTM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const bool mine = (4 * m + g == Ij);
                double add[TX];
#pragma unroll
                for (int a = 0; a < TX; ++a) add[a] = mine ? Ktp[a] : 0.0;
                advance(rho[m], Acl, add, true);
            });
        }
    });

    // ---- G: rho' (n_x x 4 per tile: lane (r, c) of row tile a holds rho[row 4I + c][state 4a + r]) -> rows, through LDS ----
#pragma unroll
    for (int m = 0; m < NTM; ++m) {
        const int I = (LPI == 16) ? m : 4 * m + g;
        const int row = T::row_of(4 * I + c);
#pragma unroll
        for (int a = 0; a < TX; ++a) Gs[(4 * a + r < NX && row >= 0) ? row * NX + 4 * a + r : dG] = -rho[m][a];
    }
    __syncthreads();
    {
        const int i = lane % LPI;
        const wg::ldsd *Gq = Lq + oG;
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int row = i + LPI * s;
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                const double x = Gq[row < n ? row * NX + a : 0];
                G[s][a] = row < n ? x : 0.0;
            }
        }
    }
    __syncthreads();

    // ---- the cold-start guess: saturated roll-forward from x0 (column 0 of a register matrix per row tile) ----
    // (here, in one block with the stores of W: its dependent chain of products runs under their address arithmetic and LDS issue)
    coldL = 0u; coldU = 0u;
    if constexpr (ROLL) {
        unsigned bL = 0u, bU = 0u;                                    // lane (r = k, c = 0) of block g: the flags of input k of instance g
        if (ra.roll) {
            constexpr int REC = NX * NX + NX * NU + NX;
            const double I4 = (r == c) ? 1.0 : 0.0;
            double At[TX][TX], xr[TX];
#pragma unroll
            for (int a = 0; a < TX; ++a) {
#pragma unroll
                for (int b = 0; b < TX; ++b) At[b][a] = mm4(A[a][b], I4);          // (A tile)': mm4(At[b][a], x_b) = A_ab x_b
                const bool v = c == 0 && 4 * a + r < NX;
                const int ia = v ? 4 * a + r : 0;
                const double x = p.rec ? p.rec[bg * REC + NX * NX + NX * NU + ia] : ra.x0[(long long)ia * p.Bsz + bg];
                xr[a] = v ? x : 0.0;
            }
            const int kk = r < NU ? r : 0;
            const double hk = (r < NU) ? 0.5 * (p.sh[ra.oUb + kk] - p.sh[ra.oLb + kk]) : 1.0;
            const unsigned one = (c == 0 && r < NU) ? (1u << r) : 0u;
            sfor<0, N>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double kx = 0.0;                                          // K_j x_j = -u_j before the clip
#pragma unroll
                for (int b = 0; b < TX; ++b) kx = mm4(KT[j][b], xr[b], kx);
                double xn[TX];                                            // A x_j first: it does not wait for the clip
                if constexpr (j + 1 < N) {
#pragma unroll
                    for (int a = 0; a < TX; ++a) {
                        double v = 0.0;
#pragma unroll
                        for (int b = 0; b < TX; ++b) v = mm4(At[b][a], xr[b], v);
                        xn[a] = v;
                    }
                }
                bL |= (kx > hk) ? (one << (j * NU)) : 0u;                 // u_j < -h: lower bound
                bU |= (kx < -hk) ? (one << (j * NU)) : 0u;
                const double un = fmin(fmax(kx, -hk), hk);
                if constexpr (j + 1 < N) {
#pragma unroll
                    for (int a = 0; a < TX; ++a) xr[a] = mm4(nBt[a], un, xn[a]);   // + (-B)(-u_j)
                }
            });
        }
        // the flags of instance gq sit in lanes 16 k + 4 gq (k < NU): gather them, hand each instance its own
        unsigned mLq[4], mUq[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            mLq[gq] = 0u; mUq[gq] = 0u;
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                mLq[gq] |= (unsigned)__builtin_amdgcn_readlane((int)bL, 16 * k + 4 * gq);
                mUq[gq] |= (unsigned)__builtin_amdgcn_readlane((int)bU, 16 * k + 4 * gq);
            }
        }
        coldL = r == 0 ? mLq[0] : (r == 1 ? mLq[1] : (r == 2 ? mLq[2] : mLq[3]));     // (r = lane >> 4 = the instance of the lane)
        coldU = r == 0 ? mUq[0] : (r == 1 ? mUq[1] : (r == 2 ? mUq[2] : mUq[3]));
    }

    // ---- W to LDS ----
    if constexpr (LPI == 16) {
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) store_tile<T, PACKED>(Wp, dW, I, J, Wacc[I][J], true, r, c);
    } else {
        // a 16 x 16 block of the tiled row space: lane l, register e holds entry (16m + l / 16 + 4e, 16m' + l % 16)
        constexpr int LDW = n + 1;
#pragma unroll
        for (int m = 0; m < NTM; ++m)
#pragma unroll
            for (int m2 = 0; m2 <= m; ++m2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = T::row_of(16 * m + (lane >> 4) + 4 * e), col = T::row_of(16 * m2 + (lane & 15));
                    const bool low = row >= 0 && col >= 0 && col <= row;
                    const double val = Wb[m * (m + 1) / 2 + m2][e];
                    if constexpr (PACKED) Wp[low ? row * (row + 1) / 2 + col : dW] = val;
                    else {
                        Wp[low ? row * LDW + col : dW] = val;
                        Wp[(low && col < row) ? col * LDW + row : dW] = val;
                    }
                }
    }
    __syncthreads();
}

}  // namespace lqmpc
