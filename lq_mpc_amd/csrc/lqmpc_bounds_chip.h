// lqmpc_bounds_chip.h -- the bound coefficients of one system (utils_class.py:837-859) without the HBM
// workspace of lqmpc_bounds.hip: two kernels per (n_x, n_u, N), everything in registers and LDS.
//
//   bounds_small: dlqr (doubling iteration for P_inf), the local radius, rho(A - BK), |A|_2, |B|_2, |K|_2, |Phi|_2 -- all products
//                 of n_x x n_x matrices.  Four instances per wavefront on the fp64 matrix core (register matrices of
//                 lqmpc_r16_setup.h: D = A'B + C on 4 x 4 tiles, block g = instance g); the SPD inverses by 2 x 2 blocks, spectral
//                 radii and spectral norms by repeated squaring of the normalised matrix (log rho = sum_j 2^-j log |Y_j|_F: no
//                 eigen-solver), Frobenius norms as two products with a matrix of ones.  No LDS, no cross-lane instruction.
//                 Leaves ten scalars per instance in HBM.
//   bounds_big:   |Gamma|_2 and lambda_min(hat H), both N n_u x N n_u.  One instance per 16 lanes (n <= 32) or per wavefront, matrix
//                 rows dealt to the lanes, the matrix in LDS.  Gamma'Gamma is the condensed Hessian of the same model with unit
//                 weights, so r16_build_P (Toeplitz form, matrix core) builds it; Householder tridiagonalisation with the lanes
//                 working on their rows; the extreme eigenvalues by multisection on the Sturm count (every lane of the instance
//                 counts at its own shift: 17x or 65x per pass instead of 2x).  hat H = kron(R, I_N) + Gamma' kron(Q, I_{N+1}) Gamma with
//                 the reference's index pairing (utils.py:316-319): for Q = q I it is kron(R, I_N) + q Gamma'Gamma (and for R = r I as
//                 well its smallest eigenvalue is r + q lambda_min(Gamma'Gamma): no second eigenproblem); a dense Q takes the formula
//                 as it stands, from the table A^d B.  Then the scalar formulas of utils.py:78-117, 186-584.
// Device code only (no standard-library header): prebuilt for the reference's shapes in lqmpc_bounds.hip, compiled at run time for
// the others (lqmpc_jit.hip).
#pragma once
#include "lqmpc_bounds.h"
#include "lqmpc_r16_setup.h"

namespace lqmpc {

constexpr int BOUNDS_REC = 10;     // doubles per instance between the two kernels: gamma rho_gamma fA fB nPhi nK eps rho_cl status spare

// ---------------- square matrices of TX x TX tiles as register matrices ----------------
template <int TX> struct Sq { double t[TX][TX]; };

template <int TX>
__device__ __forceinline__ Sq<TX> mmTN(const Sq<TX> &X, const Sq<TX> &Y)                      // X'Y
{
    Sq<TX> O;
#pragma unroll
    for (int a = 0; a < TX; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) v = mm4(X.t[k][a], Y.t[k][b], v);
            O.t[a][b] = v;
        }
    return O;
}
template <int TX>
__device__ __forceinline__ Sq<TX> mmTN(const Sq<TX> &X, const Sq<TX> &Y, const Sq<TX> &C)    // X'Y + C
{
    Sq<TX> O;
#pragma unroll
    for (int a = 0; a < TX; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            double v = C.t[a][b];
#pragma unroll
            for (int k = 0; k < TX; ++k) v = mm4(X.t[k][a], Y.t[k][b], v);
            O.t[a][b] = v;
        }
    return O;
}
template <int TX>
__device__ __forceinline__ Sq<TX> transpose(const Sq<TX> &X, double I4)                        // (a)' I = a', tile by tile
{
    Sq<TX> O;
#pragma unroll
    for (int a = 0; a < TX; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) O.t[a][b] = mm4(X.t[b][a], I4);
    return O;
}
template <int TX>
__device__ __forceinline__ Sq<TX> symmetrised(const Sq<TX> &X, double I4)
{
    const Sq<TX> Xt = transpose(X, I4);
    Sq<TX> O;
#pragma unroll
    for (int a = 0; a < TX; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) O.t[a][b] = 0.5 * (X.t[a][b] + Xt.t[a][b]);
    return O;
}
// sum of all entries of a register matrix, in every lane of its block: (x' 1)' 1
__device__ __forceinline__ double total4(double x, double ones) { return mm4(mm4(x, ones), ones); }
template <int TX>
__device__ __forceinline__ double frob2(const Sq<TX> &X, double ones)
{
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < TX; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) s = __builtin_fma(X.t[a][b], X.t[a][b], s);
    return total4(s, ones);
}
// inverse of a symmetric positive definite matrix whose padding (rows / columns beyond its order) is the identity; NaN if not SPD
template <int TX>
__device__ __forceinline__ Sq<TX> inv_spd(const Sq<TX> &M, int r, int c)
{
    Sq<TX> O;
    if constexpr (TX == 1) O.t[0][0] = small_inverse<4>(M.t[0][0], r, c);
    else {
        // [E F; F' H]^-1 by 4 x 4 tiles, S = H - F'E^-1 F
        const double Ei = small_inverse<4>(M.t[0][0], r, c);
        const double F = M.t[0][1];
        const double EiF = mm4(Ei, F);
        const double Si = small_inverse<4>(mm4(-F, EiF, M.t[1][1]), r, c);
        const double FtEi = mm4(F, Ei);
        const double X = mm4(FtEi, Si), Xt = mm4(Si, FtEi);
        O.t[0][0] = mm4(Xt, FtEi, Ei); O.t[0][1] = -X; O.t[1][0] = -Xt; O.t[1][1] = Si;
    }
    return O;
}
// log of the spectral radius of Y (Yt = Y'): |Y^(2^J)|_F^(1/2^J) by repeated squaring.  The iterate is kept in range by powers of
// two only -- Z_{j+1} = (Z_j 2^-e_j)^2 gives log rho = ln 2 sum_j e_j 2^-j + 2^-J log |Z_J| for ANY integers e_j -- so an iteration
// costs an exponent extraction and a few v_ldexp instead of a square root, a logarithm and a division (they were nine tenths of the
// kernel); a symmetric matrix cannot collapse under squaring (|Z^2|_F >= |Z|_F^2 / sqrt n), so it is rescaled every fourth
// iteration only.  -1e300 for a nilpotent matrix.
template <int TX>
__device__ __forceinline__ double log_rho(Sq<TX> Y, Sq<TX> Yt, bool symmetric, double ones)
{
    constexpr int J = 56;
    double acc = 0.0, wgt = 1.0;
    bool dead = false;
#pragma unroll 1
    for (int j = 0; j < J; ++j) {
        if (!symmetric || (j & 3) == 0) {
            const double s2 = frob2(Y, ones);
            dead = dead || !(s2 > 0.0);
            const int e = __builtin_amdgcn_frexp_exp(s2) >> 1;       // |Y|_F in [2^(e-1), 2^(e+1))
            acc = __builtin_fma(wgt, (double)e, acc);
#pragma unroll
            for (int a = 0; a < TX; ++a)
#pragma unroll
                for (int b = 0; b < TX; ++b) { Y.t[a][b] = __builtin_amdgcn_ldexp(Y.t[a][b], -e); Yt.t[a][b] = __builtin_amdgcn_ldexp(Yt.t[a][b], -e); }
        }
        wgt *= 0.5;
        const Sq<TX> Y2 = mmTN(Yt, Y);                               // Y Y
        if (symmetric) { Y = Y2; Yt = Y2; }
        else { const Sq<TX> Y2t = mmTN(Y, Yt); Y = Y2; Yt = Y2t; }
    }
    const double s2 = frob2(Y, ones);
    dead = dead || !(s2 > 0.0);
    return dead ? -1e300 : __builtin_fma(0.5 * wgt, log(dead ? 1.0 : s2), 0.6931471805599453 * acc);
}

// ---------------- kernel 1: the n_x x n_x work, four instances per wavefront ----------------
template <int NX, int NU, int N>
__device__ __forceinline__ void bounds_small(const BoundsParams &p)
{
    constexpr int TX = (NX + 3) / 4;
    static_assert(NX <= 8 && NU <= 4, "register matrices of at most 2 x 2 tiles; one tile of inputs");
    const int lane = threadIdx.x, r = lane >> 4, c = lane & 3, g = (lane >> 2) & 3;
    const long long b_raw = p.b0 + (long long)blockIdx.x * 4 + g;
    const bool valid = b_raw < p.b1;
    const long long b = valid ? b_raw : p.b1 - 1, Bsz = p.Bsz;
    const double *sh = p.sh;
    const double I4 = (r == c) ? 1.0 : 0.0, ones = 1.0;
    auto ldA = [&](int a, int k) -> double { const bool v = a < NX && k < NX; const double x = p.A[(long long)((v ? a : 0) * NX + (v ? k : 0)) * Bsz + b]; return v ? x : 0.0; };
    auto ldB = [&](int a, int k) -> double { const bool v = a < NX && k < NU; const double x = p.B[(long long)((v ? a : 0) * NU + (v ? k : 0)) * Bsz + b]; return v ? x : 0.0; };
    auto ldS = [&](int o, int a, int k, int dim, double pad) -> double {
        const bool v = a < dim && k < dim;
        const double x = sh[o + (v ? a * dim + k : 0)];
        return v ? x : ((a == k) ? pad : 0.0);
    };
    Sq<TX> A, At, H, G, Qi;
    double Bp[TX], Bt[TX];                                            // B by row tiles; (B tile)' as register matrices
#pragma unroll
    for (int a = 0; a < TX; ++a) {
#pragma unroll
        for (int bb = 0; bb < TX; ++bb) {
            A.t[a][bb] = ldA(4 * a + r, 4 * bb + c); At.t[a][bb] = ldA(4 * bb + c, 4 * a + r);
            H.t[a][bb] = ldS(p.oQ, 4 * a + r, 4 * bb + c, NX, 1.0);                       // Q, identity on the padding: it gets inverted
            Qi.t[a][bb] = ldS(p.oQinv, 4 * a + r, 4 * bb + c, NX, 0.0);
        }
        Bp[a] = ldB(4 * a + r, c); Bt[a] = ldB(4 * a + c, r);
    }
    const double Rp = ldS(p.oR, r, c, NU, 1.0), Rinv = ldS(p.oRinv, r, c, NU, 0.0);
    // G0 = B R^-1 B'
    {
        double RBt[TX];
#pragma unroll
        for (int a = 0; a < TX; ++a) RBt[a] = mm4(Rinv, Bt[a]);       // R^-1 B_a'
#pragma unroll
        for (int a = 0; a < TX; ++a)
#pragma unroll
            for (int bb = 0; bb < TX; ++bb) G.t[a][bb] = mm4(RBt[a], Bt[bb]);
        G = symmetrised(G, I4);
    }
    // ---- dlqr: P_inf by the doubling iteration (lqmpc_bounds.hip has the scalar form; control.dlqr in the reference) ----
    Sq<TX> Ak = A, Akt = At;
    bool conv = false, bad = false;
    int status = 0;
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        const Sq<TX> Hi = inv_spd(H, r, c);
        Sq<TX> HG;
#pragma unroll
        for (int a = 0; a < TX; ++a)
#pragma unroll
            for (int bb = 0; bb < TX; ++bb) HG.t[a][bb] = Hi.t[a][bb] + G.t[a][bb];
        const Sq<TX> S = inv_spd(symmetrised(HG, I4), r, c);
        const Sq<TX> W1 = mmTN(Hi, S);                               // H^-1 S
        const Sq<TX> Vt = mmTN(W1, Akt);                             // V' = S H^-1 Ak'
        const Sq<TX> GVt = mmTN(G, Vt);                              // (V G)'
        const Sq<TX> Gn = symmetrised(mmTN(GVt, Akt, G), I4);        // G + V G Ak'
        const Sq<TX> SA = mmTN(S, Ak);
        const Sq<TX> Hn = mmTN(Ak, SA, H);                           // H + Ak' S Ak
        Sq<TX> D;
#pragma unroll
        for (int a = 0; a < TX; ++a)
#pragma unroll
            for (int bb = 0; bb < TX; ++bb) D.t[a][bb] = Hn.t[a][bb] - H.t[a][bb];
        const double dn = frob2(D, ones), hn = frob2(Hn, ones);
        const Sq<TX> Akn = mmTN(Vt, Ak), Aktn = mmTN(Ak, Vt);
        if (!conv) { Ak = Akn; Akt = Aktn; G = Gn; H = symmetrised(Hn, I4); }       // (a converged instance keeps its fixed point while the others go on)
        conv = conv || (dn <= 1e-34 * hn);
        bad = bad || !(hn < 1e300);
        if (__ballot(!(conv || bad)) == 0ull) break;
    }
    if (bad) status = 2; else if (!conv) status = 1;
    // K = (R + B'PB)^-1 B'PA  (u = -K x)
    double K[TX], Kt[TX];
    {
        double SB[TX], F[TX];
        double Re = Rp;
#pragma unroll
        for (int a = 0; a < TX; ++a) {
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) w = mm4(H.t[k][a], Bp[k], w);
            SB[a] = w;
        }
#pragma unroll
        for (int k = 0; k < TX; ++k) Re = mm4(Bp[k], SB[k], Re);
        const Sq<TX> SA = mmTN(H, A);
#pragma unroll
        for (int bb = 0; bb < TX; ++bb) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < TX; ++k) v = mm4(Bp[k], SA.t[k][bb], v);
            F[bb] = v;
        }
        const double R0 = small_inverse<4>(Re, r, c);                 // (R padded with ones: a full 4 x 4 SPD block)
        const double Rm = (r < NU && c < NU) ? R0 : 0.0;
#pragma unroll
        for (int bb = 0; bb < TX; ++bb) { K[bb] = mm4(Rm, F[bb]); Kt[bb] = mm4(K[bb], I4); }
        if (!(total4(fabs(Rm), ones) < 1e300)) status = 2;
    }
    if (valid) {
#pragma unroll
        for (int bb = 0; bb < TX; ++bb) {
            if (p.K && r < NU && 4 * bb + c < NX) p.K[(long long)(r * NX + 4 * bb + c) * Bsz + b] = K[bb];
            if (p.Pinf) {
#pragma unroll
                for (int a = 0; a < TX; ++a)
                    if (4 * a + r < NX && 4 * bb + c < NX) p.Pinf[(long long)((4 * a + r) * NX + 4 * bb + c) * Bsz + b] = H.t[a][bb];
            }
        }
    }
    // ---- local radius: 1 / max_k (K Q^-1 K')_kk / bound_k^2  (utils.py:548-564, box rows of F_u; the callers pass -K) ----
    double worst = 0.0;
    {
        double X = 0.0;                                               // K Q^-1 K' (n_u x n_u)
#pragma unroll
        for (int a = 0; a < TX; ++a) {
            double QK = 0.0;                                          // (Q^-1 K')_a
#pragma unroll
            for (int k = 0; k < TX; ++k) QK = mm4(Qi.t[k][a], Kt[k], QK);
            X = mm4(Kt[a], QK, X);
        }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const double dk = total4((r == k && c == k) ? X : 0.0, ones);
            const double ub = sh[p.oub + k], lb = sh[p.olb + k];
            worst = fmax(worst, fmax(dk / (ub * ub), dk / (lb * lb)));
        }
    }
    const double eps = 1.0 / worst;
    // ---- |K|_2, rho(A - BK), |A|_2, |B|_2, |Phi|_2 ----
    Sq<TX> KtK, Acl, Aclt, AtA, Phi;
    double BtB = 0.0;
#pragma unroll
    for (int a = 0; a < TX; ++a) {
        BtB = mm4(Bp[a], Bp[a], BtB);
#pragma unroll
        for (int bb = 0; bb < TX; ++bb) {
            KtK.t[a][bb] = mm4(K[a], K[bb]);
            Acl.t[a][bb] = mm4(-Bt[a], K[bb], A.t[a][bb]);
            Aclt.t[a][bb] = mm4(-K[a], Bt[bb], At.t[a][bb]);
            Phi.t[a][bb] = (a == bb) ? I4 : 0.0;
        }
    }
    AtA = mmTN(A, A);
    {
        Sq<TX> Id = Phi;
#pragma unroll
        for (int a = 0; a < TX; ++a)
#pragma unroll
            for (int bb = 0; bb < TX; ++bb) { const int ra = 4 * a + r, cb = 4 * bb + c; Id.t[a][bb] = (ra == cb && ra < NX) ? 1.0 : 0.0; }
        Phi = Id;
#pragma unroll 1
        for (int k = 1; k <= N; ++k) Phi = mmTN(A, mmTN(Phi, A), Id);  // sum_{k=0..N} (A^k)'A^k = I + A'( .. )A
    }
    Sq<1> BB; BB.t[0][0] = BtB;
    const double nK = exp(0.5 * log_rho(KtK, KtK, true, ones));
    const double rho_cl = exp(log_rho(Acl, Aclt, false, ones));
    const double fA = exp(0.5 * log_rho(AtA, AtA, true, ones));
    const double fB = exp(0.5 * log_rho(BB, BB, true, ones));
    const double nPhi = exp(0.5 * log_rho(Phi, Phi, true, ones));
    const double qmax = sh[p.osc + 0], qmin = sh[p.osc + 1], rmax = sh[p.osc + 2];
    const double rho_K = (rho_cl + 0.4) * (rho_cl + 0.4);                // utils.py:358
    const double C_star = (1.0 + rmax * nK * nK / qmin) * fmax(1.0, qmax / qmin * 1.21);   // utils.py:364-368
    const double gamma = C_star / (1.0 - rho_K);
    const double rho_gamma = (gamma - 1.0) / gamma;
    if (valid && r == 0 && c == 0) {
        double *o = p.rec;
        o[0 * Bsz + b] = gamma; o[1 * Bsz + b] = rho_gamma; o[2 * Bsz + b] = fA; o[3 * Bsz + b] = fB; o[4 * Bsz + b] = nPhi;
        o[5 * Bsz + b] = nK; o[6 * Bsz + b] = eps; o[7 * Bsz + b] = rho_cl; o[8 * Bsz + b] = (double)status;
        if (p.eps) p.eps[b] = eps;
    }
}

// ---------------- kernel 2: the N n_u x N n_u work, LPI lanes per instance ----------------
template <int NX, int NU, int N, int LPI>
struct BigT {
    static constexpr int n = N * NU, RB = (n + LPI - 1) / LPI, LDW = n + 1, IPW = 64 / LPI, VEC = LPI * RB;
    // LDS per instance (doubles): M (n rows of stride n + 1) | v | w | d | e2 | Md (N nx nu) | dummy
    static constexpr int oM = 0, oV = n * LDW, oW = oV + VEC, oDg = oW + VEC, oE = oDg + VEC, oMd = oE + VEC, oD = oMd + N * NX * NU + 2;
    static constexpr int INST = oD + 2;
};

template <int LPI>
__device__ __forceinline__ double group_sum(double x)
{
    x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
    if constexpr (LPI == 64) { x += __shfl_xor(x, 16); x += __shfl_xor(x, 32); }
    return x;
}

// Householder tridiagonalisation of the symmetric n x n matrix at M (LDS, rows of stride LDW, destroyed), the lanes of the instance
// working on their rows (row i + LPI s); d -> dL[0..n), squared off-diagonal -> e2L[0..n-1)
template <int n, int LPI, int RB, int LDW>
__device__ __forceinline__ void tridiag_rows(wg::ldsd *M, wg::ldsd *vL, wg::ldsd *wL, wg::ldsd *dL, wg::ldsd *e2L, int i)
{
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        double x[RB], nrm2 = 0.0;
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int row = i + LPI * s;
            const double t = M[(row < n ? row : 0) * LDW + k];
            x[s] = (row > k && row < n) ? t : 0.0;
            nrm2 = __builtin_fma(x[s], x[s], nrm2);
        }
        nrm2 = group_sum<LPI>(nrm2);
        const double x0 = M[(k + 1) * LDW + k], dk = M[k * LDW + k];
        const double tail2 = nrm2 - x0 * x0;
        const bool act = tail2 > 0.0;                                 // (else: nothing below the subdiagonal in this column)
        const double alpha = (x0 > 0.0) ? -sqrt(nrm2) : sqrt(nrm2);
        const double r2 = 0.5 * (nrm2 - x0 * alpha), rinv = act ? 1.0 / (2.0 * sqrt(r2)) : 0.0;   // H = I - 2 v v', |v| = 1
        double v[RB];
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int row = i + LPI * s;
            v[s] = (x[s] - (row == k + 1 ? alpha : 0.0)) * rinv;
            v[s] = (row > k && row < n) ? v[s] : 0.0;
            vL[row] = v[s];
        }
        if (i == 0) { dL[k] = dk; const double ek = act ? alpha : x0; e2L[k] = ek * ek; }
        __syncthreads();
        double pr[RB], Kp = 0.0;                                      // p = A22 v, K = v'p
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int row = i + LPI * s, rr = row < n ? row : 0;
            double acc = 0.0;
#pragma unroll 4
            for (int cc = k + 1; cc < n; ++cc) acc = __builtin_fma(M[rr * LDW + cc], vL[cc], acc);
            pr[s] = (row > k && row < n) ? acc : 0.0;
            Kp = __builtin_fma(v[s], pr[s], Kp);
        }
        Kp = group_sum<LPI>(Kp);
#pragma unroll
        for (int s = 0; s < RB; ++s) wL[i + LPI * s] = __builtin_fma(-Kp, v[s], pr[s]);       // w = p - K v
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int row = i + LPI * s;
            if (row > k && row < n && act) {
                const double vr2 = 2.0 * v[s], wr2 = 2.0 * wL[row];
#pragma unroll 4
                for (int cc = k + 1; cc < n; ++cc) M[row * LDW + cc] = M[row * LDW + cc] - vr2 * wL[cc] - wr2 * vL[cc];
            }
        }
        __syncthreads();
    }
    if (i == 0) {
        if (n >= 2) {
            dL[n - 2] = M[(n - 2) * LDW + (n - 2)];
            const double ek = M[(n - 1) * LDW + (n - 2)];
            e2L[n - 2] = ek * ek;
        }
        dL[n - 1] = M[(n - 1) * LDW + (n - 1)];
    }
    __syncthreads();
}

// smallest and largest eigenvalue of the symmetric tridiagonal matrix (dL, e2L = squared off-diagonal) by multisection on the Sturm
// count: every lane of the instance counts the eigenvalues below its own shift, for both searches in one sweep over (d, e^2)
template <int n, int LPI>
__device__ __forceinline__ void tridiag_extremes(const wg::ldsd *dL, const wg::ldsd *e2L, int i, int q, double &emin, double &emax)
{
    double lo = 1e308, hi = -1e308;                                  // Gershgorin
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        const double rad = (j > 0 ? sqrt(e2L[j - 1]) : 0.0) + (j + 1 < n ? sqrt(e2L[j]) : 0.0);
        lo = fmin(lo, dL[j] - rad); hi = fmax(hi, dL[j] + rad);
    }
    const double span = fmax(hi - lo, 1e-300), tiny = 1e-300 + 2.3e-16 * fmax(fabs(lo), fabs(hi));
    double a0 = lo - 1e-3 * span, b0 = hi + 1e-3 * span, a1 = a0, b1 = b0;      // brackets: count(a0) < 1 <= count(b0), count(a1) < n <= count(b1)
    constexpr int PASSES = (LPI == 16) ? 15 : 11;                     // (LPI + 1)^PASSES > 2^60
    auto guard = [&](double x) { return (fabs(x) < tiny) ? ((x < 0.0) ? -tiny : tiny) : x; };
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const double st0 = (b0 - a0) / (double)(LPI + 1), st1 = (b1 - a1) / (double)(LPI + 1);
        const double x0 = __builtin_fma(st0, (double)(i + 1), a0), x1 = __builtin_fma(st1, (double)(i + 1), a1);
        double q0 = dL[0] - x0, q1 = dL[0] - x1;
        int c0 = q0 < 0.0, c1 = q1 < 0.0;
#pragma unroll 2
        for (int j = 1; j < n; ++j) {
            const double dj = dL[j], ej = e2L[j - 1];
            q0 = __builtin_fma(-ej, frcp(guard(q0)), dj - x0);       // (the reciprocal by Newton steps: an IEEE division is three times the instructions)
            q1 = __builtin_fma(-ej, frcp(guard(q1)), dj - x1);
            c0 += q0 < 0.0; c1 += q1 < 0.0;
        }
        // lanes 0 .. first-1 count below the target, lanes first .. reach it (the count is monotone in the shift)
        unsigned long long m0 = __ballot(c0 >= 1), m1 = __ballot(c1 >= n);
        if constexpr (LPI == 16) { m0 = (m0 >> (16 * q)) & 0xFFFFull; m1 = (m1 >> (16 * q)) & 0xFFFFull; }
        const int f0 = m0 ? __ffsll((long long)m0) - 1 : LPI, f1 = m1 ? __ffsll((long long)m1) - 1 : LPI;   // LPI: the root is in the last piece
        const double nb0 = (f0 < LPI) ? __builtin_fma(st0, (double)(f0 + 1), a0) : b0, nb1 = (f1 < LPI) ? __builtin_fma(st1, (double)(f1 + 1), a1) : b1;
        a0 = __builtin_fma(st0, (double)f0, a0); b0 = nb0;
        a1 = __builtin_fma(st1, (double)f1, a1); b1 = nb1;
    }
    emin = 0.5 * (a0 + b0); emax = 0.5 * (a1 + b1);
}

__device__ __forceinline__ double bc_gx(int power, int i, double eA, double fA)           // utils.py:78-95
{
    const double t = pow(eA + fA, (double)i) - pow(fA, (double)i);
    return power == 1 ? t : t * t;
}
__device__ __forceinline__ double bc_gu(int power, int i, double eA, double fA, double eB, double fB)   // utils.py:98-117
{
    const double t = (eB + fB) * bc_gx(1, i, eA, fA) + eB * pow(fA, (double)i);
    return power == 1 ? t : t * t;
}

template <int NX, int NU, int N, int LPI>
__device__ __forceinline__ void bounds_big(const BoundsParams &p, double *lds_raw)
{
    using C = BigT<NX, NU, N, LPI>;
    constexpr int n = C::n, RB = C::RB, LDW = C::LDW;
    const int lane = threadIdx.x, q = lane / LPI, i = lane % LPI;
    wg::ldsd *L = (wg::ldsd *)lds_raw + q * C::INST;
    wg::ldsd *M = L + C::oM, *vL = L + C::oV, *wL = L + C::oW, *dL = L + C::oDg, *e2L = L + C::oE, *MdL = L + C::oMd;
    const long long slot0 = p.b0 + (long long)blockIdx.x * C::IPW, Bsz = p.Bsz;
    const long long b_raw = slot0 + q;
    const bool valid = b_raw < p.b1;
    const long long b = valid ? b_raw : p.b1 - 1;
    const double *sh = p.sh;
    // Gamma'Gamma = the condensed Hessian of (A, B) with unit state weights and no input weight: P/2 of r16_build_P
    long long bg = b;
    wg::ldsd *Lg = L;
    if constexpr (LPI == 16) {
        const int gq = (lane >> 2) & 3;
        const long long sraw = slot0 + gq;
        bg = sraw < p.b1 ? sraw : p.b1 - 1;
        Lg = (wg::ldsd *)lds_raw + gq * C::INST;
    }
    const SetupArgs unit{nullptr, p.A, p.B, p.sh, Bsz, p.oI, p.oI, p.oZ};
    r16_build_P<NX, NU, N, LPI, false>(unit, bg, Lg, C::oM, C::oD);
    tridiag_rows<n, LPI, RB, LDW>(M, vL, wL, dL, e2L, i);
    double eminE, emaxE;
    tridiag_extremes<n, LPI>(dL, e2L, i, q, eminE, emaxE);
    const double nG = sqrt(fmax(0.5 * emaxE, 0.0));
    double min_H;
    if (p.q_scalar && p.r_scalar) {
        min_H = p.rs + p.qs * 0.5 * eminE;
    } else {
        if (p.q_scalar) {
            r16_build_P<NX, NU, N, LPI, false>(unit, bg, Lg, C::oM, C::oD);
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const int row = i + LPI * s;
                if (row < n) {
#pragma unroll 1
                    for (int j = 0; j < n; ++j) {                      // hat H = kron(R, I_N) + q Gamma'Gamma, index pairing of utils.py:316
                        const double kr = (row % N == j % N) ? sh[p.oR + (row / N) * NU + (j / N)] : 0.0;
                        M[row * LDW + j] = __builtin_fma(0.5 * p.qs, M[row * LDW + j], kr);
                    }
                }
            }
        } else {
            // the table A^d B (row-major (N nx) x nu), stage by stage: lane (a, k) owns entry (a, k)
            const bool el = i < NX * NU;
            const int ea = el ? i / NU : 0, ek = el ? i % NU : 0;
            double Ar[NX];
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) Ar[cc] = p.A[(long long)(ea * NX + cc) * Bsz + b];
            if (el) MdL[ea * NU + ek] = p.B[(long long)(ea * NU + ek) * Bsz + b];
#pragma unroll 1
            for (int d = 1; d < N; ++d) {
                __syncthreads();
                double t = 0.0;
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) t = __builtin_fma(Ar[cc], MdL[((d - 1) * NX + cc) * NU + ek], t);
                if (el) MdL[(d * NX + ea) * NU + ek] = t;
            }
            __syncthreads();
            // Gamma (the reference's, with a leading zero block row): block (r, c) = M_{r-1-c} for r > c, r = 0..N, c = 0..N-1
            auto gam = [&](int rho, int col) -> double {
                const int rr = rho / NX, a = rho % NX, cb = col / NU, k = col % NU;
                return rr > cb ? MdL[((rr - 1 - cb) * NX + a) * NU + k] : 0.0;
            };
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const int row = i + LPI * s;
                if (row < n) {
#pragma unroll 1
                    for (int j = 0; j < n; ++j) {
                        double t = (row % N == j % N) ? sh[p.oR + (row / N) * NU + (j / N)] : 0.0;
#pragma unroll 1
                        for (int r1 = 0; r1 < (N + 1) * NX; ++r1) {
                            const double gi = gam(r1, row);
                            const int qa = r1 / (N + 1), st = r1 % (N + 1);
                            double u = 0.0;
#pragma unroll
                            for (int cc = 0; cc < NX; ++cc) u = __builtin_fma(sh[p.oQ + qa * NX + cc], gam(cc * (N + 1) + st, j), u);
                            t = __builtin_fma(gi, u, t);
                        }
                        M[row * LDW + j] = t;
                    }
                }
            }
            __syncthreads();
            // the product is symmetric when Q is; mirror the lower triangle (as the workspace kernel does)
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                const int row = i + LPI * s;
                if (row < n)
#pragma unroll 1
                    for (int j = row + 1; j < n; ++j) M[row * LDW + j] = M[j * LDW + row];
            }
        }
        __syncthreads();
        tridiag_rows<n, LPI, RB, LDW>(M, vL, wL, dL, e2L, i);
        double emaxH;
        tridiag_extremes<n, LPI>(dL, e2L, i, q, min_H, emaxH);
    }
    // ---- the scalar formulas (every lane of the instance computes them; lane 0 writes) ----
    const double *rec = p.rec;
    const double gamma = rec[0 * Bsz + b], rho_gamma = rec[1 * Bsz + b], fA = rec[2 * Bsz + b], fB = rec[3 * Bsz + b];
    const double nPhi = rec[4 * Bsz + b], nK = rec[5 * Bsz + b], eps = rec[6 * Bsz + b], rho_cl = rec[7 * Bsz + b];
    int status = (int)rec[8 * Bsz + b];
    const double qmax = sh[p.osc + 0], qmin = sh[p.osc + 1], rmax = sh[p.osc + 2], rmin = sh[p.osc + 3];
    const double V_expert = sh[p.osc + 4], bar_u = sh[p.osc + 5], bar_du = sh[p.osc + 6];
    // energy_decreasing: xi, eta (utils_class.py:344-373)
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
    // energy_bound: alpha, beta (utils_class.py:308-342)
    double nx2 = 0.0;
    for (int a = 0; a < NX; ++a) nx2 = __builtin_fma(sh[p.ox + a], sh[p.ox + a], nx2);
    double s_in = 0.0, s_out = 0.0;
    for (int k = 0; k <= N; ++k) {                                       // utils.py:296-302
        s_out += (s_in + bc_gx(2, k, eA, fA)) * (nx2 + k * bar_u);
        s_in += bc_gu(2, k, eA, fA, eB, fB);
    }
    const double E_psi = qmax * s_out;
    double bar_gx = 0.0, bar_gu = 0.0, run = 0.0;
    for (int k = 0; k < N; ++k) {                                        // utils.py:186-223
        bar_gx += bc_gx(1, k + 1, eA, fA);
        run += bc_gu(1, k, eA, fA, eB, fB);
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
    if (valid && i == 0) {
        if (p.alpha) p.alpha[b] = alpha;
        if (p.beta) p.beta[b] = beta;
        if (p.xi) p.xi[b] = xi;
        if (p.eta) p.eta[b] = eta;
        if (p.bound) p.bound[b] = (alpha * V_expert + beta) / (1.0 - xi - eta);                // utils_class.py:858-859
        if (p.aux) {
            double *a = p.aux;
            a[0 * Bsz + b] = gamma; a[1 * Bsz + b] = rho_cl; a[2 * Bsz + b] = fA; a[3 * Bsz + b] = fB;
            a[4 * Bsz + b] = nG; a[5 * Bsz + b] = nPhi; a[6 * Bsz + b] = min_H; a[7 * Bsz + b] = nK;
        }
        if (p.status) p.status[b] = status;
    }
}

}  // namespace lqmpc
