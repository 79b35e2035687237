"""Dev experiment (CPU): primal-dual active-set iterations over a closed-loop C3 rollout for different first guesses at every step:
(a) cold: violated rows of v_unc; warm: the previous face shifted by one stage (rounds 1-2)
(b) the saturated time-varying LQR roll-forward at every step
(c) cold: the roll; warm: the shifted face (what the kernels do now).
Measured (600 / 200 instances): default mix 0.211 / 0.220 / 0.185 iterations per QP-step, hard mix 1.181 / 1.438 / 1.118."""
import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tools/proto')
from lq_mpc_amd import synth
from cold_start_guess import pdas


def run(mix, K, T=30, cfg=3):
    b = synth.make_batch(cfg, Bsz=K, mix=mix)
    nx, nu, N = b['nx'], b['nu'], b['N']; n = N * nu
    h = np.tile(0.5 * (b['ub'] - b['lb']), N); Q, R, P = b['Q'], b['R'], b['P']
    tot = {k: np.zeros(T) for k in 'abc'}; busy = np.zeros(T)
    for k in range(K):
        A, B = b['A'][:, :, k], b['B'][:, :, k]
        H, F = synth.condense_np(A, B, Q, R, P, N)
        S = P.copy(); Ks = [None] * N
        for j in range(N - 1, -1, -1):
            Kj = np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A); S = Q + A.T @ S @ (A - B @ Kj); Ks[j] = Kj

        def roll(x):
            Lb = np.zeros(n, bool); Ub = np.zeros(n, bool)
            for j in range(N):
                u = -Ks[j] @ x
                Lb[j * nu:(j + 1) * nu] = u < -h[:nu]; Ub[j * nu:(j + 1) * nu] = u > h[:nu]
                x = A @ x + B @ np.clip(u, -h[:nu], h[:nu])
            return Lb, Ub
        x = b['x0'][:, k].copy(); pL = pU = None
        for t in range(T):
            g = F @ x; vunc = np.linalg.solve(H, -g)
            cl, cu = vunc < -h, vunc > h
            if (cl | cu).any():
                busy[t] += 1
                warm = pL is not None and (pL | pU).any()
                sL, sU = (np.concatenate([pL[nu:], pL[-nu:]]), np.concatenate([pU[nu:], pU[-nu:]])) if warm else (cl, cu)
                ia, v, L, U = pdas(H, g, h, sL, sU, full=True)
                ib = pdas(H, g, h, *roll(x))[0]
                tot['a'][t] += ia; tot['b'][t] += ib; tot['c'][t] += ia if warm else ib
                pL, pU = L, U
            else:
                v = vunc; pL = np.zeros(n, bool); pU = np.zeros(n, bool)
            x = b['A_true'] @ x + b['B_true'] @ np.clip(v[:nu], -h[:nu], h[:nu])
    print(mix, 'K', K, ' share of constrained QP-steps %.3f' % (busy.sum() / (K * T)))
    for k, nm in (('a', 'violated rows + shift'), ('b', 'roll at every step'), ('c', 'roll cold + shift')):
        print('   %-22s iterations/QP-step %.3f   step 0: %.2f   per constrained later step %.2f'
              % (nm, tot[k].sum() / (K * T), tot[k][0] / K, tot[k][1:].sum() / max(busy[1:].sum(), 1)))


if __name__ == '__main__':
    run('default', int(sys.argv[1]) if len(sys.argv) > 1 else 600)
    run('hard', int(sys.argv[2]) if len(sys.argv) > 2 else 200)
