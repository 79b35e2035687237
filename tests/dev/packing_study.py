"""Dev tool: how many wave-steps (16 systems in lockstep) need the interior-point loop under different instance orders."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import synth
from oracle import oracle as orc
Bsz, T = 4096, 30
b = synth.make_batch(3, Bsz=Bsz)
N = b['N']
out = orc.rollout_batch(T, N, b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'], b['x0'], b['A_true'], b['B_true'], want_traj=True)
X = out['X']
con = np.zeros((T, Bsz), bool)
rho = np.zeros(Bsz)
for i in range(Bsz):
    H, F = orc.condense(b['A'][:, :, i], b['B'][:, :, i], b['Q'], b['R'], b['P'], N)
    G = -np.linalg.solve(H, F)            # v_unc = G x
    V = G @ X[:, :T, i]                   # (n, T)
    con[:, i] = (np.abs(V) > 0.1).any(axis=0)
    rho[i] = np.abs(V[:, 0]).max() / 0.1
print('constrained fraction of (step,system):', con.mean())
def wave_frac(order):
    c = con[:, order].reshape(T, Bsz // 16, 16)
    return c.any(axis=2).mean(), c.any(axis=2).sum(axis=0)
for name, order in (('natural', np.arange(Bsz)), ('sorted by rho(t=0)', np.argsort(-rho)), ('sorted by #constrained steps (ideal)', np.argsort(-con.sum(0), kind='stable')),
                    ('sorted by |x0|', np.argsort(-np.linalg.norm(b['x0'], axis=0)))):
    f, per_wave = wave_frac(order)
    print(f'{name:40s}: wave-steps needing IPM {f:.3f}; per-wave IPM steps: max {per_wave.max()} mean {per_wave.mean():.1f}')

# ---- better keys: simulate the clipped unconstrained law and count predicted constrained steps ----
def predicted(clip):
    cnt = np.zeros(Bsz); first = np.zeros(Bsz); area = np.zeros(Bsz)
    for i in range(Bsz):
        H, F = orc.condense(b['A'][:, :, i], b['B'][:, :, i], b['Q'], b['R'], b['P'], N)
        G = -np.linalg.solve(H, F)
        x = b['x0'][:, i].copy(); c = 0; a = 0.0
        for t in range(T):
            v = G @ x
            rho_t = np.abs(v).max() / 0.1
            if rho_t > 1: c += 1; a += rho_t - 1
            u = v[:2]
            if clip: u = np.clip(u, -0.1, 0.1)
            x = b['A_true'] @ x + b['B_true'] @ u
        cnt[i] = c; area[i] = a
    return cnt, area
for clip in (False, True):
    cnt, area = predicted(clip)
    for name, key in ((f'pred count (clip={clip}) + rho0 tiebreak', cnt + 1e-3 * np.minimum(rho, 900)), (f'pred area (clip={clip})', area)):
        f, per_wave = wave_frac(np.argsort(-key, kind='stable'))
        print(f'{name:40s}: wave-steps needing IPM {f:.3f}; per-wave IPM steps: max {per_wave.max()} mean {per_wave.mean():.1f}')

# ---- cheap keys that need no condensing: one-stage-lookahead gradient, and N-stage free-response energy ----
k1 = np.zeros(Bsz); k2 = np.zeros(Bsz)
Q, R = b['Q'], b['R']
for i in range(Bsz):
    A, Bm, x = b['A'][:, :, i], b['B'][:, :, i], b['x0'][:, i]
    g = Bm.T @ Q @ A @ x
    d = np.diag(Bm.T @ Q @ Bm + R)
    k1[i] = np.max(np.abs(g) / (d * 0.1))
    # free response over the horizon weighted by B: sum_r |B' Q A^{r+1} x|
    xx = x.copy(); acc = 0.0
    for r in range(N):
        xx = A @ xx
        acc = max(acc, np.max(np.abs(Bm.T @ Q @ xx) / (d * 0.1)))
    k2[i] = acc
for name, key in (('one-stage gradient', k1), ('max over horizon of stage gradient', k2)):
    f, per_wave = wave_frac(np.argsort(-key, kind='stable'))
    print(f'{name:40s}: wave-steps needing IPM {f:.3f}; per-wave IPM steps: max {per_wave.max()} mean {per_wave.mean():.1f}')
