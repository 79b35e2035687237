"""Dev experiment (CPU): how well a per-instance key groups the instances of a C3 rollout into wavefronts of four, measured as the share
of wave-steps with at least one constrained instance (what a 16-lane-row wavefront pays for).  Flags come from the oracle's
trajectories.  Keys: the probe's (largest stage gradient of the free response), the true number of constrained steps (the floor), and
candidates."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import synth
from oracle import oracle as orc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mix = sys.argv[2] if len(sys.argv) > 2 else 'default'
b = synth.make_batch(3, Bsz=K, mix=mix)
nx, nu, N, T = b['nx'], b['nu'], b['N'], 30
r = orc.rollout_batch(T, N, b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'], b['x0'], b['A_true'], b['B_true'], want_traj=True)
X = r['X']                                                     # (nx, T+1, K)
h = 0.5 * (b['ub'] - b['lb'])
hh = np.tile(h, N)[:, None]
flags = np.zeros((T, K), bool)
key_probe = np.zeros(K); key_probe_t = np.zeros((T, K))
Q, R = b['Q'], b['R']
for i in range(K):
    A, B = b['A'][:, :, i], b['B'][:, :, i]
    H, F = synth.condense_np(A, B, Q, R, b['P'], N)
    v = -np.linalg.solve(H, F @ X[:, :-1, i])
    flags[:, i] = (np.abs(v) > hh).any(axis=0)
    QB = Q @ B
    dinv = 1.0 / ((np.einsum('ik,ik->k', B, QB) + np.diag(R)) * h)
    x = X[:, :-1, i].copy()                                    # the probe's key at every state of the trajectory (column t)
    kt = np.zeros(T)
    for rr in range(N):
        x = A @ x
        kt = np.maximum(kt, (np.abs(QB.T @ x) * dinv[:, None]).max(axis=0))
    key_probe_t[:, i] = kt
    key_probe[i] = kt[0]
nbusy = flags.sum(axis=0)

def wave_share(order):
    f = flags[:, order]
    f = f[:, :K - K % 4].reshape(T, -1, 4)
    return f.any(axis=2).mean()

def bucket(key):                                               # the probe's 16 buckets per binade over [2^-2, 2^30)
    kk = np.where(np.isfinite(key), key, 1e300)
    raw = np.floor(16 * (np.log2(np.maximum(kk, 1e-300)) + 2)).astype(int)
    return np.clip(raw, 0, 511)

print('C3 %s, %d instances: constrained QP-steps %.4f of all; instances never constrained %.3f' % (mix, K, flags.mean(), (nbusy == 0).mean()))
print('  natural order                         wave-steps with a constrained instance: %.4f' % wave_share(np.arange(K)))
print('  probe key, 16 buckets per binade                                               %.4f' % wave_share(np.argsort(-bucket(key_probe), kind='stable')))
print('  probe key, exact sort                                                          %.4f' % wave_share(np.argsort(-key_probe, kind='stable')))
print('  true number of constrained steps (floor)                                       %.4f' % wave_share(np.argsort(-nbusy, kind='stable')))
# candidates
x0 = b['x0']
for name, key in (
    ('|x0|', np.linalg.norm(x0, axis=0)),
    ('probe key at x1 (after the first input)', key_probe_t[1]),
    ('sum of log-keys along the first 3 states', np.log(np.maximum(key_probe_t[:3], 1e-30)).sum(axis=0)),
):
    print('  %-40s exact sort                             %.4f' % (name, wave_share(np.argsort(-key, kind='stable'))))
# how the number of constrained steps relates to the key: rank correlation
from scipy.stats import spearmanr
print('  Spearman(probe key, constrained steps) = %.3f' % spearmanr(key_probe, nbusy).correlation)
# a fitted predictor: steps until the key decays below 1/4 assuming geometric decay with the plant's spectral radius
rho = max(abs(np.linalg.eigvals(b['A_true'])))
pred = np.log(np.maximum(key_probe, 1e-30) / 0.25)
print('  plant spectral radius %.3f' % rho)

# ---- candidates built on the shared plant (A_true, B_true): the closed loop every instance actually runs in ----
from scipy.linalg import solve_discrete_are
At, Bt = b['A_true'], b['B_true']
S = solve_discrete_are(At, Bt, Q, R)
Kn = np.linalg.solve(R + Bt.T @ S @ Bt, Bt.T @ S @ At)
print('  value function of the nominal plant x0\'S x0, exact sort                          %.4f' % wave_share(np.argsort(-np.einsum('ik,ij,jk->k', x0, S, x0), kind='stable')))
for theta in (0.5, 0.7, 0.85, 1.0):
    x = x0.copy(); cnt = np.zeros(K); last = np.zeros(K)
    for t in range(T):
        u = -Kn @ x
        over = (np.abs(u) / h[:, None]).max(axis=0) > theta
        cnt += over; last = np.where(over, t + 1, last)
        x = At @ x + Bt @ np.clip(u, -h[:, None], h[:, None])
    print('  saturated nominal LQR roll, steps with |Kx|/h > %.2f: count %.4f   last such step %.4f   (Spearman with the truth %.3f)'
          % (theta, wave_share(np.argsort(-cnt, kind='stable')), wave_share(np.argsort(-last, kind='stable')), spearmanr(last, nbusy).correlation))
# the same with a fractional tie-break (the margin at the last step over the threshold)
x = x0.copy(); last = np.zeros(K); marg = np.zeros(K)
for t in range(T):
    u = -Kn @ x
    m = (np.abs(u) / h[:, None]).max(axis=0)
    over = m > 0.7
    last = np.where(over, t + 1, last); marg = np.where(over, m, marg)
    x = At @ x + Bt @ np.clip(u, -h[:, None], h[:, None])
print('  ... last step over 0.70, ties broken by the margin there                        %.4f' % wave_share(np.lexsort((-marg, -last))))

# ---- the key as the probe would compute it: bucket = 16 * (last step whose first input saturates) + 16ths of a margin ----
def fh_gain(A, B, Q, R, P, N):
    S = P.copy()
    for _ in range(N):
        Kk = np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A); S = Q + A.T @ S @ (A - B @ Kk)
    return Kk
for nm, Kg in (('DARE gain of the plant', Kn), ('first gain of the N-stage problem on the plant', fh_gain(At, Bt, Q, R, b['P'], N))):
    for Tsim in (30, 16):
        x = x0.copy(); last = np.zeros(K, int); marg = (np.abs(Kg @ x) / h[:, None]).max(axis=0)
        for t in range(Tsim):
            u = -Kg @ x
            m = (np.abs(u) / h[:, None]).max(axis=0)
            over = m > 1.0
            last = np.where(over, t + 1, last); marg = np.where(over, m, marg)
            x = At @ x + Bt @ np.clip(u, -h[:, None], h[:, None])
        g = np.where(last > 0, 1.0 - 1.0 / np.maximum(marg, 1.0), np.minimum(marg, 0.999))
        bk = 16 * last + np.floor(16 * g).astype(int)
        print('  %-46s %2d simulated steps: bucketed %.4f   (by the step alone %.4f)'
              % (nm, Tsim, wave_share(np.argsort(-bk, kind='stable')), wave_share(np.argsort(-last, kind='stable'))))
