import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(5, Bsz=8192)
s = BatchSolver(0)
a = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
r = s.rollout_batch(30, *a, b['x0'], b['A_true'], b['B_true'], want_traj=True)
it = r['iters']
print(s.last_kernel(), 'iters mean/step', it.mean() / 30, 'max', it.max(), 'quantiles 50/90/99/99.9', np.percentile(it, [50, 90, 99, 99.9]))
sat = (np.abs(r['U']) >= 0.1 - 1e-12).sum(axis=0)      # saturated inputs per step (of 4) -- stage 0 only
print('instances with iters > 60:', (it > 60).sum(), ' max saturated stage-0 inputs over steps', sat.max())
