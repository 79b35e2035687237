"""Dev tool: distribution of factorisations per instance for the C3 rollout (what bounds the longest wavefront)."""
import sys, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
b = synth.make_batch(3)
s = BatchSolver(0)
args = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
r = s.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
it = r['iters'].astype(np.int64)
print('instances', it.size, 'mean fact/step', it.mean() / 30, 'max per instance', it.max(), 'status!=0', (r['status'] != 0).sum())
srt = np.sort(it)[::-1]
print('top 16:', srt[:16], ' sum of maxima per wave of 16 (sorted):', srt[::16][:8])
print('quantiles 50/90/99/99.9/100:', np.percentile(it, [50, 90, 99, 99.9, 100]))
sat = (np.abs(r['U']) >= 0.1 - 1e-12).any(axis=0)          # (T, Bsz): any input saturated at step
print('steps with a saturated applied input: mean per instance', sat.sum(axis=0).mean(), 'max', sat.sum(axis=0).max())
hard = np.argsort(-it)[:16]
print('hardest instances: iters', it[hard], 'saturated steps', sat.sum(axis=0)[hard])
# work if every wave paid max over its 16 instances (natural order vs sorted order)
print('sum over waves of max-in-wave: natural %d, sorted %d, plain sum/16 %d' % (it.reshape(-1, 16).max(axis=1).sum(), srt.reshape(-1, 16).max(axis=1).sum(), it.sum() // 16))
