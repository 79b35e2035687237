import os, sys, numpy as np
import torch
torch.zeros(1, device='cuda:0')
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
from oracle import oracle
b = synth.make_batch(3, Bsz=4096)
s = BatchSolver(0)
args = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
o = oracle.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
for mi in (50, 51, 52, 53):
    s.set_options(max_iter=mi, order=0)
    g = s.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
    err = np.abs(g['J_T'] / o['J_T'] - 1)
    print(mi, s.last_kernel(), 'status', np.bincount(g['status'], minlength=4), 'n(err>1e-9)', (err > 1e-9).sum(), 'max', err.max(), 'iters/step', g['iters'].mean() / 30)
    if mi == 50:
        w = np.argmax(err)
        du = np.abs(g['U'][:, :, w] - o['U'][:, :, w]).max(axis=0)
        print('  worst inst', w, 'first bad step', np.argmax(du > 1e-9), 'dU per step', np.round(du[:12], 6))
        print('  U gpu', g['U'][:, :6, w].T.ravel()); print('  U orc', o['U'][:, :6, w].T.ravel())
