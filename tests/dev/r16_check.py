"""Dev tool: 16-lane-row rollout kernel vs the packed kernel and the oracle on C3-shaped batches."""
import os, sys, time, numpy as np
import torch
torch.zeros(1, device='cuda:0')
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth
from oracle import oracle
b = synth.make_batch(3, Bsz=4096)
s = BatchSolver(0)
args = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
g = s.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
print(s.last_kernel(), 'status counts', np.bincount(g['status'], minlength=4), 'iters mean/step', g['iters'].mean() / 30, 'max', g['iters'].max())
o = oracle.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
print('vs oracle: dJT rel %.2e dX %.2e dU %.2e' % (np.abs(g['J_T'] / o['J_T'] - 1).max(), np.abs(g['X'] - o['X']).max(), np.abs(g['U'] - o['U']).max()))
bad = np.argsort(-np.abs(g['J_T'] / o['J_T'] - 1))[:5]
print('worst', bad, np.abs(g['J_T'] / o['J_T'] - 1)[bad], g['status'][bad], g['iters'][bad])
if len(sys.argv) > 1:
    dev = torch.device('cuda:0')
    b = synth.make_batch(3); K = b['Bsz']
    dA, dB, dx0 = (torch.from_numpy(a).to(dev) for a in (b['A'], b['B'], b['x0']))
    dJ = torch.empty(K, dtype=torch.float64, device=dev); dit = torch.empty(K, dtype=torch.int32, device=dev); dst = torch.empty(K, dtype=torch.int32, device=dev)
    for env in ('1', '0'):
        s.set_options(layout=int(env))
        for order in (1, 0):
            s.set_options(order=order)
            ts = []
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                s.rollout_batch_dev(4, 2, 10, K, 30, dA.data_ptr(), dB.data_ptr(), b['Q'], b['R'], b['P'], b['lb'], b['ub'], dx0.data_ptr(), b['A_true'], b['B_true'], dJ.data_ptr(), dstatus=dst.data_ptr(), diters=dit.data_ptr())
                s.sync(); ts.append(time.perf_counter() - t0)
            print('R16=%s order=%d %s best %.3f ms -> %.3e QP-steps/s; J sum %.9e status!=0 %d' % (env, order, s.last_kernel(), min(ts) * 1e3, K * 30 / min(ts), dJ.sum().item(), int((dst != 0).sum())))
