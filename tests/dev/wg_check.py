"""Dev tool: workgroup kernel vs oracle on C5-shaped batches (and n=40 forced), plus timing."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, _lib
from oracle import oracle

def check(cfg, Bsz, kernel, T=6, K=4):
    b = synth.make_batch(cfg, Bsz=Bsz)
    s = BatchSolver(0)
    s.set_options(kernel=kernel)
    args = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
    g = s.solve_batch(*args, b['x0'])
    o = oracle.solve_batch(*args, b['x0'])
    print(cfg, s.last_kernel(), 'solve: du0 %.2e  dVN rel %.2e  status max %d  iters mean %.2f max %d' % (
        np.abs(g['u_0'] - o['u_0']).max(), np.abs(g['V_N'] / o['V_N'] - 1).max(), g['status'].max(), g['iters'].mean(), g['iters'].max()))
    g = s.rollout_batch(T, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
    o = oracle.rollout_batch(T, *args, b['x0'], b['A_true'], b['B_true'], want_traj=True)
    print('   rollout: dJT rel %.2e dX %.2e dU %.2e status max %d iters/step %.2f' % (
        np.abs(g['J_T'] / o['J_T'] - 1).max(), np.abs(g['X'] - o['X']).max(), np.abs(g['U'] - o['U']).max(), g['status'].max(), g['iters'].mean() / T))
    try:
        x0s = np.random.default_rng(1).standard_normal((b['nx'], K)) * 0.3
        g = s.max_vn_batch(*args, x0s)
        o = oracle.max_vn_batch(*args, x0s)
        print('   maxvn: rel %.2e' % np.abs(g['M_V'] / o - 1).max())
    except Exception as e:
        print('   maxvn failed', repr(e))

check(5, 64, _lib.KERNEL_WORKGROUP)
check(4, 64, _lib.KERNEL_WORKGROUP)
if len(sys.argv) > 1:
    import torch
    b = synth.make_batch(5, Bsz=int(sys.argv[1]))
    s = BatchSolver(0)
    args = (b['N'], b['A'], b['B'], b['Q'], b['R'], b['P'], b['lb'], b['ub'])
    for kern in (_lib.KERNEL_WORKGROUP,):
        s.set_options(kernel=kern)
        for rep in range(2):
            t0 = time.time(); g = s.rollout_batch(30, *args, b['x0'], b['A_true'], b['B_true']); dt = time.time() - t0
            print('rollout T=30 Bsz=%d %s: %.3f s -> %.3e QP-steps/s (host-inclusive), iters/step %.3f' % (b['Bsz'], s.last_kernel(), dt, b['Bsz'] * 30 / dt, g['iters'].mean() / 30))
        for rep in range(2):
            t0 = time.time(); g = s.solve_batch(*args, b['x0']); dt = time.time() - t0
            print('solve Bsz=%d: %.3f s -> %.3e QP/s, iters %.2f' % (b['Bsz'], dt, b['Bsz'] / dt, g['iters'].mean()))
