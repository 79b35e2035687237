"""Dev tool: time lqmpc_bounds_batch_dev on the synthetic configs, on-chip kernels against the HBM-workspace kernel."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from lq_mpc_amd import BatchSolver, synth, KERNEL_GENERIC, KERNEL_AUTO
dev = torch.device('cuda', 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
for cfg, bsz in ((3, None), (4, None), (5, None), (2, None)):
    b = synth.make_batch(cfg, Bsz=bsz)
    nx, nu, N, Bsz = b['A'].shape[0], b['B'].shape[1], b['N'], b['Bsz']
    dA = torch.from_numpy(b['A']).to(dev); dB = torch.from_numpy(b['B']).to(dev)
    dMV = torch.full((Bsz,), 0.5, dtype=torch.float64, device=dev); dlev = torch.full((Bsz,), 5e-3, dtype=torch.float64, device=dev)
    outs = [torch.empty(Bsz, dtype=torch.float64, device=dev) for _ in range(5)]
    dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
    xb, pb = np.ascontiguousarray(b['x0'][:, 0]), np.array([0.1, 1.0, 0.6])
    res = {}
    for name, kern in (('chip', KERNEL_AUTO), ('workspace', KERNEL_GENERIC)):
        if name == 'workspace' and cfg == 5: continue
        s.set_options(kernel=kern)
        def go():
            s.bounds_batch_dev(nx, nu, N, Bsz, dA, dB, b['Q'], b['R'], b['lb'], b['ub'], dlev, dlev, dMV, xb, pb, 1.0,
                               dalpha=outs[0], dbeta=outs[1], dxi=outs[2], deta=outs[3], dbound=outs[4], dstatus=dst)
        go(); go(); torch.cuda.synchronize()
        s.timer_begin()
        for _ in range(3): go()
        ms = s.timer_end() / 3
        res[name] = (ms, [o.double().nan_to_num().sum().item() for o in outs], int((dst != 0).sum().item()))
        print(f"C{cfg} {name:9s} {ms:9.3f} ms  {Bsz / ms * 1e3:.3e} systems/s  status!=0: {res[name][2]}  kernel {s.last_kernel()}", flush=True)
    if len(res) == 2:
        print("   max rel dev of the sums:", max(abs(a - c) / max(abs(c), 1e-300) for a, c in zip(res['chip'][1], res['workspace'][1])))
s.set_options(kernel=KERNEL_AUTO)
