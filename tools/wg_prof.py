import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from lq_mpc_amd import BatchSolver, synth, _lib
_lib.LIB_PATH = os.path.abspath(os.environ["LQMPC_LIB"])
dev = torch.device("cuda", 0)
s = BatchSolver(0, stream=torch.cuda.current_stream(dev).cuda_stream)
b = synth.make_batch(5, Bsz=4096)
nx, nu, N, Bsz, T = 8, 4, 30, 4096, 30
dA = torch.from_numpy(b["A"]).to(dev); dB = torch.from_numpy(b["B"]).to(dev); dx0 = torch.from_numpy(b["x0"]).to(dev)
dJ = torch.empty(Bsz, dtype=torch.float64, device=dev)
for rep in range(2):
    s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, b["Q"], b["R"], b["P"], b["lb"], b["ub"], dx0, b["A_true"], b["B_true"], dJ)
    torch.cuda.synchronize()
print(s.last_kernel(), dJ.sum().item())
