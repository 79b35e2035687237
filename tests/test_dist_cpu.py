"""CPU, world_size 2 over gloo: the sharding / gather logic of the multi-GPU path.

The per-shard compute is the oracle here (tests may use it); on the GPU box bench.py runs the same
sharding with the HIP path and backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lq_mpc_amd import dist as ld
from lq_mpc_amd import synth
from oracle import oracle as orc


def test_shard_bounds_cover_the_batch():
    for Bsz in (1, 7, 64, 65, 262144):
        for world in (1, 2, 3, 8):
            b = [ld.shard_bounds(Bsz, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == Bsz
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, Bsz, T, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = synth.make_batch(3, Bsz=Bsz)
        sh = ld.shard_batch(b, rank, world)
        lo, hi = sh["shard"]
        assert sh["A"].shape[-1] == hi - lo
        J = orc.rollout_batch(T, sh["N"], sh["A"], sh["B"], sh["Q"], sh["R"], sh["P"], sh["lb"], sh["ub"], sh["x0"],
                              sh["A_true"], sh["B_true"], threads=2)["J_T"]
        full = ld.all_gather_costs(torch.from_numpy(J), Bsz)
        cols = 4
        rows = (hi - lo) // cols * cols
        lo_s, hi_s, mean, std = ld.column_stats(torch.from_numpy(J[:rows].reshape(-1, cols)))
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"stats_{rank}.npy"), torch.stack([lo_s, hi_s, mean, std]).numpy())
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), J[:rows].reshape(-1, cols))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Bsz", [64, 37])
def test_two_rank_sharded_rollout_matches_single_process(tmp_path, Bsz):
    T, world = 5, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, Bsz, T, str(tmp_path)), nprocs=world, join=True)
    b = synth.make_batch(3, Bsz=Bsz)
    ref = orc.rollout_batch(T, b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"], b["x0"],
                            b["A_true"], b["B_true"], threads=2)["J_T"]
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"full_{r}.npy"), ref)      # every rank holds the gathered curve
    table = np.concatenate([np.load(tmp_path / f"rows_{r}.npy") for r in range(world)], axis=0)
    want = np.stack([table.min(0), table.max(0), table.mean(0), table.std(0)])
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / f"stats_{r}.npy"), want, rtol=1e-9, atol=1e-12)
