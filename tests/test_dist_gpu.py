"""GPU, two ranks on the one leased MI355X: bench.py's sharded path end to end.

`python bench.py --gpus 2 --backend gloo --all-on-gpu0` (started by tests/conftest.py before anything touched the GPU) builds ONE
global batch, cuts it with lq_mpc_amd.dist.shard_batch, rolls every shard out through the C ABI on cuda:0 and gathers J_T with
lq_mpc_amd.dist.all_gather_costs -- the code path `--gpus N` runs under RCCL on an 8-GPU node, with gloo standing in for the
collective because RCCL refuses two ranks on one device.  The gathered curve must equal the single-process HIP result bit for
bit (no result depends on which instances share a launch) and the CPU oracle to 1e-8."""
import json

import numpy as np
import pytest

from lq_mpc_amd import synth
from oracle import oracle as orc

from conftest import GOLDEN, TWO_RANK

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_match_single_process(two_rank_job, solver):
    assert two_rank_job is not None, "the 2-rank launcher was not started (conftest.pytest_configure: needs -m gpu and /dev/kfd)"
    rc = two_rank_job["proc"].wait(timeout=900)
    err = open(two_rank_job["err"]).read()
    assert rc == 0, f"{' '.join(two_rank_job['cmd'])} -> rc {rc}\n{err[-4000:]}"
    lines = [ln for ln in open(two_rank_job["out"]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    world, cfg = 2, TWO_RANK["config"]
    Bglobal = world * TWO_RANK["bsz"]
    assert out["n_gpus"] == world and out["config"]["batch_global"] == Bglobal and out["config"]["batch_per_gpu"] == TWO_RANK["bsz"]
    assert out["scaling"] == "weak" and out["value"] > 0 and out["config"]["status_nonzero"] == 0
    assert out["gather"]["mode"] == "final" and out["gather"]["own_shard_intact"] is True
    J2 = np.load(two_rank_job["dump"])
    assert J2.shape == (Bglobal,)
    b = synth.make_batch(cfg, Bsz=Bglobal, fixture_dir=GOLDEN)
    T = b["T"]
    a = (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"], b["x0"], b["A_true"], b["B_true"])
    g = solver.rollout_batch(T, *a)
    assert np.all(g["status"] == 0)
    assert np.array_equal(J2, g["J_T"]), f"sharded vs single-process HIP: max rel {np.max(np.abs(J2 - g['J_T']) / np.abs(g['J_T'])):.3e}"
    ref = orc.rollout_batch(T, *a)["J_T"]
    assert np.max(np.abs(J2 - ref) / np.abs(ref)) < 1e-8
    assert abs(out["gather"]["J_T_sum"] - float(J2.sum())) <= 1e-9 * abs(float(J2.sum()))
