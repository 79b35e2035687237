"""GPU, bench.py's sharded path end to end on the one leased MI355X.

tests/conftest.py starts these jobs before anything in the test process touches the GPU:
* `bench.py --gpus 2 --backend gloo --all-on-gpu0`: ONE global batch cut with lq_mpc_amd.dist.shard_batch, every shard rolled out
  through the C ABI on cuda:0, J_T gathered with lq_mpc_amd.dist.all_gather_costs -- gloo standing in for the collective because
  RCCL refuses two ranks on one device;
* `bench.py --force-dist [...]`: a world of one rank on RCCL (backend nccl): init_process_group with device_id, the device-side
  all_gather_into_tensor, barrier and all_reduce, in both --gather modes and with --config 4 -- the exact calls of an N-GPU launch.
Every gathered curve must equal the single-process HIP result bit for bit (no result depends on which instances share a launch)
and the CPU oracle to 1e-8."""
import json

import numpy as np
import pytest

from lq_mpc_amd import synth
from oracle import oracle as orc

from conftest import GOLDEN, TWO_RANK

pytestmark = pytest.mark.gpu


def _job(dist_jobs, name):
    assert dist_jobs is not None, "the bench.py jobs were not started (conftest: needs -m gpu and /dev/kfd)"
    j = dist_jobs[name]
    rc = int(open(j["rc"]).read().strip())
    err = open(j["err"]).read()
    assert rc == 0, f"{' '.join(j['cmd'])} -> rc {rc}\n{err[-4000:]}"
    lines = [ln for ln in open(j["out"]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    return json.loads(lines[0]), np.load(j["npy"])


def _single_process(solver, cfg, Bglobal):
    b = synth.make_batch(cfg, Bsz=Bglobal, fixture_dir=GOLDEN)
    a = (b["N"], b["A"], b["B"], b["Q"], b["R"], b["P"], b["lb"], b["ub"], b["x0"], b["A_true"], b["B_true"])
    g = solver.rollout_batch(b["T"], *a)
    assert np.all(g["status"] == 0)
    return b, a, g


def test_two_ranks_on_one_gpu_match_single_process(dist_jobs, solver):
    out, J2 = _job(dist_jobs, "two_rank_gloo")
    world, cfg = 2, TWO_RANK["config"]
    Bglobal = world * TWO_RANK["bsz"]
    assert out["n_gpus"] == world and out["config"]["batch_global"] == Bglobal and out["config"]["batch_per_gpu"] == TWO_RANK["bsz"]
    assert out["scaling"] == "weak" and out["value"] > 0 and out["config"]["status_nonzero"] == 0
    assert out["gather"]["mode"] == "final" and out["gather"]["own_shard_intact"] is True
    assert J2.shape == (Bglobal,)
    b, a, g = _single_process(solver, cfg, Bglobal)
    assert np.array_equal(J2, g["J_T"]), f"sharded vs single-process HIP: max rel {np.max(np.abs(J2 - g['J_T']) / np.abs(g['J_T'])):.3e}"
    ref = orc.rollout_batch(b["T"], *a)["J_T"]
    assert np.max(np.abs(J2 - ref) / np.abs(ref)) < 1e-8
    assert abs(out["gather"]["J_T_sum"] - float(J2.sum())) <= 1e-9 * abs(float(J2.sum()))


@pytest.mark.parametrize("name,mode,cfg,bsz", [("nccl_world1_final", "final", 3, TWO_RANK["bsz"]),
                                               ("nccl_world1_per_step", "per-step", 3, TWO_RANK["bsz"]),
                                               ("nccl_world1_c4", "final", 4, 4096)])
def test_rccl_branch_world_of_one(dist_jobs, solver, name, mode, cfg, bsz):
    """The nccl (= RCCL) branch of bench.py executed on the lease: same J_T as the plain single-process call, bit for bit."""
    out, J = _job(dist_jobs, name)
    assert out["n_gpus"] == 1 and out["gather"]["world_size"] == 1 and out["gather"]["mode"] == mode
    assert "RCCL" in out["gather"]["backend"] and out["gather"]["own_shard_intact"] is True
    assert out["config"]["status_nonzero"] == 0 and out["value"] > 0
    assert J.shape == (bsz,)
    b, a, g = _single_process(solver, cfg, bsz)
    assert np.array_equal(J, g["J_T"])
    ref = orc.rollout_batch(b["T"], *a)["J_T"]
    assert np.max(np.abs(J - ref) / np.abs(ref)) < 1e-8
