import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the 2-rank rehearsal of bench.py's sharded path (tests/test_dist_gpu.py): Bsz per rank, rollout length
TWO_RANK = dict(config=3, bsz=1024, steps=2, warmup=1)


def _gpu_run_selected(config):
    expr = config.getoption("markexpr", "") or ""
    return "gpu" in expr and "not gpu" not in expr and os.path.exists("/dev/kfd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._lqmpc_two_rank = None
    if _gpu_run_selected(config) and not hasattr(config, "workerinput"):
        # Two ranks on the one leased GPU (gloo for the collective, both on cuda:0).  The launcher has to be a child of a
        # process that has NOT initialised the GPU, so it is started here, before any test imports the HIP library; the test
        # waits for it.  stdout = bench.py's JSON line, the gathered J_T goes to a .npy.
        tmp = tempfile.mkdtemp(prefix="lqmpc_2rank_")
        out, err, dump = (os.path.join(tmp, f) for f in ("bench.json", "bench.err", "J_T.npy"))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-on-gpu0",
               "--config", str(TWO_RANK["config"]), "--bsz", str(TWO_RANK["bsz"]), "--steps", str(TWO_RANK["steps"]),
               "--warmup", str(TWO_RANK["warmup"]), "--no-extras", "--no-cpu-baseline", "--dump", dump]
        proc = subprocess.Popen(cmd, stdout=open(out, "w"), stderr=open(err, "w"), cwd=ROOT,
                                env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        config._lqmpc_two_rank = dict(proc=proc, out=out, err=err, dump=dump, cmd=cmd)


def pytest_unconfigure(config):
    job = getattr(config, "_lqmpc_two_rank", None)
    if job and job["proc"].poll() is None:
        job["proc"].kill()


@pytest.fixture(scope="session")
def two_rank_job(request):
    return request.config._lqmpc_two_rank


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def solver():
    from lq_mpc_amd import BatchSolver
    s = BatchSolver(0)
    yield s
    s.close()
