import os
import signal
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# bench.py's sharded path on the one leased GPU (tests/test_dist_gpu.py): per-rank batch size and steps of the rehearsals
TWO_RANK = dict(config=3, bsz=1024, steps=2, warmup=1)
_COMMON = ["--steps", str(TWO_RANK["steps"]), "--warmup", str(TWO_RANK["warmup"]), "--no-extras", "--no-cpu-baseline"]
DIST_JOBS = {
    # two ranks sharing cuda:0, gloo for the collective (RCCL refuses two ranks on one device)
    "two_rank_gloo": ["--gpus", "2", "--backend", "gloo", "--all-on-gpu0", "--config", "3", "--bsz", str(TWO_RANK["bsz"])],
    # a world of ONE rank on RCCL: init_process_group("nccl", device_id), device-side all_gather_into_tensor, barrier, all_reduce --
    # every call an N-GPU launch makes, executed on the lease
    "nccl_world1_final": ["--force-dist", "--config", "3", "--bsz", str(TWO_RANK["bsz"])],
    "nccl_world1_per_step": ["--force-dist", "--gather", "per-step", "--config", "3", "--bsz", str(TWO_RANK["bsz"])],
    "nccl_world1_c4": ["--force-dist", "--config", "4", "--bsz", "4096"],
}


def _gpu_run_selected(config):
    expr = config.getoption("markexpr", "") or ""
    return "gpu" in expr and "not gpu" not in expr and os.path.exists("/dev/kfd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._lqmpc_dist = None


def pytest_collection_modifyitems(config, items):
    """Start the multi-process bench.py jobs only when tests/test_dist_gpu.py is part of a GPU run.  Their launcher has to be the
    child of a process that has NOT initialised the GPU, so they start here -- after collection, before any fixture builds the
    solver -- one after the other in their own session (process group), and the `dist_jobs` fixture waits for them before anything
    in this process touches the GPU."""
    if config._lqmpc_dist is not None or hasattr(config, "workerinput") or not _gpu_run_selected(config):
        return
    if not any("test_dist_gpu" in it.nodeid and it.get_closest_marker("gpu") for it in items):
        return
    tmp = tempfile.mkdtemp(prefix="lqmpc_dist_")
    jobs, script = {}, []
    for name, extra in DIST_JOBS.items():
        f = {k: os.path.join(tmp, f"{name}.{k}") for k in ("out", "err", "npy", "rc")}
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra + _COMMON + ["--dump", f["npy"]]
        script.append(" ".join(cmd) + f" > {f['out']} 2> {f['err']}; echo $? > {f['rc']}")
        jobs[name] = dict(cmd=cmd, **f)
    proc = subprocess.Popen(["bash", "-c", "\n".join(script)], cwd=ROOT, start_new_session=True,
                            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    config._lqmpc_dist = dict(proc=proc, jobs=jobs)


def pytest_unconfigure(config):
    job = getattr(config, "_lqmpc_dist", None)
    if job and job["proc"].poll() is None:
        try:
            os.killpg(job["proc"].pid, signal.SIGKILL)          # the whole session: launcher, torch.distributed.run, the ranks
        except ProcessLookupError:
            pass


@pytest.fixture(scope="session")
def dist_jobs(request):
    """Waits for the bench.py jobs (if this run started any) and returns {name: {cmd, out, err, npy, rc}}."""
    job = request.config._lqmpc_dist
    if job is None:
        return None
    job["proc"].wait(timeout=1500)
    return job["jobs"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def solver(dist_jobs):                     # (after the multi-process jobs: nothing else on the GPU while they run)
    from lq_mpc_amd import BatchSolver
    s = BatchSolver(0)
    yield s
    s.close()
