"""CPU: bench.py's launch logic (the advisor's round-1 finding: `bench.py --gpus N` silently measured one GPU).

`python bench.py --gpus N` without a launcher must either start N ranks itself or refuse loudly; a WORLD_SIZE that disagrees with
--gpus must be an error, not a warning.  (The sharded path itself runs in tests/test_dist_cpu.py on gloo and, on the GPU box, in
tests/test_dist_gpu.py with two ranks on the leased GPU.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_visible_is_refused_before_any_gpu_call():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
