"""CPU: the N > 1 launch path of bench.py.  `python bench.py --gpus N` with no torchrun environment must become N
ranks (torch.distributed.run, 127.0.0.1 rendezvous) and must refuse to report an N-GPU number from fewer devices."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(300)
def test_bench_self_launches_two_gloo_ranks():
    res = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--rays", "64")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout          # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["scaling"] == "weak"
    assert line["rays_per_rank"] == 64 and line["global_batch_rays"] == 128


@pytest.mark.timeout(300)
def test_bench_strong_mode_splits_the_global_batch():
    res = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--strong")
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "strong" and line["global_batch_rays"] == 32768 and line["rays_per_rank"] == 16384


@pytest.mark.timeout(600)
def test_bench_eight_gloo_ranks_dry_run():
    """The driver's 8-GPU command shape under gloo: rendezvous of 8 ranks, the strong split 8 x 4096 of the 32,768-ray
    batch, the 40 spiral frames dealt 5 per rank, every rank answering the census, identical-parameter check."""
    res = _run("--gpus", "8", "--dry-run", "--backend", "gloo", "--strong", timeout=540)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 8 and line["world_size"] == 8
    assert line["rays_per_rank"] == 4096 and line["rays_of_all_ranks"] == 32768 and line["frames_of_all_ranks"] == 40
    assert line["rccl_ranks_seen"] == list(range(8)) and line["ranks_identical"] is True
    res = _run("--gpus", "8", "--dry-run", "--backend", "gloo", timeout=540)        # weak: N_rand per rank
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["rays_per_rank"] == 4096 and line["global_batch_rays"] == 32768 and line["scaling"] == "weak"


def test_init_failure_names_the_environment(monkeypatch):
    """A process group that cannot form must fail with the backend, the device and the NCCL/HSA environment in the
    message (the 8-GPU run happens on a box nobody watches), not hang."""
    sys.path.insert(0, ROOT)
    import nerf_pytorch_amd  # noqa: F401
    from nerf_pytorch_amd import parallel
    import torch.distributed as dist
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("NCCL_DEBUG", "WARN")

    def boom(**kw):
        raise RuntimeError("simulated fabric failure")
    monkeypatch.setattr(dist, "init_process_group", boom)
    with pytest.raises(RuntimeError) as ei:
        parallel.init_distributed(backend="gloo")
    msg = str(ei.value)
    assert "process group init failed on rank 0/2" in msg and "simulated fabric failure" in msg and "NCCL_DEBUG" in msg


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    res = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert res.returncode == 2
    assert "refusing to report" in res.stderr


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert res.returncode != 0 and "--gpus 2" in (res.stderr + res.stdout)
