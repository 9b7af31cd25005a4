"""GPU (-m gpu): the RCCL branch of nerf_pytorch_amd.parallel EXECUTED on this 1-GPU box.

A node with several GPUs is not available to these tests, and RCCL refuses two ranks on one device, so rounds 1-4 covered the
data-parallel path with gloo only.  NERF_FORCE_PROCESS_GROUP=1 (parallel.FORCE_GROUP) builds a ONE-rank process group on backend
"nccl" (= RCCL on ROCm) and disables every world-size-1 short cut: the probe all-reduce of init_distributed, broadcast_parameters,
GradientSync's asynchronous all-reduces with the real ProcessGroupNCCL work objects (started under the backward on kernels that
were enqueued through ctypes on torch's current stream), finish(), ranks_seen / ranks_identical (all-gather) and bench.py's
multi_gpu block all run through RCCL.  With one rank the sum over ranks is the identity, so every number must be BIT-identical to
the step without a process group -- which is exactly what makes ordering mistakes visible (an all-reduce that ran before the
kernels that produce the bucket would hand back a stale bucket).

Each test runs in its own process (the process group is process-global state)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, os, sys
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
import nerf_pytorch_amd as npa
import workloads as wl
from nerf_pytorch_amd import parallel

dev = torch.device("cuda", 0)
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
cfg = wl.LEGO
n = 1024
batch = wl.lego_batch(n, seed=9).to(dev)
target = torch.rand(n, 3, generator=torch.Generator().manual_seed(4)).to(dev)
rnd = {k: v.to(dev) for k, v in wl.synthetic_randoms(n, 64, 128, seed=2).items() if k in ("t_rand", "u")}


def run(steps, sync_factory):
    Pc, Pf = wl.scene_params()
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    parallel.broadcast_parameters([nc, nf])
    opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)
    sync = sync_factory([nc, nf])
    args = dict(chunk=1 << 15, ndc=False, near=cfg["near"], far=cfg["far"], use_viewdirs=True, network_fn=nc, network_query_fn=None,
                N_samples=64, N_importance=128, network_fine=nf, perturb=1.0, white_bkgd=True, raw_noise_std=0.)
    grads = []
    for _ in range(steps):
        rgb, _, _, ex = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, randoms=rnd, retraw=True, **args)
        opt.zero_grad()
        (npa.img2mse(rgb, target) + npa.img2mse(ex["rgb0"], target)).backward()
        if sync is not None:
            sync.finish()
        grads.append(torch.cat([nc.last_flat_grad, nf.last_flat_grad]).clone())
        opt.step()
    started = sync.started if sync is not None else 0
    if sync is not None:
        sync.close()
    return grads, torch.cat([nc.flat_params(), nf.flat_params()]).detach().clone(), started, (nc, nf)


# 1. no process group: the plain single-process step
assert not dist.is_initialized()
g0, w0, _, _ = run(3, lambda models: None)

# 2. the same steps with a forced one-rank RCCL group
rank, world, dev2 = parallel.init_distributed(backend="nccl", force_group=True)
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
g1, w1, started, nets = run(3, lambda models: parallel.GradientSync(models))
out = {
    "backend": dist.get_backend(), "world": dist.get_world_size(), "started": started,
    "grads_bit_identical": all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(g0, g1)),
    "params_bit_identical": bool(torch.equal(w0.view(torch.int32), w1.view(torch.int32))),
    "grad_nonzero": bool(float(g1[0].abs().max()) > 0),
    "ranks_seen": parallel.ranks_seen(), "ranks_identical": parallel.ranks_identical([m.flat_params() for m in nets]),
    "rccl": list(torch.cuda.nccl.version()) if isinstance(torch.cuda.nccl.version(), (tuple, list)) else torch.cuda.nccl.version(),
}
# 3. the plain (non-overlapped) exchange and a second bucket inside one backward (render(chunk < N_rand): one node per chunk)
nc, nf = nets
for m in nets:
    m.zero_grad()
sync = parallel.GradientSync([nc, nf])
args = dict(chunk=512, ndc=False, near=cfg["near"], far=cfg["far"], use_viewdirs=True, network_fn=nc, network_query_fn=None,
            N_samples=64, N_importance=128, network_fine=nf, perturb=1.0, white_bkgd=True, raw_noise_std=0.)
rgb, _, _, ex = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, randoms=rnd, retraw=True, **args)
(npa.img2mse(rgb, target) + npa.img2mse(ex["rgb0"], target)).backward()
sync.finish()
sync.close()
chunked = torch.cat([p.grad.reshape(-1) for m in nets for p in m.param_list()])
for m in nets:
    m.zero_grad()
rgb, _, _, ex = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, randoms=rnd, retraw=True, **args)
(npa.img2mse(rgb, target) + npa.img2mse(ex["rgb0"], target)).backward()
parallel.allreduce_gradients(list(nets))
plain = torch.cat([p.grad.reshape(-1) for m in nets for p in m.param_list()])
out["chunked_equals_plain"] = bool(torch.equal(chunked.view(torch.int32), plain.view(torch.int32)))
out["chunked_nonzero"] = bool(float(chunked.abs().max()) > 0)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["NERF_PRECISION"] = "fp16x3"
    return env


@pytest.mark.timeout(600)
def test_gradient_sync_over_a_one_rank_rccl_group_is_bit_identical():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT], capture_output=True, text=True, timeout=540, env=_env(), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, res.stdout[-2000:]
    out = json.loads(line[-1][len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["started"] == 6, out                 # two buckets per step, both exchanged asynchronously under / after the backward
    assert out["grad_nonzero"] and out["grads_bit_identical"] and out["params_bit_identical"], out
    assert out["ranks_seen"] == [0] and out["ranks_identical"] is True
    assert out["chunked_nonzero"] and out["chunked_equals_plain"], out


@pytest.mark.timeout(600)
def test_bench_line_over_a_one_rank_rccl_group():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-group", "--steps", "3",
                          "--warmup", "2", "--single-datapath", "--no-gate", "--no-cpu-baseline", "--no-eager-baseline", "--no-configs",
                          "--sustained-s", "0"], capture_output=True, text=True, timeout=540, env=_env(), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    m = line["multi_gpu"]
    assert line["n_gpus"] == 1 and line["world_size"] == 1 and line["value"] > 0
    assert m["backend"] == "nccl" and m["rccl_ranks_seen"] == [0] and m["ranks_identical"] is True
    assert m["allreduce_ms"] > 0 and m["rccl_version"] and "RCCL" in line["collective"]
    assert m["overlap_started"] == 2 * (2 + 3 + 3), m      # two buckets per step over warmup + steps + the per-kernel pass
