"""GPU (-m gpu): the data-parallel path with the REAL kernels on this 1-GPU box: two ranks share cuda:0
(NERF_ALLOW_SHARED_GPU=1) and exchange gradients over gloo (RCCL refuses two ranks on one device; on a multi-GPU node the
same code runs one rank per GPU over RCCL).  Checks what tests/test_parallel_cpu.py checks with oracle gradients, but
through render() -> HIP backward -> flat gradient buckets -> all-reduce -> fused Adam, and bench.py's N > 1 line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, precision="fp16x3"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), NERF_ALLOW_SHARED_GPU="1")
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nerf_pytorch_amd as npa
    import workloads as wl
    from nerf_pytorch_amd import parallel
    r, w, dev = parallel.init_distributed(backend="gloo")
    assert dev.type == "cuda"
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    Pc, Pf = wl.scene_params(seed=rank)                          # ranks start from DIFFERENT weights ...
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    with torch.no_grad():                                        # ... and have rendered once (fragment repack cached)
        npa.render_rays(wl.synthetic_rays(8, seed=1).to(dev), nc, None, 64, N_importance=128, network_fine=nf)
    parallel.broadcast_parameters([nc, nf])                      # must invalidate that cache
    n = 256
    cfg = wl.LEGO
    batch = wl.lego_batch(n, seed=9).to(dev)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    rnd_all = {k: v.to(dev) for k, v in wl.synthetic_randoms(n, 64, 128, seed=2).items() if k in ("t_rand", "u")}
    args = dict(chunk=1 << 15, ndc=False, near=cfg["near"], far=cfg["far"], use_viewdirs=True, network_fn=nc, network_query_fn=None,
                N_samples=64, N_importance=128, network_fine=nf, perturb=1.0, white_bkgd=True, raw_noise_std=0.)
    npa.set_precision(precision)

    def grads(rays, tgt, rnd):
        for m in (nc, nf):
            m.zero_grad()
        rgb, _, _, ex = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=rays, randoms=rnd, **args)
        (npa.img2mse(rgb, tgt) + npa.img2mse(ex["rgb0"], tgt)).backward()
    full = None
    if rank == 0:                                                # reference: the full batch on one rank
        grads(batch, target, rnd_all)
        full = torch.cat([nc.last_flat_grad, nf.last_flat_grad]).cpu().numpy().copy()
    sh_rays, sh_tgt = parallel.shard_rays(batch, target)
    lo, hi = parallel.shard_slice(n, rank, world)
    sync = parallel.GradientSync([nc, nf])                       # the coarse bucket's exchange starts under the fine backward
    grads(sh_rays.contiguous(), sh_tgt, {k: v[lo:hi].contiguous() for k, v in rnd_all.items()})
    assert sync.started == 2, sync.started
    sync.finish()
    sync.close()
    assert parallel._flat_grad_of(nc) is nc.last_flat_grad and parallel._flat_grad_of(nf) is nf.last_flat_grad
    avg = torch.cat([nc.last_flat_grad, nf.last_flat_grad]).cpu().numpy().copy()
    opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)
    opt.step()
    with torch.no_grad():                                        # the step must reach the kernels on every rank
        img = npa.render_rays(wl.synthetic_rays(16, seed=3).to(dev), nc, None, 64, N_importance=128, network_fine=nf)["rgb_map"]
    q.put((rank, full, avg, torch.cat([nc.flat_params(), nf.flat_params()]).detach().cpu().numpy().copy(), img.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
def test_two_ranks_on_one_gpu_match_the_full_batch(precision):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, precision)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=480) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, full, avg0, w0, img0), (_, _, avg1, w1, img1) = res
    assert np.array_equal(avg0, avg1), "ranks hold different gradients after the all-reduce"
    assert np.array_equal(w0, w1) and np.array_equal(img0, img1), "ranks diverged after broadcast + Adam"
    # mean of the shard gradients == gradient of the full batch (the loss is a mean over rays): fp32 summation order only
    assert float(np.abs(avg0 - full).max()) <= 2e-5 * float(np.abs(full).max()), float(np.abs(avg0 - full).max() / np.abs(full).max())


@pytest.mark.timeout(600)
def test_bench_line_with_two_ranks():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["NERF_ALLOW_SHARED_GPU"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                          "--warmup", "1", "--rays", "1024", "--single-datapath", "--no-gate", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=540, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["config"]["global_batch_rays"] == 2048
    assert line["value"] > 0 and "all-reduce" in line["collective"] and line["scaling"] == "weak"
    # round 6: every rank's own clock (a slow GCD or a rank stuck in the exchange is visible), the fabric and the knobs of the run
    by = line["multi_gpu"]["ms_per_step_by_rank"]
    assert len(by["by_rank"]) == 2 and by["min"] <= by["max"] <= 1.001 * line["ms_per_step"] + 1e-6
    assert "fabric_topology" in line["multi_gpu"] and line["multi_gpu"]["dist_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _frames_worker(rank, world, port, q, savedir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), NERF_ALLOW_SHARED_GPU="1")
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nerf_pytorch_amd as npa
    import workloads as wl
    from nerf_pytorch_amd import parallel
    r, w, dev = parallel.init_distributed(backend="gloo")
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    Pc, Pf = wl.scene_params()
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    H, W, focal = 20, 24, 30.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([torch.as_tensor(wl.pose_spherical(float(th), -30.0, 4.0), dtype=torch.float32) for th in np.linspace(-180, 180, 5 + 1)[:-1]]).to(dev)
    rk = dict(network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128, network_fine=nf, perturb=0., white_bkgd=True,
              raw_noise_std=0., ndc=False, near=2., far=6., use_viewdirs=True)
    with torch.no_grad():
        rgbs, disps = parallel.render_path(poses, (H, W, focal), K, 1 << 15, rk, savedir=os.path.join(savedir, "parallel"))
        single = None
        if rank == 0:       # the single-process function on the same poses (run_nerf.py:137-175)
            single = npa.render_path(poses, (H, W, focal), K, 1 << 15, rk, savedir=os.path.join(savedir, "single"))
    q.put((rank, rgbs, disps, single, parallel.frames_of_rank(5)))
    dist.barrier()
    parallel.shutdown()


@pytest.mark.timeout(600)
def test_frame_parallel_render_path_equals_the_single_process_one(tmp_path):
    """BASELINE configs[4] as a PATH (round 6): parallel.render_path deals the poses round-robin over two ranks sharing this GPU, each
    renders and writes ITS frames, rank 0 gathers -- arrays and PNG files bit-identical to render_path's on one rank; the other rank
    returns (None, None); no collective on the rendering path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    for d in ("parallel", "single"):
        os.makedirs(tmp_path / d)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=480) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, rgbs, disps, single, fr0), (_, rgbs1, disps1, _, fr1) = res
    assert rgbs1 is None and disps1 is None and fr0 == [0, 2, 4] and fr1 == [1, 3]
    assert rgbs.shape == (5, 20, 24, 3) and disps.shape == (5, 20, 24) and rgbs.dtype == np.float32
    assert np.array_equal(rgbs, single[0]) and np.array_equal(disps, single[1], equal_nan=True)
    for i in range(5):
        a, b = open(tmp_path / "parallel" / f"{i:03d}.png", "rb").read(), open(tmp_path / "single" / f"{i:03d}.png", "rb").read()
        assert a == b and a[:8] == b"\x89PNG\r\n\x1a\n", i
