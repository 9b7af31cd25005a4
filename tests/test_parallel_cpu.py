"""CPU, world_size 2, gloo: the data-parallel logic around the hot path (nerf-pytorch_amd/parallel.py).

The HIP kernels need a GPU, so the per-rank gradients here come from the oracle; what is tested is the
N > 1 control flow itself: ray sharding, ONE all-reduce over the flat gradient bucket the .grad tensors
are views of, identical Adam steps on every rank, and frame dealing for render_only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nerf_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_flat_grad(P, rays, target, n_samples=16):
    """coarse-only oracle loss gradient, flattened in state_dict order"""
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = orc.trace_rays(rays, Pg, None, n_samples, 0, perturb=0., white_bkgd=True)
    orc.mse(out["rgb_map"], target).backward()
    return torch.cat([Pg[k].grad.reshape(-1) for k, _ in orc.param_shapes()])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel
    torch.set_num_threads(2)
    r, w, dev = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights ...
    net = npa.NeRF(**kw)
    net._packed, net._packed_key = {"fp32": "stale repack"}, ("stale",)
    parallel.broadcast_parameters([net])               # ... and are made identical by one broadcast
    # c10d writes do not advance tensor versions: the broadcast must drop the cached fragment repack itself
    assert net._packed is None and net._packed_key is None
    P0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    n = 32
    rays_all = orc.synthetic_rays(n, seed=4)
    target_all = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))
    batch = torch.stack([rays_all[:, 0:3], rays_all[:, 3:6]], 0)
    sh_rays, sh_tgt = parallel.shard_rays(batch, target_all)
    lo, hi = parallel.shard_slice(n, rank, world)
    assert torch.equal(sh_rays[0], rays_all[lo:hi, 0:3]) and torch.equal(sh_tgt, target_all[lo:hi])
    # per-rank gradient of the local shard, installed the way the HIP backward installs it:
    # every .grad is a view into one flat bucket
    flat = _oracle_flat_grad(P0, rays_all[lo:hi], target_all[lo:hi])
    net.last_flat_grad = flat
    for nm, off, shape in npa.hip_backend.param_table():
        dict(net.named_parameters())[nm].grad = flat[off:off + int(np.prod(shape))].view(shape)
    assert parallel._flat_grad_of(net) is flat
    parallel.allreduce_gradients([net])
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    opt.step()
    # GradientSync: bucket A is reported final while its .grads do not exist yet -> its exchange starts right there
    # (under the rest of the backward); bucket B's .grads exist (accumulation) -> exchanged in finish(); C copied by autograd
    import sys
    render = sys.modules[parallel.__name__.rsplit(".", 1)[0] + ".render"]     # (the package attribute `render` is the function)
    nets = [npa.NeRF(**kw) for _ in range(3)]
    sync = parallel.GradientSync(nets)
    fl = [torch.full((npa.hip_backend.N_PARAMS,), float(10 * (i + 1) + rank)) for i in range(3)]
    table = npa.hip_backend.param_table()

    def install(net, flat, views=True):
        for nm, off, shape in table:
            v = flat[off:off + int(np.prod(shape))].view(shape)
            dict(net.named_parameters())[nm].grad = v if views else v.clone()
    install(nets[1], fl[1])
    for i in (0, 1, 2):
        render._grad_ready(nets[i], fl[i])
    assert sync.started == 2 and set(sync.pending) == {id(nets[0]), id(nets[2])}
    install(nets[0], fl[0])
    install(nets[2], fl[2], views=False)
    sync.finish()
    sync.close()
    assert render.GRAD_READY_HOOKS == [] and not sync.pending
    for i in range(3):          # mean over the two ranks of 10(i+1) + rank
        want = 10.0 * (i + 1) + 0.5
        assert all(bool((p.grad == want).all()) for p in nets[i].parameters()), i
    # a network that reports TWO buckets in one backward (render(chunk < N_rand): one autograd node per chunk): autograd
    # sums them out of place, i.e. it reads bucket #1 whose exchange was started early -- the averaged .grad must still be
    # the mean over ranks of (b1 + b2), and a GradientSync created for the same networks retires the forgotten one
    forgotten = parallel.GradientSync(nets)
    sync2 = parallel.GradientSync(nets)
    assert len(render.GRAD_READY_HOOKS) == 1, "a second GradientSync for the same networks must retire the first"
    for nn in nets:
        for pp in nn.parameters():
            pp.grad = None
    b1 = torch.full((npa.hip_backend.N_PARAMS,), 1.0 + rank)
    b2 = torch.full((npa.hip_backend.N_PARAMS,), 100.0 + 10 * rank)
    render._grad_ready(nets[0], b1)
    assert sync2.started == 1
    render._grad_ready(nets[0], b2)                 # second report: exchange of b1 finished and turned back into a share
    assert id(nets[0]) in sync2.multi and not sync2.pending
    for nm, off, shape in table:                    # what autograd does with two contributions: an out-of-place sum
        n_el = int(np.prod(shape))
        dict(nets[0].named_parameters())[nm].grad = b1[off:off + n_el].view(shape) + b2[off:off + n_el].view(shape)
    with sync2:
        sync2.finish()
    assert render.GRAD_READY_HOOKS == [] and not sync2.multi
    want = (1.0 + 2.0) / 2 + (100.0 + 110.0) / 2
    assert all(bool(((p.grad - want).abs() < 1e-4).all()) for p in nets[0].parameters())
    frames = parallel.frames_of_rank(7)
    got = parallel.gather_frames([np.full((2, 2), i) for i in frames], frames, 7)
    q.put((rank, flat.numpy().copy(), net.flat_params().detach().numpy().copy(), {k: v.numpy().copy() for k, v in P0.items()},
           [int(f[0, 0]) for f in got]))        # numpy: pickled by value (torch tensors travel as fds of a dying process)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, w0, P0, fr0), (_, g1, w1, P1, fr1) = res
    for k in P0:
        assert np.array_equal(P0[k], P1[k]), "broadcast_parameters left the ranks with different weights"
    assert np.array_equal(g0, g1) and np.array_equal(w0, w1), "ranks diverged after all-reduce + Adam"
    P0 = {k: torch.tensor(v) for k, v in P0.items()}
    g0 = torch.tensor(g0)
    # averaged shard gradients == gradient of the full batch (the loss is a mean over rays)
    n = 32
    rays_all = orc.synthetic_rays(n, seed=4)
    target_all = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))
    full = _oracle_flat_grad(P0, rays_all, target_all)
    assert (g0 - full).abs().max() <= 1e-6 * max(1.0, float(full.abs().max())) + 1e-7
    assert fr0 == list(range(7)) and fr1 == list(range(7))


def test_shard_slice_requires_even_split():
    from nerf_pytorch_amd import parallel
    assert parallel.shard_slice(4096 * 8, 3, 8) == (3 * 4096, 4 * 4096)
    with pytest.raises(ValueError):
        parallel.shard_slice(10, 0, 4)
    assert parallel.frames_of_rank(40, 3, 8) == [3, 11, 19, 27, 35]


def _forced_worker(q):
    """one rank, NERF_FORCE_PROCESS_GROUP semantics (init_distributed(force_group=True)): every collective runs over the one-rank group"""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    import sys
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel
    assert parallel._single() and parallel.ranks_seen() == [0]
    r, w, dev = parallel.init_distributed(backend="gloo", force_group=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1 and not parallel._single()
    render = sys.modules[parallel.__name__.rsplit(".", 1)[0] + ".render"]
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    net = npa.NeRF(**kw)
    net._packed, net._packed_key = {"fp32": "stale repack"}, ("stale",)
    parallel.broadcast_parameters([net])
    assert net._packed is None                      # the broadcast ran (and dropped the cached repack)
    sync = parallel.GradientSync([net])
    flat = torch.arange(npa.hip_backend.N_PARAMS, dtype=torch.float32)
    want = flat.clone()
    render._grad_ready(net, flat)
    assert sync.started == 1 and id(net) in sync.pending        # a real (gloo) work object is pending
    for nm, off, shape in npa.hip_backend.param_table():
        dict(net.named_parameters())[nm].grad = flat[off:off + int(np.prod(shape))].view(shape)
    net.last_flat_grad = flat
    sync.finish()
    sync.close()
    ok = bool(torch.equal(flat, want))               # sum over one rank / 1
    seen, same = parallel.ranks_seen(), parallel.ranks_identical([net.flat_params()])
    dist.destroy_process_group()
    q.put((ok, seen, same))


@pytest.mark.timeout(300)
def test_forced_one_rank_group_runs_the_collectives():
    """NERF_FORCE_PROCESS_GROUP / init_distributed(force_group=True): the world-size-1 short cuts are off (the 1-GPU box executes
    the RCCL branch this way, tests/test_rccl_one_rank_gpu.py; here over gloo)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(q,))
    p.start()
    ok, seen, same = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert ok and seen == [0] and same is True


# ---------------------------------------------------------------- 8 ranks, STRONG semantics of BASELINE configs[3]
N_STRONG, SC_STRONG, SF_STRONG = 64, 8, 8          # the CPU-sized stand-in of 32,768 rays x (64 + 128): 8 rays per rank


def _strong_loss_grads(Pc, Pf, rays, target, rnd):
    """oracle training loss of run_nerf.py:765-771 on (rays, target) with the injected draws; flat gradients of both networks"""
    Pc_g = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
    Pf_g = {k: v.clone().requires_grad_(True) for k, v in Pf.items()}
    out = orc.trace_rays(rays, Pc_g, Pf_g, SC_STRONG, SF_STRONG, perturb=1.0, white_bkgd=False, raw_noise_std=1.0, **rnd)
    (orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)).backward()
    flat = lambda P: torch.cat([P[k].grad.reshape(-1) for k, _ in orc.param_shapes()])
    return flat(Pc_g), flat(Pf_g)


def _strong_inputs():
    rays = orc.synthetic_rays(N_STRONG, seed=14)
    target = torch.rand(N_STRONG, 3, generator=torch.Generator().manual_seed(3))
    return rays, target


def _strong_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel
    torch.set_num_threads(1)
    r, w, dev = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    Pc, Pf = orc.scene_params(seed=2)
    nets = [npa.NeRF(**kw), npa.NeRF(**kw)]
    if rank == 0:                                       # only rank 0 holds the scene; the others get it by broadcast
        nets[0].load_state_dict(Pc)
        nets[1].load_state_dict(Pf)
    parallel.broadcast_parameters(nets)
    P = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in nets]
    rays_all, target_all = _strong_inputs()
    # what bench.py's Session does for --strong: every rank draws the GLOBAL batch's randoms from an identically seeded generator
    # and keeps its rows; rays and targets are sliced the same way
    rnd = parallel.shard_randoms(parallel.global_randoms(N_STRONG, SC_STRONG, SF_STRONG, 1.0, torch.Generator().manual_seed(4242)))
    lo, hi = parallel.shard_slice(N_STRONG, rank, world)
    assert all(v.shape[0] == hi - lo for v in rnd.values()) and set(rnd) == {"t_rand", "noise_c", "u", "noise_f"}
    grads = _strong_loss_grads(P[0], P[1], rays_all[lo:hi], target_all[lo:hi], rnd)
    table = npa.hip_backend.param_table()
    sync = parallel.GradientSync(nets)
    import sys
    render = sys.modules[parallel.__name__.rsplit(".", 1)[0] + ".render"]
    for m, flat in zip(nets, grads):                    # the backward reports each bucket final, then autograd installs the views
        render._grad_ready(m, flat)
        for nm, off, shape in table:
            dict(m.named_parameters())[nm].grad = flat[off:off + int(np.prod(shape))].view(shape)
    assert sync.started == 2
    sync.finish()
    sync.close()
    same = parallel.ranks_identical([m.last_flat_grad for m in nets])
    q.put((rank, grads[0].numpy().copy(), grads[1].numpy().copy(), bool(same)))
    dist.barrier()
    parallel.shutdown()
    assert not dist.is_initialized() and parallel.FORCE_GROUP is False


@pytest.mark.timeout(600)
def test_eight_rank_strong_batch_equals_the_single_process_step():
    """BASELINE configs[3] as bench.py --gpus 8 --strong runs it, on 8 gloo ranks with the oracle as the per-rank gradient: ONE global
    batch, its random draws made once per rank from an identically seeded generator and sliced (parallel.global_randoms /
    shard_randoms), rays and targets sliced with shard_slice, per-rank mean losses, gradients averaged by GradientSync -- equal to the
    single-process gradient of the whole batch with the whole draws.  (The world-2 test above covers the weak form; this is the control
    flow the driver's 8-GPU run executes that no GPU of this build has ever run.)"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=480) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[3] for r in res), "ranks hold different averaged gradients"
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
    Pc, Pf = orc.scene_params(seed=2)
    rays_all, target_all = _strong_inputs()
    from nerf_pytorch_amd import parallel
    rnd_all = parallel.global_randoms(N_STRONG, SC_STRONG, SF_STRONG, 1.0, torch.Generator().manual_seed(4242))
    full_c, full_f = _strong_loss_grads(Pc, Pf, rays_all, target_all, rnd_all)
    for got, want in ((torch.tensor(res[0][1]), full_c), (torch.tensor(res[0][2]), full_f)):
        assert float(want.abs().max()) > 0
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-9, float((got - want).abs().max() / want.abs().max())


# ---------------------------------------------------------------- frame-parallel render_path (BASELINE configs[4]) on two gloo ranks
def _frames_worker(rank, world, port, q, savedir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel
    parallel.init_distributed(backend="gloo")
    render = sys.modules[parallel.__name__.rsplit(".", 1)[0] + ".render"]
    seen = []

    def fake_render(H, W, K, chunk=None, c2w=None, **kw):       # the kernels need a GPU: a frame that encodes its pose
        seen.append(float(c2w[0, 3]))
        rgb = torch.full((H, W, 3), float(c2w[0, 3]) / 10.0)
        return rgb, rgb[..., 0] * 2.0, rgb[..., 0], {}
    render.render = fake_render
    poses = torch.stack([torch.eye(4) for _ in range(5)])
    for i in range(5):
        poses[i, 0, 3] = float(i)
    rgbs, disps = parallel.render_path(poses, (6, 8, 10.0), None, 1 << 15, {}, savedir=savedir)
    q.put((rank, rgbs, disps, seen))
    dist.barrier()
    parallel.shutdown()


@pytest.mark.timeout(300)
def test_frame_parallel_render_path_deals_gathers_and_writes(tmp_path):
    """parallel.render_path (round 6): poses dealt round-robin (frames_of_rank), every rank renders and writes ITS frames under their
    global numbers, rank 0 returns the stacked arrays in pose order, the others (None, None) -- run_nerf.py:137-175's return value
    from G processes.  (The same function with the real kernels, bit-compared with the single-process render_path:
    tests/test_two_ranks_gpu.py.)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, rgbs, disps, seen0), (_, rgbs1, disps1, seen1) = res
    files = sorted(os.listdir(tmp_path))
    assert seen0 == [0.0, 2.0, 4.0] and seen1 == [1.0, 3.0]            # round-robin, every pose exactly once
    assert rgbs1 is None and disps1 is None
    assert rgbs.shape == (5, 6, 8, 3) and disps.shape == (5, 6, 8) and rgbs.dtype == np.float32
    for i in range(5):
        assert np.all(rgbs[i] == np.float32(i / 10.0)) and np.all(disps[i] == np.float32(i / 10.0) * 2)
    assert [f for f in files if f.endswith(".png")] == [f"{i:03d}.png" for i in range(5)]
