"""CPU: C-ABI surface, host-side objects (NeRF module / flat parameter vector / config parser / create_nerf)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import nerf_oracle as orc
import nerf_pytorch_amd as npa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    declared = set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(npa.build.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/nerf_hip.h but not exported"
    assert declared == set(npa.hip_backend.EXPORTS), declared ^ set(npa.hip_backend.EXPORTS)
    L = npa.hip_backend.lib()
    assert L.nerf_abi_version() == npa.hip_backend.ABI_VERSION == 10 and L.nerf_param_count() == 595844
    assert L.nerf_packed_floats() % 4 == 0


def test_argument_errors_are_codes_not_crashes():
    L = npa.hip_backend.lib()
    assert L.nerf_pack_params(None, None, None) == -1
    assert b"null pointer" in L.nerf_last_error()
    assert L.nerf_field_fwd(None, None, 11, None, 4, 4, None, None, None) == -1
    # the pair repack: null pointers, one packed buffer for both networks, an unknown stream bit, the reduced split (its two extra
    # launches are per network: nerf_pack_params_split(split = 2))
    assert L.nerf_pack_params_split_pair(None, None, None, None, 5, 1, None) == -1 and b"null pointer" in L.nerf_last_error()
    assert L.nerf_pack_params_split_pair(0x1000, 0x2000, 0x3000, 0x2000, 5, 1, None) == -1 and b"own packed buffers" in L.nerf_last_error()
    assert L.nerf_pack_params_split_pair(0x1000, 0x2000, 0x3000, 0x4000, 2, 1, None) == -1
    assert L.nerf_pack_params_split_pair(0x1000, 0x2000, 0x3000, 0x4000, 5, 2, None) == -1
    assert L.nerf_act_floats(0, 64) == 0
    for n, S in ((4096, 192), (5, 3)):
        P = n * S
        Pp = (P + 31) // 32 * 32       # the split datapaths save 32-point tiles
        # per-datapath sizes (ABI v9): fp32 rows [P][F] ...
        a0 = P * (9 * 256 + 128 + 64 + 32) + n * 32
        a0 = (a0 + 3) // 4 * 4 + 9 * P * 8
        assert L.nerf_act_floats_dp(n, S, 0) == a0 and L.nerf_delta_floats_dp(n, S, 0) == P * (9 * 256 + 128)
        # ... 16-bit tiles: 8 trunk regions + view branch + encoding + per-point direction tiles at 2 bytes per element, the dump
        # region (8192 words), the per-ray fp32 direction encoding, bitmasks, 2048 words of staging slack
        a1 = Pp * (8 * 256 + 128 + 64) // 2 + 8192 + n * 32 + max(Pp * 32 // 2, n * 32 * 4)
        a1 = (a1 + 3) // 4 * 4 + 9 * P * 8 + 2048
        assert L.nerf_act_floats_dp(n, S, 1) == a1
        assert L.nerf_delta_floats_dp(n, S, 1) == Pp * (8 * 256 + 128 + 4) // 2 + 8192 + 2048 + 4      # + the fp16 split's scale words
        # the two-argument forms: a buffer either datapath may write
        assert L.nerf_act_floats(n, S) == max(a0, a1) and L.nerf_delta_floats(n, S) == max(L.nerf_delta_floats_dp(n, S, 0), L.nerf_delta_floats_dp(n, S, 1))
        # datapath 2 (ABI v10, "fp16x3w"): the same layout twice -- the second copy holds the lo words
        assert L.nerf_act_floats_dp(n, S, 2) == 2 * a1 and L.nerf_delta_floats_dp(n, S, 2) == 2 * L.nerf_delta_floats_dp(n, S, 1)
        assert L.nerf_workspace_floats_dp(n, 64, 128, 1, 2) > L.nerf_workspace_floats_dp(n, 64, 128, 1, 1)
        assert L.nerf_act_floats_dp(n, S, 3) == 0 and L.nerf_delta_floats_dp(n, S, -1) == 0
    assert L.nerf_act_floats_dp(4096, 192, 1) < 0.5 * L.nerf_act_floats_dp(4096, 192, 0)            # 4.8 vs 10.6 KB / point
    assert L.nerf_delta_floats_dp(4096, 192, 1) < 0.5 * L.nerf_delta_floats_dp(4096, 192, 0)
    assert (L.nerf_wgrad_partial_floats(n, S) - (128 * 256 + 128)) % 595844 == 0      # per-chunk partials + fold scratch (G | dbv)
    # round 6: the range scan takes only buffers the library's own fp16 forward / delta chain wrote; the two-word splits are checked
    assert L.nerf_range_scan(None, 4, 4, None, None) == -1 and b"null pointer" in L.nerf_last_error()
    assert L.nerf_range_scan(0x100000, 4, 4, 0x200000, None) == -1 and b"fp16" in L.nerf_last_error()      # an unknown buffer
    assert L.nerf_field_fwd_split(0x1000, 0x2000, 11, 0x3000, 4, 4, 0x4000, None, 4, None) == -1             # split 4 does not exist
    assert L.nerf_field_dgrad_split(0x1000, 0x2000, 0x3000, 4, 4, 0x4000, 2, None) == -1
    assert L.nerf_field_wgrad_phase(0x1000, 0x2000, 0x3000, 4, 4, 0x4000, 0x5000, 0, 7, 7, 0x6000, None) == -1   # datapath 7 does not exist


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(npa.hip_backend, "_LIB", None)
    monkeypatch.setattr(npa.build, "LIB_PATH", "/nonexistent/libnerf_hip.so")
    with pytest.raises(npa.hip_backend.NerfHipError):
        npa.hip_backend.lib()


def test_cpu_tensors_are_rejected_not_silently_computed():
    x = torch.randn(4, 3)
    with pytest.raises(npa.hip_backend.NerfHipError):
        npa.hip_backend.embed(x, 10)


def test_nerf_module_state_dict_and_flat_binding():
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    torch.manual_seed(3)
    m = npa.NeRF(**kw)
    want = [nm for nm, _ in orc.param_shapes()]
    assert list(m.state_dict().keys()) == want
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(s) for _, s in orc.param_shapes()]
    assert m._is_bound()
    P = orc.make_params(5)
    m.load_state_dict(P)
    assert m._is_bound()
    flat = m.flat_params()
    for nm, off, shape in npa.hip_backend.param_table():
        assert torch.equal(flat[off:off + int(np.prod(shape))].view(shape), P[nm])
    # optimizer updates go through to the flat vector
    opt = torch.optim.SGD(m.parameters(), lr=1.0)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(flat[:10], P["pts_linears.0.weight"].reshape(-1)[:10] - 1.0)
    # dtype / device moves re-bind
    m2 = m.to(torch.device("cpu"))
    assert m2._is_bound()
    # same default init as the reference architecture under the same seed
    torch.manual_seed(3)
    ref_first = torch.nn.Linear(63, 256).weight
    torch.manual_seed(3)
    assert torch.equal(npa.NeRF(**kw).pts_linears[0].weight, ref_first)


def test_other_architectures_build_the_layer_by_layer_module():
    """NeRF(...) with any arguments the reference's constructor accepts: the BASELINE architecture is the fused-kernel class,
    everything else a DenseNeRF with the reference's state_dict (names, order, shapes) and default initialisation."""
    import copy
    import workloads as wl
    fused = npa.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    assert type(fused) is npa.NeRF and type(copy.deepcopy(fused)) is npa.NeRF
    for arch in (wl.arch_of(use_viewdirs=False), wl.arch_of(D=6, W=128, multires=6, multires_views=2),
                 wl.arch_of(D=4, W=64, multires=-1, multires_views=-1, output_ch=4), wl.arch_of(W=128)):
        A = {k: arch[k] for k in ("D", "W", "input_ch", "input_ch_views", "output_ch", "skips", "use_viewdirs")}
        torch.manual_seed(3)
        net = npa.NeRF(**A)
        assert type(net).__name__ == "DenseNeRF" and not isinstance(net, npa.NeRF)
        assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == wl.arch_param_shapes(arch)
        assert (net.multires, net.multires_views) == (arch["multires"], arch["multires_views"] if arch["use_viewdirs"] else -1)
        torch.manual_seed(3)
        assert torch.equal(net.pts_linears[0].weight, torch.nn.Linear(arch["input_ch"], arch["W"]).weight)
        assert type(copy.deepcopy(net)).__name__ == "DenseNeRF"
    assert type(npa.NeRF()).__name__ == "DenseNeRF"            # the reference's defaults: D=8, W=256, 3 + 3 inputs, no view directions
    with pytest.raises(ValueError):
        npa.NeRF(D=5, W=64, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True)       # the reference fails at its first forward


def test_config_parser_reads_reference_style_files(tmp_path):
    cfg = tmp_path / "lego.txt"
    cfg.write_text("expname = blender_paper_lego\nbasedir = ./logs\ndatadir = ./data/nerf_synthetic/lego\n"
                   "dataset_type = blender\n\nno_batching = True\n\nuse_viewdirs = True\nwhite_bkgd = True\n"
                   "lrate_decay = 500\n\nN_samples = 64\nN_importance = 128\nN_rand = 1024\n\n"
                   "precrop_iters = 500\nprecrop_frac = 0.5\n\nhalf_res = True\n")
    a = npa.config_parser().parse_args(["--config", str(cfg), "--N_rand", "4096"])
    assert (a.expname, a.N_rand, a.N_importance, a.use_viewdirs, a.white_bkgd, a.half_res, a.no_batching) == \
           ("blender_paper_lego", 4096, 128, True, True, True, True)
    assert (a.netdepth, a.netwidth, a.chunk, a.netchunk, a.multires, a.multires_views, a.perturb, a.lrate) == \
           (8, 256, 32768, 65536, 10, 4, 1.0, 5e-4)
    d = npa.config_parser().parse_args([])
    assert d.N_rand == 4096 and d.dataset_type == "llff" and d.i_weights == 10000 and not d.lindisp


def test_create_nerf_builds_the_reference_kwargs(tmp_path):
    a = npa.config_parser().parse_args(["--expname", "t", "--basedir", str(tmp_path), "--use_viewdirs",
                                       "--N_importance", "128", "--white_bkgd", "--dataset_type", "blender"])
    (tmp_path / "t").mkdir()
    tr, te, start, grad_vars, opt = npa.create_nerf(a, device=torch.device("cpu"))
    assert start == 0 and len(grad_vars) == 48 and isinstance(opt, torch.optim.Adam)
    assert set(tr) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                       "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"}
    assert te["perturb"] is False and te["raw_noise_std"] == 0. and tr["ndc"] is False
    assert isinstance(tr["network_fn"], npa.NeRF) and isinstance(tr["network_fine"], npa.NeRF)
    # checkpoint round trip in the reference's format (run_nerf.py:792-800, reload :216-233)
    torch.save({"global_step": 7, "network_fn_state_dict": tr["network_fn"].state_dict(),
                "network_fine_state_dict": tr["network_fine"].state_dict(),
                "optimizer_state_dict": opt.state_dict()}, tmp_path / "t" / "000007.tar")
    tr2, _, start2, _, _ = npa.create_nerf(a, device=torch.device("cpu"))
    assert start2 == 7
    assert torch.equal(tr2["network_fn"].flat_params(), tr["network_fn"].flat_params())


def test_create_nerf_builds_a_mixed_pair_on_one_path():
    """A legal reference command line whose two networks differ in architecture (--netwidth_fine 128 next to the default
    coarse network, run_nerf.py:435-442): render_rays evaluates a pair on ONE path, so create_nerf builds BOTH layer by layer;
    two default networks stay on the fused kernels; force_dense is not needed for equal pairs."""
    base = ["--expname", "t", "--basedir", "/nonexistent", "--use_viewdirs", "--N_importance", "64", "--no_reload"]
    for extra, kinds in (([], ("NeRF", "NeRF")), (["--netwidth_fine", "128"], ("DenseNeRF", "DenseNeRF")),
                         (["--netdepth", "6"], ("DenseNeRF", "DenseNeRF")), (["--netwidth", "128", "--netwidth_fine", "128"], ("DenseNeRF", "DenseNeRF"))):
        args = npa.config_parser().parse_args(base + extra)
        tr, _, _, grad_vars, _ = npa.create_nerf(args, device=torch.device("cpu"), fused_adam=False)
        assert (type(tr["network_fn"]).__name__, type(tr["network_fine"]).__name__) == kinds, (extra, kinds)
        assert len(grad_vars) == len(list(tr["network_fn"].parameters())) + len(list(tr["network_fine"].parameters()))
    forced = npa.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, force_dense=True)
    fused = npa.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    assert type(forced).__name__ == "DenseNeRF" and type(fused) is npa.NeRF
    assert list(forced.state_dict().keys()) == list(fused.state_dict().keys())


def test_flat_adam_is_state_dict_compatible_with_torch_adam():
    """FlatAdam = torch.optim.Adam arithmetic on flat moment buffers; checkpoints round-trip both ways (run_nerf.py:792-800)."""
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    torch.manual_seed(0)
    a, b = npa.NeRF(**kw), npa.NeRF(**kw)
    b.load_state_dict(a.state_dict())
    oa, ob = npa.FlatAdam(a.parameters(), lr=5e-4), torch.optim.Adam(b.parameters(), lr=5e-4)
    table = npa.hip_backend.param_table()

    def set_grads(seed):
        fg = torch.randn(595844, generator=torch.Generator().manual_seed(seed))
        for m in (a, b):
            for (nm, off, shape), p in zip(table, m.param_list()):
                g = fg[off:off + p.numel()].view(shape)
                p.grad = g if m is a else g.clone()          # a: views of one flat bucket, like the HIP backward
    for it in range(3):
        set_grads(it)
        oa.step()
        ob.step()
    assert torch.equal(a.flat_params(), b.flat_params())
    assert len(npa.optim._segments(list(a.parameters()))) == 1
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys()
    for i in range(24):
        assert torch.equal(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"])
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 3.0
    # cross-loading, then one more identical step
    ob2 = torch.optim.Adam(b.parameters(), lr=5e-4)
    ob2.load_state_dict(sa)
    oa2 = npa.FlatAdam(a.parameters(), lr=5e-4)
    oa2.load_state_dict(sb)
    set_grads(7)
    oa2.step()
    ob2.step()
    assert torch.equal(a.flat_params(), b.flat_params())
    # lr schedule of train() (run_nerf.py:780-784) reaches the fused step through param_groups
    for g in oa2.param_groups:
        g["lr"] = 1e-4
    before = a.flat_params().clone()
    set_grads(8)
    oa2.step()
    assert 0 < (a.flat_params() - before).abs().max() < 2.1e-4


def test_sample_ray_batch_matches_get_rays_on_the_selected_pixels():
    import numpy as np
    H, W, focal = 20, 30, 25.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    pose = torch.tensor([[1.0, 0, 0, 0.1], [0, 0.8, -0.6, 0.2], [0, 0.6, 0.8, 4.0]])
    img = torch.rand(H, W, 3)
    g = torch.Generator().manual_seed(0)
    rays, tgt = npa.sample_ray_batch(H, W, K, pose, img, 64, generator=g)
    assert rays.shape == (2, 64, 3) and tgt.shape == (64, 3)
    ro, rd = npa.get_rays(H, W, K, pose)
    # every sampled ray is a ray of the full grid, at the pixel whose colour was taken, and no pixel repeats
    flat_d, flat_c = rd.reshape(-1, 3), img.reshape(-1, 3)
    idx = [int(((flat_d - rays[1][n]).abs().sum(-1) < 1e-6).nonzero()[0]) for n in range(64)]
    assert len(set(idx)) == 64
    assert torch.equal(flat_c[idx], tgt) and torch.equal(rays[0], ro.reshape(-1, 3)[idx])
    # central precrop (run_nerf.py:738-747)
    rays2, _ = npa.sample_ray_batch(H, W, K, pose, img, 32, precrop_frac=0.5, generator=g)
    dH, dW = int(H // 2 * 0.5), int(W // 2 * 0.5)
    crop = rd[H // 2 - dH:H // 2 + dH, W // 2 - dW:W // 2 + dW].reshape(-1, 3)
    for n in range(32):
        assert ((crop - rays2[1][n]).abs().sum(-1) < 1e-6).any()


def test_frame_sink_orders_frames_and_writes_the_same_pngs(tmp_path):
    """render_path's output side (f-4) on host tensors: worker-thread PNGs == synchronous writer, frames in order."""
    import sys
    R = sys.modules[npa.render.__module__]      # the module (the package exports the function under the same name)
    g = torch.Generator().manual_seed(3)
    frames = [(torch.rand(9, 7, 3, generator=g) * 1.4 - 0.2, torch.rand(9, 7, generator=g)) for _ in range(5)]
    sink = R._FrameSink(str(tmp_path))
    for i, (rgb, disp) in enumerate(frames):
        sink.push(i, rgb, disp)
    rgbs, disps = sink.close()
    assert rgbs.shape == (5, 9, 7, 3) and disps.shape == (5, 9, 7)
    for i, (rgb, disp) in enumerate(frames):
        assert np.array_equal(rgbs[i], rgb.numpy()) and np.array_equal(disps[i], disp.numpy())
        ref = tmp_path / f"ref{i}.png"
        R._write_png(str(ref), R.to8b(rgb.numpy()))
        assert open(tmp_path / f"{i:03d}.png", "rb").read() == open(ref, "rb").read()


def test_precision_selection_and_saved_row_views():
    """set_precision accepts the four datapaths and rejects anything else; saved_rows inverts the tile layouts of
    csrc/nerf_common.h: 16-bit elements (bf16 / fp16 by precision), element (p, f) of an F-wide region at (p/32)*F*32 + f*32 + p%32
    (deltas, encodings) or, for the rows the forward saves, in 16-point tiles with the row16h row order."""
    assert npa.hip_backend.PRECISIONS == ("fp32", "fp16x3", "bf16x3", "fp16_fp8c", "fp16x3w")
    prev = npa.get_precision()
    try:
        for mode in npa.hip_backend.PRECISIONS:
            npa.set_precision(mode)
            assert npa.get_precision() == mode
        for bad in ("fp16", "mixed"):
            with pytest.raises(ValueError):
                npa.set_precision(bad)
    finally:
        npa.set_precision(prev)
    hb = npa.hip_backend
    n_rays, S, P = 10, 7, 70
    reg = hb.buffer_regions(n_rays, S, False)
    flat = torch.arange(reg["total"], dtype=torch.float32)
    assert torch.equal(hb.saved_rows(flat, n_rays, S, "h1", "fp32"), flat[reg["h1"]:reg["h1"] + P * 256].view(P, 256))
    widths = [("h%d" % i, 256) for i in range(8)] + [("hv", 128), ("enc", 64)]
    for precision, dt in (("bf16x3", torch.bfloat16), ("fp16x3", torch.float16)):
        reg, dreg = hb.buffer_regions(n_rays, S, True), hb.buffer_regions(n_rays, S, True, is_delta=True)
        assert all(reg[k] % 4 == 0 and dreg.get(k, 0) % 4 == 0 for k in reg)           # 16-byte aligned regions
        want = {name: torch.randn(P, F).to(dt).float() for name, F in widths}
        wantd = {name: torch.randn(P, F).to(dt).float() for name, F in widths + [("graw", 4)] if name != "enc"}
        act, delta = torch.zeros(reg["total"]), torch.zeros(dreg["total"])
        a16, d16 = act.view(dt), delta.view(dt)
        for name, F in widths + [("graw", 4)]:
            p, f = torch.meshgrid(torch.arange(P), torch.arange(F), indexing="ij")
            tile32 = (p // 32) * F * 32 + f * 32 + p % 32                              # 32-point feature-major tiles
            rowh = (f // 16) * 16 + 8 * ((f >> 1) & 1) + 2 * ((f >> 2) & 3) + (f & 1)     # feature 4q + r at row 8*(r>>1) + 2q + (r&1)
            if name != "graw":
                # rows saved by the forward: 16-point tiles, row16h row order (one paired store instruction = 8 consecutive rows = two
                # full lines); the 64-wide encoding stays in 32-point tiles
                a16[2 * reg[name] + ((p // 16) * F * 16 + rowh * 16 + p % 16 if F in (256, 128) else tile32)] = want[name].to(dt)
            if name != "enc":
                d16[2 * dreg[name] + tile32] = wantd[name].to(dt)
        for name, F in widths:
            assert torch.equal(hb.saved_rows(act, n_rays, S, name, precision), want[name]), name
        for name in wantd:
            assert torch.equal(hb.delta_rows(delta, n_rays, S, name, precision), wantd[name]), name
        # regions do not overlap: every region ends before the next one starts (16-bit elements over P rounded up to a tile)
        Pa = (P + 31) // 32 * 32
        order = sorted((reg[k], k) for k in reg if k not in ("total", "feat", "lo"))
        sizes = {**{n_: Pa * F // 2 for n_, F in widths}, "dir": n_rays * 32, "mask": 9 * P * 8}
        for (o0, k0), (o1, _k1) in zip(order, order[1:]):
            assert o0 + sizes.get(k0, 0) <= o1, (k0, o0, o1)
        assert order[-1][0] + sizes.get(order[-1][1], 0) <= reg["total"]
    # the two-word layout ("fp16x3w"): the one-word layout twice; part="lo" reads the same regions of the second copy
    reg1, reg2 = hb.buffer_regions(n_rays, S, 1), hb.buffer_regions(n_rays, S, 2)
    assert reg2["lo"] == reg1["total"] and reg2["total"] == 2 * reg1["total"] and all(reg2[k] == reg1[k] for k in reg1 if k not in ("total", "lo"))
    dreg1, dreg2 = hb.buffer_regions(n_rays, S, 1, is_delta=True), hb.buffer_regions(n_rays, S, 2, is_delta=True)
    assert dreg2["lo"] == dreg1["total"] and dreg2["total"] == 2 * dreg1["total"] and dreg2["scale"] == dreg1["scale"]
    both = torch.cat([act, 2 * act])                      # (`act` holds the fp16 case's rows: the last pass of the loop above)
    assert torch.equal(hb.saved_rows(both, n_rays, S, "h3", "fp16x3w"), want["h3"])
    lo_view = hb.saved_rows(both, n_rays, S, "h3", "fp16x3w", part="lo")
    assert lo_view.shape == want["h3"].shape and not torch.equal(lo_view, want["h3"])
    with pytest.raises(hb.NerfHipError):
        hb.saved_rows(act, n_rays, S, "h3", "fp16x3", part="lo")
    # one store instruction of that kernel (j fixed, q = 0..3) covers rows {2j, 2j+1} and {8+2j, 8+2j+1}: two full lines
    r16h = npa.hip_backend._row16h
    for r0 in (0, 2):
        rows = sorted(r16h(4 * q + r) for q in range(4) for r in (r0, r0 + 1))
        assert rows == list(range(4 * r0, 4 * r0 + 8))
    assert sorted(r16h(f) for f in range(256)) == list(range(256))


def test_flat_adam_rejects_options_the_fused_kernel_ignores():
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    m = npa.NeRF(**kw)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    for key, val in (("weight_decay", 1e-2), ("amsgrad", True), ("maximize", True)):
        opt = npa.FlatAdam(m.parameters(), lr=1e-3)
        opt.param_groups[0][key] = val
        with pytest.raises(NotImplementedError, match=key):
            opt.step()
    ref = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-2)
    with pytest.raises(NotImplementedError, match="weight_decay"):
        npa.FlatAdam(m.parameters(), lr=1e-3).load_state_dict(ref.state_dict())


def test_sample_ray_batch_rejects_oversized_batches():
    H, W = 8, 10
    K = np.array([[5.0, 0, 5.0], [0, 5.0, 4.0], [0, 0, 1]])
    pose = torch.eye(4)[:3]
    img = torch.rand(H, W, 3)
    npa.sample_ray_batch(H, W, K, pose, img, 80)
    with pytest.raises(ValueError, match="without replacement"):
        npa.sample_ray_batch(H, W, K, pose, img, 81)
    with pytest.raises(ValueError, match="without replacement"):
        npa.sample_ray_batch(H, W, K, pose, img, 21, precrop_frac=0.5)       # crop = 4 x 4... = 2*2 x 2*2 pixels


def test_workloads_are_neutral_and_reproducible():
    """workloads.py imports neither the product nor the oracle; the oracle re-exports the same generators."""
    import workloads as wl
    src = open(os.path.join(ROOT, "workloads.py")).read()
    assert "nerf_oracle" not in src.replace("oracle/nerf_oracle.py", "") and "import nerf_pytorch_amd" not in src
    assert orc.synthetic_rays is wl.synthetic_rays and orc.scene_params is wl.scene_params
    b = wl.fern_batch(64, seed=3)
    assert b.shape == (2, 64, 3) and bool((b[1][:, 2] < 0).all())
    flat = orc.assemble_render_rays(wl.FERN["H"], wl.FERN["W"], wl.intrinsics(wl.FERN), b[0], b[1], True, 0., 1.)
    assert torch.allclose(flat[:, 2], torch.full((64,), -1.0), atol=1e-5)      # NDC: origins on the near plane z = -1
    sc = wl.blender_scene(H=16, W=16, n_train=3, n_test=1)
    assert sc["images"].shape == (5, 16, 16, 3) and len(sc["i_split"][0]) == 3 and float(sc["images"].min()) >= 0.0
    g = np.load(os.path.join(ROOT, "tests", "golden", "gate_lego.npz"))
    assert abs(float(wl.lego_batch(1024, seed=31).double().abs().sum()) - float(g["rays_checksum"])) < 1e-5
    assert float(g["target_psnr_db"]) >= 30.0


def test_blender_scene_round_trips_through_the_on_disk_format(tmp_path):
    import json
    import workloads as wl
    sc = wl.blender_scene(H=12, W=12, n_train=2, n_test=1)
    wl.write_blender_scene(sc, str(tmp_path))
    meta = json.load(open(tmp_path / "transforms_train.json"))
    assert len(meta["frames"]) == 2 and abs(meta["camera_angle_x"] - sc["camera_angle_x"]) < 1e-12
    assert np.allclose(np.array(meta["frames"][1]["transform_matrix"]), sc["poses"][1].numpy())
    assert open(tmp_path / "train" / "r_0.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"


def test_workspace_size_query_and_lease_pool():
    """nerf_workspace_floats = act(coarse) + act(fine) + delta(larger) + partial(larger); 0 for inference; the Python
    pool re-issues a returned buffer (same pointer) and never hands out one that is still leased."""
    L = npa.hip_backend.lib()
    for n, sc, nf in ((4096, 64, 128), (1024, 64, 0), (7, 5, 3)):
        big = sc + nf
        want = L.nerf_act_floats(n, sc) + (L.nerf_act_floats(n, big) if nf else 0) + L.nerf_delta_floats(n, big) + \
            L.nerf_wgrad_partial_floats(n, big)
        assert L.nerf_workspace_floats(n, sc, nf, 1) == want == npa.hip_backend.workspace_floats(n, sc, nf)
        assert L.nerf_workspace_floats(n, sc, nf, 0) == 0
    assert L.nerf_workspace_floats(0, 64, 128, 1) == 0
    hb = npa.hip_backend
    for dp, prec in ((0, "fp32"), (1, "fp16x3"), (1, "bf16x3"), (1, "fp16_fp8c")):
        want = L.nerf_act_floats_dp(4096, 64, dp) + L.nerf_act_floats_dp(4096, 192, dp) + L.nerf_delta_floats_dp(4096, 192, dp) + L.nerf_wgrad_partial_floats(4096, 192)
        assert L.nerf_workspace_floats_dp(4096, 64, 128, 1, dp) == want == hb.workspace_floats(4096, 64, 128, True, prec)
        assert hb.act_floats(4096, 192, prec) == L.nerf_act_floats_dp(4096, 192, dp) and hb.delta_floats(4096, 192, prec) == L.nerf_delta_floats_dp(4096, 192, dp)
    # the split datapaths keep twice the rays of the fp32 datapath per launch under the same budget, and the 32,768-ray batch of
    # configs[3] stays under 50 GB of saved activations (fp32 rows: ~90 GB)
    assert hb.max_saved_rays(64, 128, "fp16x3") >= 20480 and hb.max_saved_rays(64, 128, "fp32") >= 8192
    assert hb.saved_bytes(32768, 64, 128, "fp16x3") <= 50e9 < hb.saved_bytes(32768, 64, 128, "fp32")
    with pytest.raises(ValueError):
        hb.act_floats(4, 4, "fp64")
    # the default budget (48 GiB) covers every N_rand of the BASELINE configs without recomputation
    assert npa.hip_backend.max_saved_rays(64, 128) >= 8192 and npa.hip_backend.max_saved_rays(64, 128) % 1024 == 0
    # ... and the 32,768-ray chunk of configs[3] keeps the saved activations of its four sub-chunks resident
    assert 4 * npa.hip_backend.saved_bytes(8192, 64, 128) <= npa.hip_backend.SAVE_TOTAL_BYTES < 288e9
    w = npa.hip_backend.Workspace()
    a = w.take(1000, "cpu")
    b = w.take(1000, "cpu")
    assert a.data_ptr() != b.data_ptr()
    w.give(a)
    c = w.take(900, "cpu")
    assert c.data_ptr() == a.data_ptr()          # re-issued; b is still leased and was not touched
    w.give(c); w.give(c); w.give(None)
    assert len(w._free["cpu"]) == 1
    big = w.take(10_000_000, "cpu")              # a much larger request does not squat in a small buffer, nor vice versa
    assert big.numel() == 10_000_000
    w.give(big)
    assert w.take(1000, "cpu").data_ptr() == a.data_ptr()


def test_batchify_rays_slices_injected_randoms(monkeypatch):
    """render(chunk < N) with injected randoms: every chunk must consume ITS rows of the random tensors
    (run_nerf.py:54-66 slices the rays; the draws of a chunk belong to its rays)."""
    import sys
    import torch
    import nerf_pytorch_amd  # noqa: F401
    render = sys.modules["nerf_pytorch_amd.render"]
    seen = []

    def fake_render_rays(ray_batch, **kw):
        seen.append((ray_batch.clone(), {k: v.clone() for k, v in kw["randoms"].items()}))
        return {"rgb_map": ray_batch[:, :3]}
    monkeypatch.setattr(render, "render_rays", fake_render_rays)
    rays = torch.arange(10 * 11, dtype=torch.float32).reshape(10, 11)
    rnd = {"t_rand": torch.arange(10 * 4, dtype=torch.float32).reshape(10, 4), "u": torch.arange(10 * 6, dtype=torch.float32).reshape(10, 6)}
    out = render.batchify_rays(rays, chunk=4, randoms=rnd)
    assert [r.shape[0] for r, _ in seen] == [4, 4, 2]
    for i, (r, kw) in enumerate(seen):
        assert torch.equal(kw["t_rand"], rnd["t_rand"][4 * i:4 * i + 4]) and torch.equal(kw["u"], rnd["u"][4 * i:4 * i + 4])
    assert torch.equal(out["rgb_map"], rays[:, :3])
    import pytest
    with pytest.raises(ValueError):
        render.batchify_rays(rays, chunk=4, randoms={"t_rand": rnd["t_rand"][:7]})


def test_buffer_tags_refuse_mismatched_pairings():
    """C ABI: the library records which layout its forward / dgrad entry points wrote into a scratch buffer and refuses a
    dgrad or weight-gradient call that pairs buffers of different datapaths (NERF_E_BADARG) instead of computing garbage.
    Host-only: with n_rays = 0 no kernel is launched and no pointer is dereferenced, so fake (aligned) addresses do."""
    import ctypes
    import nerf_pytorch_amd as npa
    hb = npa.hip_backend
    L = hb.lib()
    packed, rays, z, raw, act, act2, delta, d_raw, partial, grad, params = (0x10000 * (k + 1) for k in range(11))
    kind = lambda buf: L.nerf_buffer_layout(buf, None, None, None)
    assert kind(act) == -1                                                   # unknown buffer
    assert L.nerf_field_fwd_split(packed, rays, 11, z, 0, 64, raw, act, 0, None) == 0
    assert kind(act) == 4                                                    # rows in 16-point bf16 tiles
    assert L.nerf_field_fwd(packed, rays, 11, z, 0, 64, raw, act2, None) == 0
    assert kind(act2) == 0                                                   # fp32 point-major rows
    # a split-bf16 dgrad on the fp32 forward's save buffer (other bitmask order, other layout): refused
    assert L.nerf_field_dgrad_split(packed, act2, d_raw, 0, 64, delta, 0, None) == -1
    assert b"exact-fp32 forward" in L.nerf_last_error()
    assert L.nerf_field_dgrad(packed, act, d_raw, 0, 64, delta, None) == -1
    # other sample count than the forward's: refused
    assert L.nerf_field_dgrad_split(packed, act, d_raw, 0, 32, delta, 0, None) == -1
    # bf16 rows + fp16 deltas: no weight-gradient datapath contracts that pair
    assert L.nerf_field_dgrad_split(packed, act, d_raw, 0, 64, delta, 1, None) == 0
    is_delta = ctypes.c_int(0)
    assert L.nerf_buffer_layout(delta, ctypes.byref(is_delta), None, None) == 3 and is_delta.value == 1
    assert L.nerf_field_wgrad_phase(act, delta, d_raw, 0, 64, partial, grad, 0, -1, 7, params, None) == -1
    assert b"different datapaths" in L.nerf_last_error()
    # the matching pair: datapath -1 resolves to 4; an explicit other datapath is refused
    assert L.nerf_field_dgrad_split(packed, act, d_raw, 0, 64, delta, 0, None) == 0
    assert L.nerf_field_wgrad_phase(act, delta, d_raw, 0, 64, partial, grad, 0, -1, 7, params, None) == 0
    assert L.nerf_field_wgrad_phase(act, delta, d_raw, 0, 64, partial, grad, 0, 4, 7, params, None) == 0
    assert L.nerf_field_wgrad_phase(act, delta, d_raw, 0, 64, partial, grad, 0, 5, 7, params, None) == -1
    assert b"datapath 5 requested" in L.nerf_last_error()
    assert L.nerf_field_wgrad_phase(act, delta, d_raw, 0, 64, partial, grad, 0, 2, 7, params, None) == -1          # no such datapath any more
    # act and delta swapped
    assert L.nerf_field_wgrad_phase(delta, act, d_raw, 0, 64, partial, grad, 0, -1, 7, params, None) == -1
    # buffers the library never wrote are not checked (datapath must then be given)
    assert L.nerf_field_wgrad_phase(0x900000, 0xA00000, d_raw, 0, 64, partial, grad, 0, 5, 7, params, None) == 0
    assert L.nerf_field_wgrad_phase(0x900000, 0xA00000, d_raw, 0, 64, partial, grad, 0, -1, 7, params, None) == -1


# ---------------------------------------------------------------- build hygiene: no heavy kernel spills
HEAVY_KERNELS = ["field_fwd16r_kernel<2, nerf::SplitF16, false>", "field_fwd16r_kernel<0, nerf::SplitF16, false>",
                 "field_fwd16r_kernel<0, nerf::SplitF16, true>", "field_fwd16r_last2_kernel<nerf::SplitF16>",
                 "field_dgrad3r_kernel<nerf::SplitF16, false>", "wgrad1_kernel<nerf::SplitF16, 1>", "wgrad256_kernel", "wgrad_kernel",
                 # the two-word ("fp16x3w") forms of the three heavy kernels
                 "field_fwd16r_kernel<3, nerf::SplitF16, false>", "field_dgrad3r_kernel<nerf::SplitF16, true>", "wgrad1_kernel<nerf::SplitF16, 3>",
                 "render_infer_kernel<16, nerf::SplitF16>", "field_fwd_kernel<true>", "field_fwd_kernel<false>", "field_dgrad_kernel"]


def test_no_heavy_kernel_spills():
    """hipcc's -Rpass-analysis=kernel-resource-usage table of the build that produced the library (build.py records it next to the
    library at every build): every MFMA kernel of the three datapaths has ScratchSize 0 and no spilled registers, and the two-waves-
    per-SIMD kernels stay at 2 (round 4's saving forward sat at 256 VGPRs + 44 B/lane of scratch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_nerf_build", os.path.join(ROOT, "nerf-pytorch_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    table = b.resource_usage()
    assert table, "libnerf_hip.resources.json missing: the library was not built by build.py"
    for want in HEAVY_KERNELS:
        rows = {k: v for k, v in table.items() if want in k}
        assert rows, f"{want}: not in the resource table ({len(table)} kernels)"
        for k, v in rows.items():
            assert v["scratch_bytes_per_lane"] == 0, (k, v)
            assert v.get("vgpr_spills", 0) == 0, (k, v)      # (SGPR "spills" go to VGPR lanes, not to memory: counted in vgprs)
            assert v["vgprs"] + v["agprs"] <= 512, (k, v)
            if "dgrad3r" not in k:              # the delta chain runs one wave per SIMD by design (458 registers)
                assert v["occupancy_waves_per_simd"] >= 2, (k, v)
    spilling = sorted(k for k, v in table.items() if v.get("scratch_bytes_per_lane", 0))
    assert not spilling, spilling


def test_pytest_det_samples_are_the_references_numpy_linspace(monkeypatch):
    """render_rays(pytest=True, perturb=0): the reference builds the deterministic CDF samples with np.linspace in float64 and casts
    them (run_nerf_helpers.py:213-215).  That is NOT torch.linspace's fp32 sequence (round 5 assumed it was): they differ by one ulp
    in some entries for most sample counts -- so the pytest + det branch hands the kernel the reference's numbers explicitly, and the
    plain det branch (pytest=False: helpers:205 is torch.linspace in the reference too) keeps torch.linspace."""
    differing = {}
    for n in (2, 3, 64, 128):
        ours = torch.linspace(0.0, 1.0, steps=n, dtype=torch.float32)
        theirs = torch.Tensor(np.linspace(0.0, 1.0, n))            # float64 -> float32, as torch.Tensor(ndarray) does in the reference
        differing[n] = int((ours != theirs).sum())
        assert float((ours - theirs).abs().max()) <= 2.0 ** -24
        assert torch.equal(ours, torch.linspace(0., 1., steps=n))   # t_vals / det u of the reference (run_nerf.py:357, helpers:205)
    assert differing[2] == 0 and differing[3] == 0 and differing[64] > 0 and differing[128] > 0, differing
    import sys
    render = sys.modules["nerf_pytorch_amd.render"]
    seen = {}

    class _Stop(Exception):
        pass

    def spy(cfg, rays, rnd, *rest):
        seen.update(rnd)
        raise _Stop()
    monkeypatch.setattr(render._RenderRays, "apply", staticmethod(spy))
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    net = npa.NeRF(**kw)
    for pytest_flag, expect_u in ((True, True), (False, False)):
        seen.clear()
        with pytest.raises(_Stop):
            render.render_rays(torch.zeros(5, 11), net, None, 64, N_importance=128, perturb=0., pytest=pytest_flag)
        assert ("u" in seen) == expect_u
        if expect_u:
            assert torch.equal(seen["u"], torch.Tensor(np.linspace(0., 1., 128)).expand(5, 128))


def test_network_query_fn_selects_the_fused_or_the_hooked_path(monkeypatch):
    """None and create_nerf's own function are the stock query (fused kernels); any other callable is a user hook that render_rays
    calls per pass (run_nerf.py:385, :401) -- never silently ignored."""
    import sys
    render = sys.modules["nerf_pytorch_amd.render"]
    a = npa.config_parser().parse_args(["--use_viewdirs", "--N_importance", "128"])
    a.basedir = a.expname = None
    tr = npa.create_nerf(a, device=torch.device("cpu"))[0]
    assert render._is_builtin_query(None) and render._is_builtin_query(tr["network_query_fn"])
    mine = lambda pts, viewdirs, net: None
    assert not render._is_builtin_query(mine)
    went = []
    monkeypatch.setattr(render, "_render_rays_hooked", lambda *a_, **k_: went.append(a_[3]) or {"rgb_map": None})
    rays = torch.zeros(4, 11)
    assert render.render_rays(rays, tr["network_fn"], mine, 8) == {"rgb_map": None} and went == [mine]


def test_range_monitor_reads_the_words_and_warns_once_per_worsening():
    """hip_backend.RangeMonitor.poll() on fabricated result words (the scan itself needs a GPU: tests/test_gpu_fp16x3.py): fp16 patterns
    -> values, a warning at >= 32768 naming set_precision("bf16x3"), again only when it got worse, NaN patterns -> inf, the deltas' words
    reported separately."""
    import warnings
    hb = npa.hip_backend

    class Done:
        def query(self):
            return True

        def synchronize(self):
            pass
    m = hb.RangeMonitor()
    words = lambda *w: torch.tensor(list(w), dtype=torch.int32)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        m.inflight.append((words(0, 0x4300, 0, 0x4c00), Done()))          # 3.5 and 16: healthy
        m.poll()
        assert not caught and m.max_seen == 3.5 and m.max_delta == 16.0
        m.inflight.append((words(1, 0x7900, 0, 0x4c00), Done()))          # an activation of 40960
        m.poll()
        assert len(caught) == 1 and "activation" in str(caught[0].message) and 'set_precision("bf16x3")' in str(caught[0].message)
        m.inflight.append((words(1, 0x7900, 0, 0x4c00), Done()))          # the same again: no second warning
        m.poll()
        assert len(caught) == 1
        m.inflight.append((words(1, 0x7a00, 1, 0x7c00), Done()))          # worse, and a delta at inf
        m.poll()
        assert len(caught) == 3 and "scaled delta" in str(caught[2].message)
        m.inflight.append((words(1, 0x7e00, 0, 0), Done()))               # a NaN pattern among the rows: the cliff itself
        m.poll()
    rep = m.report()
    assert rep["max_activation"] == float("inf") and rep["max_scaled_delta"] == float("inf") and rep["warnings"] == 4
    assert rep["warn_at"] == 32768.0 and rep["limit"] == 65504.0
    off = hb.RangeMonitor()
    off.every = 0
    off.after_forward([], 0)                                               # switched off: nothing is scanned, nothing is counted
    assert off.calls == 0 and not off.inflight
