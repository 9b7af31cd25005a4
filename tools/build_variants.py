"""Build experimental variants of libnerf_hip.so (timing probes, not product code): patched copies of csrc/ compiled to
nerf-pytorch_amd/libexp_<name>.so (git-ignored, shipped by gpurun); tools/exp_fwd3.py <libname> times them."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-pytorch_amd", "csrc")
SOURCES = ["api.hip", "pack.hip", "ray_ops.hip", "field_fwd.hip", "field_bwd.hip", "field_fwd_bf16.hip", "field_bwd_bf16.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

def patch(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) >= 1, (path, old[:60])
    open(path, "w").write(s.replace(old, new) if count == 0 else s.replace(old, new, count))

VARIANTS = {
    "base": lambda d: None,
    # weight-gradient staging with the round-1 lane map (2-way ds_write_b64 bank conflicts)
    "oldwg": lambda d: open(os.path.join(d, "field_bwd.hip"), "w").write(
        subprocess.run(["git", "-C", ROOT, "show", "18bf52c:nerf-pytorch_amd/csrc/field_bwd.hip"], capture_output=True, text=True, check=True).stdout),
    # forward: no ReLU bitmasks (upper bound of what cheaper mask construction can buy)
    "nomask": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (with_mask) {\n            unsigned w[4]", "        if (false) {\n            unsigned w[4]"),
    # forward: no row stores (masks kept)
    "norows": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (valid) {\n#pragma unroll\n            for (int nb = 0; nb < 16; ++nb)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) store_val(region, W,", "        if (false) {\n#pragma unroll\n            for (int nb = 0; nb < 16; ++nb)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) store_val(region, W,"),
    # 16-point forward, bf16 rows: no row stores (pack + exchange kept) / no pack, exchange or stores / no ReLU bitmasks
    "f16_norows": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (pair_valid) nt_store(tile_base + (16 * nb + r0) * 8 + lane_pair_off, word);", "        if (word == 0x12345678u) nt_store(tile_base + (16 * nb + r0) * 8 + lane_pair_off, word);"),
    "f16_nosave": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (SAVE == 2) { store_pair(region, W, nb, r0, h[4 * nb + r0], h[4 * nb + r0 + 1]); return; }", "        if (SAVE == 2) return;"),
    "f16_nomask": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (!SAVE) return;\n        unsigned w[4] = {0u, 0u, 0u, 0u};", "        return;\n        unsigned w[4] = {0u, 0u, 0u, 0u};"),
    # bf16 weight-gradient GEMM: operand DMA without the nt hint
    "wg1_plain": lambda d: patch(os.path.join(d, "field_bwd.hip"), '"global_load_lds_dwordx4 %1, %2 nt\\n\\t"', '"global_load_lds_dwordx4 %1, %2\\n\\t"'),
    # paired bf16 stores without their per-lane predicate (exec-mask branch around every store): valid for P % 128 == 0 only
    "nopred": lambda d: (patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (pair_valid) nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);", "        nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);"),
                         patch(os.path.join(d, "field_device_bf16.h"), "            if (pair_valid) nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, word);", "            nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, word);")),
    # bf16 weight-gradient GEMM: 39 instead of 59 point chunks (13 jobs x 39 = 507 workgroups = 2 rounds)
    "wg1_39": lambda d: patch(os.path.join(d, "field_bwd.hip"), "(n_jobs == 13 ? 59 : 64)", "(n_jobs == 13 ? 39 : 64)"),
    # field kernels without their weight DMA after the first chunks (WRONG results: timing only -- what the L2->LDS stream costs)
    "nodma": lambda d: patch(os.path.join(d, "field_device.h"), "        if (nf > 0) dma_chunk<NWAVES>(next_src, lds + (buf ^ 1) * CHUNK_FLOATS, nf, wave, lane);", "        if (nf > 0 && c_next < 2) dma_chunk<NWAVES>(next_src, lds + (buf ^ 1) * CHUNK_FLOATS, nf, wave, lane);"),
    # weight DMA without saving / restoring M0 around every 4 KiB (M0 declared clobbered instead)
    "dma_m0": lambda d: dma_m0(d),
    # dgrad: no delta stores
    "dgrad_nostore": lambda d: patch(os.path.join(d, "field_bwd_bf16.hip"), "    auto store_q = [&](auto part, size_t off) {\n        if (!valid) return;", "    auto store_q = [&](auto part, size_t off) {\n        return;"),
    # weight-gradient GEMM: the two waves of a SIMD in anti-phase (waves 0-3 MFMA then stage, waves 4-7 stage then MFMA)
    "wg_anti": lambda d: wg_anti(d, prio=False),
    "wg_anti_prio": lambda d: wg_anti(d, prio=True),
    # s_setprio around the MFMA phase / around the staging phase
    "wg_prio_mfma": lambda d: wg_prio(d, True),
    "wg_prio_stage": lambda d: wg_prio(d, False),
}

def dma_m0(d):
    f = os.path.join(d, "field_device.h")
    src = open(f).read()
    old4 = src[src.index("__device__ inline void dma_4k("):src.index("// every wave copies one contiguous span of the chunk")]
    new4 = '''__device__ inline void dma_4k(const float* gsrc_lane, unsigned lds_dst_uniform) {
    asm volatile(
        "s_mov_b32 m0, %1\\n\\t"
        "s_nop 0\\n\\t"
        "global_load_lds_dwordx4 %0, off\\n\\t"
        "global_load_lds_dwordx4 %0, off offset:1024\\n\\t"
        "global_load_lds_dwordx4 %0, off offset:2048\\n\\t"
        "global_load_lds_dwordx4 %0, off offset:3072"
        :
        : "v"(gsrc_lane), "s"(lds_dst_uniform)
        : "memory", "m0");
}
'''
    open(f, "w").write(src.replace(old4, new4))

def wg_anti(d, prio):
    # (an if/else with both orders spills 1.6k VGPRs; two predicated copies of the staging around one compute do not)
    f = os.path.join(d, "field_bwd.hip")
    hi, lo = ("__builtin_amdgcn_s_setprio(1); ", " __builtin_amdgcn_s_setprio(0);") if prio else ("", "")
    patch(f, """        compute(0);
        if (st + 1 < n_stages) swrite_from(rv, 1, st + 1);
        __syncthreads();""", f"""        if (sop != 0 && st + 1 < n_stages) {{ {hi}swrite_from(rv, 1, st + 1);{lo} }}
        compute(0);
        if (sop == 0 && st + 1 < n_stages) {{ {hi}swrite_from(rv, 1, st + 1);{lo} }}
        __syncthreads();""")
    patch(f, """        compute(1);
        if (st + 2 < n_stages) swrite_from(rw, 0, st + 2);
        __syncthreads();""", f"""        if (sop != 0 && st + 2 < n_stages) {{ {hi}swrite_from(rw, 0, st + 2);{lo} }}
        compute(1);
        if (sop == 0 && st + 2 < n_stages) {{ {hi}swrite_from(rw, 0, st + 2);{lo} }}
        __syncthreads();""")

def wg_prio(d, mfma):
    f = os.path.join(d, "field_bwd.hip")
    a, b = ("1", "0") if mfma else ("0", "1")
    patch(f, """        compute(0);
        if (st + 1 < n_stages) swrite_from(rv, 1, st + 1);""", f"""        __builtin_amdgcn_s_setprio({a}); compute(0); __builtin_amdgcn_s_setprio({b});
        if (st + 1 < n_stages) swrite_from(rv, 1, st + 1);""")
    patch(f, """        compute(1);
        if (st + 2 < n_stages) swrite_from(rw, 0, st + 2);""", f"""        __builtin_amdgcn_s_setprio({a}); compute(1); __builtin_amdgcn_s_setprio({b});
        if (st + 2 < n_stages) swrite_from(rw, 0, st + 2);""")

def main(names):
    inc = os.path.join(ROOT, "include")
    for name in names:
        d = os.path.join("/tmp/exp_src", name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(CSRC, d)
        VARIANTS[name](d)
        # the sources include "../../include/nerf_hip.h": give the copy the same relative layout
        os.makedirs(os.path.join("/tmp/exp_src", "..", "include"), exist_ok=True)
        out = os.path.join(ROOT, "nerf-pytorch_amd", f"libexp_{name}.so")
        src = [os.path.join(d, s) for s in SOURCES]
        for s in src:
            patch_inc = open(s).read().replace('"../../include/nerf_hip.h"', f'"{inc}/nerf_hip.h"')
            open(s, "w").write(patch_inc)
        res = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + src + ["-o", out], capture_output=True, text=True)
        print(name, "->", out if res.returncode == 0 else res.stderr[-2000:])

if __name__ == "__main__":
    main(sys.argv[1:] or list(VARIANTS))
