"""Timing-only variants of the weight-ring forward (csrc/field_ring.h, field_fwd_ring.hip): patched copies of csrc/ compiled
to nerf-pytorch_amd/libexp_<name>.so (git-ignored; they travel with gpurun).  Most of them compute WRONG results on
purpose: they remove one ingredient (barriers, DMA, fragment reads, operand split, encodings) to see what it costs.
    python tools/ring_variants.py [name ...]      # build (all by default), in parallel
    python tools/exp_ring_variants.py             # on the GPU: time every libexp_*.so"""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-pytorch_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import build as nbuild


def patch(path, old, new):
    s = open(path).read()
    assert s.count(old) >= 1, (path, old[:70])
    open(path, "w").write(s.replace(old, new))


R = lambda d: os.path.join(d, "field_ring.h")
F = lambda d: os.path.join(d, "field_fwd_ring.hip")
DEV = lambda d: os.path.join(d, "field_device.h")


def nobar(d):
    patch(R(d), "        else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();\n    }\n    // behind the MFMAs",
          "        else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    }\n    // behind the MFMAs")


def nodma(d):
    patch(R(d), "        if constexpr (K < 4) fetch_part<K>(dma_chunk);", "        if constexpr (K < 4) { if (dma_chunk < -5) fetch_part<K>(dma_chunk); }")


def nolds(d):      # no fragment requests (the three sets keep the first unit's fragments)
    patch(R(d), "        lo.w[i] = p[(2 * i + 1) * 64];", "        if (reinterpret_cast<uintptr_t>(p) == 0x7fff0) lo.w[i] = p[(2 * i + 1) * 64];")
    patch(R(d), "        nxt.w[i] = pn[(2 * i) * 64];", "        if (reinterpret_cast<uintptr_t>(p) == 0x7fff0) nxt.w[i] = pn[(2 * i) * 64];")
    patch(F(d), "    Frag fa, fb, fl;\n", "    Frag fa, fb, fl;\n    ring.request_first(fa); ring.request_first(fb); ring.request_first(fl);\n")


def halflds(d):     # hi fragments only: half the LDS read traffic
    patch(R(d), "        lo.w[i] = p[(2 * i + 1) * 64];", "        if (reinterpret_cast<uintptr_t>(p) == 0x7fff0) lo.w[i] = p[(2 * i + 1) * 64];")
    patch(F(d), "    Frag fa, fb, fl;\n", "    Frag fa, fb, fl;\n    ring.request_first(fl);\n")


def nosplit(d):
    patch(R(d), "            if constexpr (more && ((G == 4 && t == 0) || (G == 2 && t < 2))) {", "            if constexpr (false) {")


B = lambda d: os.path.join(d, "field_bwd_ring.hip")


def nostore(d):     # dgrad: no delta stores; forward: no row stores
    patch(B(d), "        nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, __builtin_amdgcn_perm(nbr, own, pair_sel));",
          "        if (own == 0x12345678u) nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, __builtin_amdgcn_perm(nbr, own, pair_sel));")
    patch(F(d), "        nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);", "        if (word == 0x12345678u) nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);")


def nolds_b(d):
    nolds(d)
    patch(B(d), "    Frag fa, fb, fl;\n    ring.request_first(fa);", "    Frag fa, fb, fl;\n    ring.request_first(fa); ring.request_first(fb); ring.request_first(fl);")


def noepi(d):       # dgrad: no ReLU-mask arithmetic in the layer epilogue (acc copied as is); forward: no ReLU bitmask construction
    patch(B(d), "        const unsigned bit = (m[i >> 5] >> (i & 31)) & 1u;\n        d[i] = bit ? acc[i >> 4][i & 15] : 0.0f;", "        d[i] = acc[i >> 4][i & 15];")
    nomask(d)


def nomask(d):      # forward: no ReLU bitmask construction (bits per unit, shift / exchange / store per layer)
    patch(F(d), "        unsigned b = h[4 * nb + r] > 0.0f ? 1u << (8 * (nb & 3) + r) : 0u;\n        asm volatile(\"\" : \"+v\"(b));\n        mw[nb >> 2] |= b;",
          "        (void)nb; (void)r;")
    patch(F(d), "    auto finish_mask = [&](int layer) __attribute__((always_inline)) {     // the words save_mask16 builds, bit for bit\n        if (!SAVE) return;",
          "    auto finish_mask = [&](int layer) __attribute__((always_inline)) {\n        return;")


def noencsave(d):   # forward: the xyz encodings are not written
    patch(F(d), "                if (col >= 0) nt_store(reinterpret_cast<__bf16*>(a.act + al.enc)", "                if (col == -77) nt_store(reinterpret_cast<__bf16*>(a.act + al.enc)")


W = lambda d: os.path.join(d, "field_bwd.hip")


def wg_coarse(n):   # bf16 weight-gradient GEMM: n point chunks instead of 39 for launches below 400k points (13 jobs x 19 = one round)
    return lambda d: patch(W(d), "(n_jobs == 13 ? 39 : 64)", "(n_jobs == 13 ? (P < 400000 ? %d : 39) : 64)" % n)


def noenc(d):
    patch(DEV(d), "            sincosf(xv * pow2f(fr), &sn, &cs);\n            e[2 * m] = sn;", "            sn = xv * pow2f(fr); cs = sn * 0.5f;\n            e[2 * m] = sn;")


def prio_b(d):
    patch(F(d), "    ring.ready();\n", "    ring.ready();\n    if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n")


def prio_a(d):
    patch(F(d), "    ring.ready();\n", "    ring.ready();\n    if (wave < 4) __builtin_amdgcn_s_setprio(1);\n")


def nosched(d):     # hipcc's own order inside a unit
    s = open(R(d)).read()
    a = s.index("template <int NB, typename Tail>")
    b = s.index("// NU units of one contraction")
    body = s[a:b].replace("        __builtin_amdgcn_sched_barrier(0);\n", "")
    open(R(d), "w").write(s[:a] + body + s[b:])


def head(d):        # the committed sources (git HEAD) instead of the working tree: A/B of an uncommitted change against "base"
    for f in os.listdir(d):
        r = subprocess.run(["git", "-C", ROOT, "show", f"HEAD:nerf-pytorch_amd/csrc/{f}"], capture_output=True)
        if r.returncode == 0:
            open(os.path.join(d, f), "wb").write(r.stdout)


VARIANTS = {
    "base": [], "head": [head], "nobar": [nobar], "nodma": [nodma], "nolds": [nolds_b], "halflds": [halflds], "nosplit": [nosplit], "noenc": [noenc],
    "prio_b": [prio_b], "prio_a": [prio_a], "nosched": [nosched],
    "mfmaonly": [nobar, nodma, nolds_b, nosplit], "nostore": [nostore], "wg_c19": [wg_coarse(19)], "wg_c26": [wg_coarse(26)], "noepi": [noepi], "noepi_nostore": [noepi, nostore], "mfmaonly_nostore": [nobar, nodma, nolds_b, nosplit, nostore],

    "nobar_nodma": [nobar, nodma], "nomask": [nomask], "noencsave": [noencsave], "nomask_nostore": [nomask, nostore],
    "nosave_at_all": [nomask, nostore, noencsave],
}


def build(name):
    d = os.path.join("/tmp", "rv", name, "csrc")          # api.hip includes ../../include/nerf_hip.h
    shutil.rmtree(os.path.dirname(d), ignore_errors=True)
    shutil.copytree(CSRC, d)
    for fn in VARIANTS[name]:
        fn(d)
    out = os.path.join(ROOT, "nerf-pytorch_amd", f"libexp_{name}.so")
    cmd = [nbuild._hipcc()] + nbuild.FLAGS + [os.path.join(d, s) for s in nbuild.SOURCES] + ["-o", out] + nbuild.LIBS
    os.makedirs("/tmp/rv/include", exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "nerf_hip.h"), "/tmp/rv/include/nerf_hip.h")
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, (r.stdout + r.stderr)[-2000:]


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(max_workers=6) as ex:
        for name, rc, log in ex.map(build, names):
            print(name, "ok" if rc == 0 else "FAILED\n" + log, flush=True)
