"""Bare MFMA streams of the field kernels' shapes (tools/probe/mfma_probe.hip): ms, share of the 2.5 PFLOP/s bf16 peak and
cycles per MFMA at 2.4 GHz, with random operand data (zeros clock ~20 % higher: MI355X_MICROARCH.md, DVFS)."""
import ctypes, os, time, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libmfma_probe.so"))
L.probe_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
out = torch.zeros(64, device=dev)
blocks = 6144
for label, data in (("random bf16 pairs", (torch.randn(4096 * 4, device=dev).bfloat16().view(torch.int16).to(torch.int32) & 0xffff) * 65537),
                    ("zeros", torch.zeros(4096 * 4, device=dev, dtype=torch.int32))):
    data = data.to(torch.int32).contiguous()
    for mode, name, units, flop in ((0, "32x32x16, 1 wave/SIMD (dgrad shape)", 240, 32 * 32 * 16 * 2), (1, "16x16x32, 2 waves/SIMD (forward shape)", 260, 16 * 16 * 32 * 2)):
        for fill in (0, 1, 2):
            def run():
                assert L.probe_mfma(mode, fill, data.data_ptr(), out.data_ptr(), blocks, units, s) == 0
            for _ in range(2): run()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): run()
            torch.cuda.synchronize(); t = (time.perf_counter() - t) / 5
            waves = blocks * (4 if mode == 0 else 8)
            mf = waves * units * 12
            print(f"{label:18s} {name:40s} fill {fill}: {t * 1e3:7.3f} ms  {mf * flop / t / 1e15:5.2f} PFLOP/s = {mf * flop / t / 2.5e15:5.1%} of 2.5 PF; "
                  f"{t * 2.4e9 / (mf / 1024):5.1f} clk(2.4 GHz) per MFMA per SIMD", flush=True)

# ---- what a tf32-class datapath would issue: main term on bf16, both correction terms on block-scaled fp8 (K = 128 per MFMA)
L.probe_mfma8.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
data = (torch.randn(4096 * 4, device=dev).bfloat16().view(torch.int16).to(torch.int32) & 0xffff) * 65537
data = (data & 0x7e7e7e7e).to(torch.int32).contiguous()         # as fp8 e4m3 bytes: finite values only
units = 260                                                     # the forward's unit count; one group = 4 k-steps x 4 units
groups = units // 16
for fill in (0, 1, 2):
    def run8():
        assert L.probe_mfma8(fill, data.data_ptr(), out.data_ptr(), blocks, groups, s) == 0
    def run1():
        assert L.probe_mfma(1, fill, data.data_ptr(), out.data_ptr(), blocks, groups * 16, s) == 0
    res = {}
    for name, fn in (("bf16 x3 (MODE 1)", run1), ("bf16 + 2 x fp8 K128", run8)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t) / 5
    a, b = res["bf16 x3 (MODE 1)"], res["bf16 + 2 x fp8 K128"]
    print(f"same products, fill {fill}: bf16x3 {a * 1e3:7.3f} ms   bf16 main + fp8 corrections {b * 1e3:7.3f} ms   ratio {b / a:.3f}", flush=True)
