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
