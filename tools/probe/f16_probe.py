"""fp16 three-term split, hardware questions (tools/probe/f16_probe.hip): subnormal operands in the f16 MFMAs, the
conversions' rounding / subnormal / overflow behaviour, and rate + board power of an f16 stream next to the bf16 stream of the
same shape.  Prints a text report (copied to profiles/r04_f16_probe.txt)."""
import ctypes, os, subprocess, threading, time
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libf16_probe.so"))
vp = ctypes.c_void_p
L.probe_f16_values.argtypes = [vp, ctypes.c_int, vp, vp]
L.probe_f16_cvt.argtypes = [vp, ctypes.c_int, vp, vp]
L.probe_f16_stream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp]
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream


def h2f(bits):
    return float(np.array([bits], dtype=np.uint16).view(np.float16)[0])


print("== 1. operands through v_mfma_f32_16x16x32_f16 (K = 32) and v_mfma_f32_32x32x16_f16 (K = 16): A = a everywhere, B = b everywhere")
cases = [(0x3C00, 0x3C00, "1 * 1 (control)"), (0x0400, 0x3C00, "min normal 2^-14 * 1"), (0x03FF, 0x3C00, "largest subnormal * 1"),
         (0x0010, 0x3C00, "subnormal 2^-20 * 1"), (0x0001, 0x6400, "smallest subnormal 2^-24 * 1024"), (0x0010, 0x0010, "2^-20 * 2^-20 (both subnormal)"),
         (0x3C00, 0x0001, "1 * 2^-24 (B subnormal)"), (0x7BFF, 0x7BFF, "65504 * 65504"), (0x7C00, 0x3C00, "inf * 1"), (0x8010, 0x3C00, "-2^-20 * 1")]
ct = torch.tensor([[a, b] for a, b, _ in cases], dtype=torch.int32, device=dev).contiguous()
out = torch.zeros(len(cases), 4, device=dev)
assert L.probe_f16_values(ct.data_ptr(), len(cases), out.data_ptr(), s) == 0
torch.cuda.synchronize()
flushed = False
for (a, b, name), o in zip(cases, out.cpu().tolist()):
    fa, fb = h2f(a), h2f(b)
    e32, e16 = 32 * fa * fb, 16 * fa * fb
    ok = (o[0] == e32 or (np.isinf(e32) and np.isinf(o[0]))) and (o[1] == e16 or (np.isinf(e16) and np.isinf(o[1])))
    if not ok:
        flushed = True
    print(f"  {name:36s} a={fa:.6g} b={fb:.6g}: 16x16x32 -> {o[0]:.9g} (exact {e32:.9g})  32x32x16 -> {o[1]:.9g} (exact {e16:.9g})  {'OK' if ok else 'DIFFERS'}")
print(f"  => f16 MFMA subnormal operands: {'FLUSHED or inexact' if flushed else 'honoured exactly (no flush)'}")

print("== 2. v_cvt_pk_f16_f32 / v_fma_mix_f32 (a - f32(hi)) / second conversion: against numpy float16 (round to nearest even)")
vals = np.array([1.0, 0.1, 3.1415927, 3e-6, 6.0e-5, 6.2e-5, 1e-7, 2.98e-8, 2.99e-8, 8.95e-8, 65504.0, 65519.9, 65520.0, 1e5, -0.3333333, 1.00048828125, 1.000732421875, 5.9604645e-08, 1e-9],
                dtype=np.float32)
rng = np.random.default_rng(0)
vals = np.concatenate([vals, (rng.standard_normal(4096) * np.exp(rng.uniform(-18, 8, 4096))).astype(np.float32)])
vin = torch.from_numpy(vals).to(dev)
vo = torch.zeros(len(vals), 4, dtype=torch.int32, device=dev)
assert L.probe_f16_cvt(vin.data_ptr(), len(vals), vo.data_ptr(), s) == 0
torch.cuda.synchronize()
vo = vo.cpu().numpy().astype(np.uint32)
with np.errstate(over="ignore"):
    hi_ref = vals.astype(np.float16)
    lo_ref32 = (vals - hi_ref.astype(np.float32)).astype(np.float32)
    lo_ref = lo_ref32.astype(np.float16)
hi_bits, lo_bits = vo[:, 0].astype(np.uint16), vo[:, 2].astype(np.uint16)
lo32 = vo[:, 1].view(np.float32) if vo[:, 1].dtype == np.uint32 else None
fin = np.isfinite(hi_ref.astype(np.float32))
m_hi = hi_bits == hi_ref.view(np.uint16)
m_lo32 = (lo32 == lo_ref32) | ~fin
m_lo = (lo_bits == lo_ref.view(np.uint16)) | ~fin
for i in range(19):
    print(f"  {vals[i]:<14.9g} hi {hi_bits[i]:#06x} ({hi_bits[i:i+1].view(np.float16)[0]!s:>10}) numpy {hi_ref[i:i+1].view(np.uint16)[0]:#06x}   a-hi {lo32[i]:<14.6g} numpy {lo_ref32[i]:<14.6g}"
          f"   lo {lo_bits[i]:#06x} numpy {lo_ref[i:i+1].view(np.uint16)[0]:#06x}")
print(f"  all {len(vals)} values: hi matches numpy RNE {m_hi.mean():.4f}, a - hi exact {m_lo32.mean():.4f}, lo matches {m_lo.mean():.4f}")
rec = hi_bits.view(np.float16).astype(np.float64) + lo_bits.view(np.float16).astype(np.float64)
rel = np.abs(rec[fin] - vals[fin].astype(np.float64)) / np.maximum(np.abs(vals[fin].astype(np.float64)), 1e-30)
big = np.abs(vals[fin]) > 1e-3
print(f"  hi + lo vs value: max relative error {rel[big].max():.3e} for |v| > 1e-3 (2^-22 = {2**-22:.3e}); max absolute error for |v| <= 1e-3: "
      f"{np.abs(rec[fin] - vals[fin].astype(np.float64))[~big].max():.3e} (2^-25 = {2**-25:.3e})")


def smi():
    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
    return o[1] if len(o) > 1 else ""


print("== 3. bare MFMA streams, random operand data: f16 next to bf16 (ms per launch, share of the 2.5 PFLOP/s dense peak; rocm-smi while running)")
out1 = torch.zeros(64, device=dev)
blocks = 6144
rnd = torch.randn(4096 * 4 * 2, device=dev)
data = {0: ((rnd.bfloat16().view(torch.int16).to(torch.int32) & 0xffff).view(-1, 2) * torch.tensor([1, 65536], device=dev)).sum(1).to(torch.int32).contiguous(),
        1: ((rnd.half().view(torch.int16).to(torch.int32) & 0xffff).view(-1, 2) * torch.tensor([1, 65536], device=dev)).sum(1).to(torch.int32).contiguous()}
for mode, name, units, flop in ((1, "16x16x32, 2 waves/SIMD (forward shape)", 260, 16 * 16 * 32 * 2), (0, "32x32x16, 1 wave/SIMD (dgrad shape)", 240, 32 * 32 * 16 * 2)):
    for fill in (0, 1, 2):
        for f16 in (0, 1):
            d = data[f16]

            def run():
                assert L.probe_f16_stream(mode, fill, f16, d.data_ptr(), out1.data_ptr(), blocks, units, s) == 0
            for _ in range(3): run()
            torch.cuda.synchronize()
            samples, stop = [], [False]

            def sampler():
                while not stop[0]:
                    samples.append(smi()); time.sleep(0.05)
            th = threading.Thread(target=sampler); th.start()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 2.0:
                for _ in range(20): run()
                torch.cuda.synchronize(); n += 20
            t = (time.perf_counter() - t0) / n
            stop[0] = True; th.join()
            mf = blocks * (4 if mode == 0 else 8) * units * 12
            print(f"  {name:40s} fill {fill} {'f16 ' if f16 else 'bf16'}: {t * 1e3:7.3f} ms  {mf * flop / t / 2.5e15:5.1%} of 2.5 PF   smi: {samples[-1] if samples else ''}", flush=True)
