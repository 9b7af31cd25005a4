"""fp8 correction terms, hardware questions (tools/probe/f8_probe.hip): (1) which K index does byte j of lane l's A / B operand
of v_mfma_scale_f32_16x16x128_f8f6f4 hold, (2) what do the E8M0 scale operands multiply by, (3) what does
v_cvt_scalef32_pk_fp8_f32 do with its scale, with values beyond 448 and below 2^-9."""
import ctypes, os
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libf8_probe.so"))
vp = ctypes.c_void_p
L.probe_mfma8.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
L.probe_cvt8.argtypes = [vp, ctypes.c_int, vp, ctypes.c_float, vp]
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
A = (torch.randn(16, 128, generator=g) * 2).to(torch.float8_e4m3fn)      # [row][k]
B = (torch.randn(128, 16, generator=g) * 2).to(torch.float8_e4m3fn)      # [k][col]
ref = A.float() @ B.float()


def frag(M_rows_k, hyp):
    """M_rows_k [16][128] fp8 (row or column index first, k second) -> per-lane 32 bytes under a layout hypothesis"""
    byt = M_rows_k.view(torch.uint8)
    out = torch.zeros(64, 32, dtype=torch.uint8)
    for l in range(64):
        r, grp = l % 16, l // 16
        for j in range(32):
            k = {"k = 32 g + j": 32 * grp + j, "k = 4 j + g (interleaved)": None, "k = 16 g + j | 64 + 16 g + (j - 16)": (16 * grp + j) if j < 16 else (64 + 16 * grp + j - 16)}[hyp]
            if k is None:
                k = 16 * (j // 4) + 4 * grp + j % 4
            out[l, j] = byt[r, k]
    return out.view(torch.int32).contiguous()


print("== 1. operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3), D[4 (l / 16) + r][l % 16] = c[r]")
ok_hyp = None
for hyp in ("k = 32 g + j", "k = 16 g + j | 64 + 16 g + (j - 16)", "k = 4 j + g (interleaved)"):
    a = frag(A, hyp).to(dev)
    b = frag(B.t().contiguous(), hyp).to(dev)
    d = torch.zeros(64, 4, device=dev)
    assert L.probe_mfma8(a.data_ptr(), b.data_ptr(), d.data_ptr(), 0x7f7f7f7f, 0x7f7f7f7f, s) == 0
    torch.cuda.synchronize()
    got = torch.zeros(16, 16)
    dc = d.cpu()
    for l in range(64):
        for r in range(4):
            got[4 * (l // 16) + r, l % 16] = dc[l, r]
    err = float((got - ref).abs().max())
    print(f"  hypothesis `{hyp}`: max |D - A B| = {err:.3e} (|AB| max {float(ref.abs().max()):.1f})")
    if err < 1e-3 * float(ref.abs().max()) and ok_hyp is None:
        ok_hyp = hyp
print("  =>", ok_hyp)
print("== 2. scale operands (byte 0 of each lane's dword, op_sel 0): result / unscaled result")
a = frag(A, ok_hyp).to(dev); b = frag(B.t().contiguous(), ok_hyp).to(dev)
for sa, sb in ((127, 127), (128, 127), (127, 125), (121, 113), (127 - 6, 127 - 14)):
    d = torch.zeros(64, 4, device=dev)
    assert L.probe_mfma8(a.data_ptr(), b.data_ptr(), d.data_ptr(), sa * 0x01010101, sb * 0x01010101, s) == 0
    torch.cuda.synchronize()
    got = d.cpu()[0, 0] / ref[0, 0]
    print(f"  scale_a byte {sa}, scale_b byte {sb}: factor {float(got):.6g} = 2^{np.log2(abs(float(got))):.3f} (expected 2^{sa + sb - 254})")
print("== 3. v_cvt_scalef32_pk_fp8_f32(old, a, b, scale, hi): bytes vs torch.float8_e4m3fn(a / scale), saturation")
vals = torch.tensor([1.0, -1.5, 0.3, 448.0, 500.0, 1e4, 2.0 ** -6, 2.0 ** -9, 2.0 ** -10, 3.0 * 2.0 ** -10, 0.0, -0.0, 17.0, 19.0, float("inf"), float("nan")])
for scale in (1.0, 0.25, 2.0 ** -14, 3.0):
    out = torch.zeros(vals.numel(), dtype=torch.int32, device=dev)
    assert L.probe_cvt8(vals.to(dev).data_ptr(), vals.numel(), out.data_ptr(), ctypes.c_float(scale), s) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.uint32)
    got = []
    for i in range(vals.numel() // 2):
        lo, hi = int(o[2 * i]), int(o[2 * i + 1])
        got += [lo & 0xff, (lo >> 8) & 0xff]
        assert (lo >> 16) == 0x2222 and (hi & 0xffff) == 0x1111 and (hi >> 16) == (lo & 0xffff), (hex(lo), hex(hi))
    dec = torch.tensor(got, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    want = (vals / scale).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    print(f"  scale {scale:g}: in {vals.tolist()}")
    print(f"           got  {dec.tolist()}")
    print(f"           want {want.tolist()}   (torch RNE of clamp(a / scale, +-448))")
