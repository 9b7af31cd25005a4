import ctypes, os, time, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "liback_probe.so"))
L.probe_ack.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
w = torch.randn(36 * 16384, device=dev); dst = torch.empty(2 * 1024**3, device=dev); out = torch.zeros(64, device=dev)
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=1, reps=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
names = {0: "no stores", 1: "plain", 2: "nt", 3: "sc1", 4: "sc0 sc1", 5: "nt + vmcnt(0)"}
blocks, phases = 6144, 36
for nst in (4, 16):
    for pol in (0, 1, 2, 3, 4, 5):
        t = timeit(lambda: L.probe_ack(pol, nst, w.data_ptr(), dst.data_ptr(), dst.numel() * 4, blocks, phases, out.data_ptr(), s))
        gb = 0 if pol == 0 else blocks * phases * nst * 8192 / 1e9
        print("stores/phase/wave %2d  %-14s: %.3f ms  (%.1f GB written, %.2f TB/s)  %.3f us per phase per CU-slot" % (nst, names[pol], t * 1e3, gb, gb / t / 1e3, t * 1e6 / (blocks / 256 * phases)), flush=True)
