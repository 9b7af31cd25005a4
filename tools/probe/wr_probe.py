import ctypes, os, time, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libwr_probe.so"))
L.probe_write.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
n = 2 * 1024**3; x = torch.empty(n, device=dev)       # 8 GiB
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=2, reps=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for mode in (0, 1, 2, 3):
    for nt in (0, 1):
        for blocks in (256, 1024, 4096):
            t = timeit(lambda: L.probe_write(mode, nt, x.data_ptr(), n * 4, blocks, s))
            print("write mode %d nt %d blocks %5d: %.3f ms  %.2f TB/s" % (mode, nt, blocks, t * 1e3, n * 4 / t / 1e12), flush=True)
