import ctypes, os, time, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libcoal_probe.so"))
L.probe_coal.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
n = 512 * 1024**2; x = torch.empty(n, device=dev)       # 2 GiB
s = torch.cuda.current_stream().cuda_stream
for pattern in (0, 1):
    for _ in range(3):
        L.probe_coal(pattern, x.data_ptr(), n * 4, 2048, s)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        L.probe_coal(pattern, x.data_ptr(), n * 4, 2048, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("pattern %d: %.3f ms  %.2f TB/s" % (pattern, dt * 1e3, n * 4 / dt / 1e12), flush=True)
