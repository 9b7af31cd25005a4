import torch, time
dev = torch.device("cuda", 0)
def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
n = 2 * 1024**3  # floats = 8 GiB
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
t = timeit(lambda: x.fill_(1.0)); print("fill 8GiB: %.3f ms  %.2f TB/s" % (t*1e3, n*4/t/1e12))
t = timeit(lambda: y.copy_(x)); print("copy 8GiB: %.3f ms  %.2f TB/s (r+w)" % (t*1e3, 2*n*4/t/1e12))
t = timeit(lambda: x.sum()); print("sum 8GiB: %.3f ms  %.2f TB/s" % (t*1e3, n*4/t/1e12))
