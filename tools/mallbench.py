import torch, sys
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in [8, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048]:
    x = torch.ones(mb * 1024 * 1024 // 8, dtype=torch.float64, device=dev)
    t = timeit(lambda: x.sum())
    print("sum over %5d MB: %8.1f us  %7.0f GB/s" % (mb, t * 1e3, mb * 1.048576 / t))
    del x
