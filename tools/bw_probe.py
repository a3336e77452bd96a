import torch, sys
dev="cuda"
def timeit(fn, iters=10):
    for _ in range(2): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters
n = 1<<30  # 4 GiB of floats
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
ms = timeit(lambda: x.fill_(1.0)); print("fill 4GiB  : %.3f ms  %.0f GB/s write" % (ms, 4*n/ms/1e6))
ms = timeit(lambda: y.copy_(x)); print("copy 4GiB  : %.3f ms  %.0f GB/s (r+w)" % (ms, 8*n/ms/1e6))
ms = timeit(lambda: x.sum()); print("sum 4GiB   : %.3f ms  %.0f GB/s read" % (ms, 4*n/ms/1e6))
z = torch.empty(n//4, device=dev)
ms = timeit(lambda: torch.nn.functional.interpolate(z.view(32,128,256,256)[:, :32], scale_factor=2, mode='nearest')); 
print("nearest x2 (1:4 r:w) : %.3f ms %.0f GB/s" % (ms, 32*32*256*256*4*5/ms/1e6))
