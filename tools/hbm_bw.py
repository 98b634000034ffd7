"""Raw HBM rates of this GPU with torch kernels (fill = write, copy = read + write, sum = read) on 2 GiB."""
import torch, time
x = torch.empty(512*1024*1024, dtype=torch.float32, device="cuda")  # 2 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
GB = x.numel() * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print("fill  %.2f GB in %.3f ms = %.2f TB/s write" % (GB, ms, GB / ms))
ms = t(lambda: y.copy_(x)); print("copy  %.2f GB in %.3f ms = %.2f TB/s r+w" % (GB, ms, 2 * GB / ms))
ms = t(lambda: x.sum()); print("sum   %.2f GB in %.3f ms = %.2f TB/s read" % (GB, ms, GB / ms))
