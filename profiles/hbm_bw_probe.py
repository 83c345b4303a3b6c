import torch, time
dev = torch.device("cuda")
n = 720 * 1024 * 1024 // 4
a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it
w = t(lambda: a.fill_(1.0)); c = t(lambda: b.copy_(a)); r = t(lambda: a.sum())
gb = n * 4 / 1e9
print("fill  %.3f ms  %.2f TB/s (write only)" % (w * 1e3, gb / w / 1e3))
print("copy  %.3f ms  %.2f TB/s (read+write)" % (c * 1e3, 2 * gb / c / 1e3))
print("sum   %.3f ms  %.2f TB/s (read only)" % (r * 1e3, gb / r / 1e3))
