"""Achievable HBM bandwidth on this GPU for plain streaming kernels (torch fill / sum / copy), the practical roof the
memory-bound kernels of this repo are compared with."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.float32)
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tw = t(lambda: x.fill_(1.0)); tr = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
print("write %.2f TB/s | read %.2f TB/s | copy %.2f TB/s (read+write bytes)" % (n / tw / 1e12, n / tr / 1e12, 2 * n / tc / 1e12))
for mb in (64, 256):
    m = mb << 20
    xs, ys = x[: m // 4], y[: m // 4]
    print("%d MB: write %.2f read %.2f copy %.2f TB/s" % (mb, m / t(lambda: xs.fill_(1.0), 50) / 1e12, m / t(lambda: xs.sum(), 50) / 1e12, 2 * m / t(lambda: ys.copy_(xs), 50) / 1e12))
