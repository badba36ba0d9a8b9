import torch
x = torch.empty(256 * 1024 * 1024, device="cuda")  # 1 GiB
y = torch.empty_like(x)
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
us = t(lambda: x.fill_(1.0)); print("fill 1 GiB:", us, "us", x.numel() * 4 / us / 1e6, "TB/s write")
us = t(lambda: y.copy_(x)); print("copy 1 GiB:", us, "us", 2 * x.numel() * 4 / us / 1e6, "TB/s r+w")
us = t(lambda: x.sum()); print("sum 1 GiB:", us, "us", x.numel() * 4 / us / 1e6, "TB/s read")
xs = x[:64 * 1024 * 1024]
us = t(lambda: xs.fill_(1.0)); print("fill 256 MiB:", us, "us", xs.numel() * 4 / us / 1e6, "TB/s write")
