import torch, time
dev = torch.device('cuda')
x = torch.empty(2 * 1024 ** 3, device=dev, dtype=torch.float32).fill_(1.0)   # 8 GiB
y = torch.empty_like(x[:1024 ** 3])
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.sum())
print('read  8 GiB: %.3f ms  %.2f TB/s' % (ms, x.numel() * 4 / ms / 1e9))
ms = t(lambda: y.copy_(x[:1024 ** 3]))
print('copy  4+4 GiB: %.3f ms  %.2f TB/s (r+w)' % (ms, 2 * y.numel() * 4 / ms / 1e9))
ms = t(lambda: y.fill_(2.0))
print('write 4 GiB: %.3f ms  %.2f TB/s' % (ms, y.numel() * 4 / ms / 1e9))
