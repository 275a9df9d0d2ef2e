import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
GB = x.numel() * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print("fill  (write only) %.2f TB/s" % (GB / ms))
ms = t(lambda: y.copy_(x)); print("copy  (r+w)        %.2f TB/s total" % (2 * GB / ms))
ms = t(lambda: x.sum()); print("sum   (read only)  %.2f TB/s" % (GB / ms))
h = torch.empty(1 << 28, dtype=torch.bfloat16, device="cuda")
ms = t(lambda: h.copy_(x)); print("cvt f32->bf16 (4r+2w) %.2f TB/s total" % (1.5 * GB / ms))
