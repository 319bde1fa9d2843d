import torch, time
x = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
y = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for name, fn, nbytes in (("fill (write 1 GiB)", lambda: x.fill_(3), 1<<30), ("copy (read+write 1 GiB each)", lambda: y.copy_(x), 2<<30),
                         ("fill 32 MiB", lambda: x[:32<<20].fill_(1), 32<<20)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    print(f"{name}: {best*1e3:.1f} us -> {nbytes/best/1e9:.2f} TB/s")
