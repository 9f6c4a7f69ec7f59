import torch, time
x = torch.empty(3_000_000_000, dtype=torch.float32, device="cuda").uniform_()
for f, name, nbytes in [(lambda: x.sum(), "sum (read only)", x.numel()*4), (lambda: x.max(), "max (read only)", x.numel()*4)]:
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    print(name, f"{ms:.3f} ms", f"{nbytes/ms/1e6:.0f} GB/s")
y = torch.empty_like(x[:1_000_000_000])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
y.copy_(x[:1_000_000_000]); torch.cuda.synchronize()
e0.record()
for _ in range(5): y.copy_(x[:1_000_000_000])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/5
print("copy 4 GB", f"{ms:.3f} ms", f"{8e9/ms/1e6:.0f} GB/s (read+write)")
y.zero_(); torch.cuda.synchronize()
e0.record()
for _ in range(5): y.zero_()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/5
print("fill 4 GB", f"{ms:.3f} ms", f"{4e9/ms/1e6:.0f} GB/s (write only)")
