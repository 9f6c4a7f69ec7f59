"""config-3 shaped data through chq_filter_project_record (single-pass kernel) and the two steps; for rocprofv3 runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_select

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
opts = sys.argv[3:]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
a = torch.randint(0, 1000, (n,), dtype=torch.int32, device=dev, generator=g)
b = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 100, generator=g)
c = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 1100, generator=g)
d = torch.randint(0, 10, (n,), dtype=torch.int32, device=dev, generator=g)
e = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 2, generator=g)
torch.cuda.synchronize()
ctx = chq.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_option("time_kernels", 1)
for kv in opts:
    k, v = kv.split("="); ctx.set_option(k, int(v))
rec = chq.DeviceRecordBatch.from_device_pointers(
    [("a", "i", a.data_ptr()), ("b", "f", b.data_ptr()), ("c", "f", c.data_ptr()), ("d", "i", d.data_ptr()), ("e", "f", e.data_ptr())], n, ctx=ctx)
sel = parse_select("select a, a + b as ab, d * 2 as d2, e / 3.0 as e3 from t where a + b > c and d < 5.0 or e > 1.0")
al = [[] for _ in range(5)]
for it in range(4):
    if mode in ("both", "fused"):
        t0 = time.perf_counter()
        o = chq.filter_project_record(sel.selection, sel.projection, rec, al, ctx=ctx)
        st = ctx.last_stats(); o.release()
        print("one-pass", st["launches"], "launch", st["kernel_ns"] / 1e6, "ms kernel", (time.perf_counter() - t0) * 1e3, "ms wall", flush=True)
    if mode in ("both", "two"):
        t0 = time.perf_counter()
        f = chq.filter_record(rec, al, sel.selection, ctx=ctx)
        k1 = ctx.last_stats()["kernel_ns"] / 1e6
        p = chq.project_record(sel.projection, f, al, ctx=ctx)
        torch.cuda.synchronize()
        print("two-step filter kernel", k1, "ms; filter+project wall", (time.perf_counter() - t0) * 1e3, "ms", flush=True)
        f.release(); p.release()
