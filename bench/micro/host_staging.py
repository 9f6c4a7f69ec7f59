"""Where the time of a host-batch call goes: H2D staging, the kernels, D2H of the result -- against the PCIe rate that
pinned memory reaches on the same box (torch pinned copies).  120 MB in (10 M rows x 3 x 4 B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, torch
import chapterhouseqe_amd as chq
from chapterhouseqe_amd.sqlparse import parse_expr


def best(f, reps=8):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


ctx = chq.Context(0)
e = parse_expr("value2 > 10.0")
for n in [1_000_000, 10_000_000, 50_000_000]:
    rng = np.random.default_rng(1)
    rb = pa.RecordBatch.from_arrays([pa.array(np.arange(n, dtype=np.int32)), pa.array((rng.random(n) * 100).astype(np.float32)),
                                     pa.array((rng.random(n) * 100).astype(np.float32))], names=["id", "value1", "value2"])
    al = [[], [], []]
    nbytes = 12 * n
    t_h2d = best(lambda: chq.DeviceRecordBatch.from_host(rb, ctx).release())
    dev = chq.DeviceRecordBatch.from_host(rb, ctx)
    t_d2h = best(lambda: dev.to_host())
    t_dev = best(lambda: chq.filter_record(dev, al, e, ctx=ctx).release())
    t_all = best(lambda: chq.filter_record(rb, al, e, ctx=ctx))
    # PCIe reference points: torch, pinned vs pageable
    pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pag = torch.empty(nbytes, dtype=torch.uint8); pag.fill_(1)
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    def cp(dst, src):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t_pin_h2d = best(lambda: cp(d, pin)); t_pin_d2h = best(lambda: cp(pin, d))
    t_pag_h2d = best(lambda: cp(d, pag)); t_pag_d2h = best(lambda: cp(pag, d))
    t_memcpy = best(lambda: pag.copy_(pin))
    t_fresh = best(lambda: np.empty(nbytes, dtype=np.uint8).fill(0), reps=4)
    g = lambda t: nbytes / t / 1e9
    print(f"n={n}: {nbytes / 1e6:.0f} MB | chq H2D {t_h2d * 1e3:.2f} ms ({g(t_h2d):.1f} GB/s) | chq D2H {t_d2h * 1e3:.2f} ms ({g(t_d2h):.1f} GB/s) | "
          f"device-resident filter {t_dev * 1e3:.3f} ms | host filter_record {t_all * 1e3:.2f} ms ({n / t_all:.3e} rows/s)", flush=True)
    print(f"      torch pinned H2D {g(t_pin_h2d):.1f} GB/s, D2H {g(t_pin_d2h):.1f} GB/s | pageable H2D {g(t_pag_h2d):.1f}, D2H {g(t_pag_d2h):.1f} | "
          f"host memcpy {g(t_memcpy):.1f} GB/s | touch fresh pages {g(t_fresh):.1f} GB/s", flush=True)
    del dev, pin, pag, d
