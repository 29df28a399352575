"""What this box's HBM streams at, for the roofline discussion: read-only (torch.sum), write-only (fill_),
copy (read + write) over a 16 GB fp32 array — the size of one layer's message buffer."""
import time, torch
n = 4_000_000_000
x = torch.empty(n, dtype=torch.float32, device="cuda")
x.fill_(1.0)
y = torch.empty(n // 2, dtype=torch.float32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
r = t(lambda: x.sum()); print(f"read  {n*4/r/1e12:.2f} TB/s ({r*1e3:.2f} ms)")
w = t(lambda: x.fill_(2.0)); print(f"write {n*4/w/1e12:.2f} TB/s ({w*1e3:.2f} ms)")
c = t(lambda: y.copy_(x[: n // 2])); print(f"copy  {n*4/c/1e12:.2f} TB/s r+w ({c*1e3:.2f} ms)")
