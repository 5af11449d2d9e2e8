"""Dev probe (GPU): streaming rate (read + write in place) of individual 1.15 GB allocations."""
import torch
dev = "cuda:0"
def rate(t, reps=5):
    t.mul_(1.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): t.mul_(1.0)
    e1.record(); torch.cuda.synchronize()
    return 2 * t.numel() * 4 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12
ts = []
for k in range(16):
    t = torch.ones((6_000_000, 48), dtype=torch.float32, device=dev)
    ts.append(t)
    print(k, hex(t.data_ptr() >> 21), "TB/s", round(rate(t), 3), flush=True)
print("again:", [round(rate(t), 3) for t in ts])
small = [torch.ones((6_000_000, 12), dtype=torch.float32, device=dev) for _ in range(8)]
print("288 MB allocations:", [round(rate(t), 3) for t in small])
