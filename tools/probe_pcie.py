import torch, time, numpy as np
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device='cuda')
for _ in range(2):
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print("H2D pinned GB/s", 5 * n / (time.perf_counter() - t) / 1e9)
t = time.perf_counter()
for _ in range(5):
    h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
print("D2H pinned GB/s", 5 * n / (time.perf_counter() - t) / 1e9)
p = np.ones(n, dtype=np.uint8)
hn = h.numpy()
t = time.perf_counter(); hn[:] = p; print("memcpy 1 thread GB/s", n / (time.perf_counter() - t) / 1e9)
import threading
def part(a, b): hn[a:b] = p[a:b]
for T in (4, 8, 12, 16):
    th = [threading.Thread(target=part, args=(i * n // T, (i + 1) * n // T)) for i in range(T)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]
    print("memcpy", T, "threads GB/s", n / (time.perf_counter() - t) / 1e9)
import os; print("cpus", len(os.sched_getaffinity(0)))
