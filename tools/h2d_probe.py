import torch, time
n = 128 << 20
a = torch.empty(n, dtype=torch.uint8); a.fill_(1)
p = torch.empty(n, dtype=torch.uint8).pin_memory(); p.fill_(2)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, src in (("pageable", a), ("pinned", p)):
    for _ in range(2): d.copy_(src); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): d.copy_(src, non_blocking=True); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(name, round(dt * 1e3, 2), "ms", round(n / dt / 1e9, 1), "GB/s")
import numpy as np, threading
src = a.numpy(); dst = p.numpy()
def cp(lo, hi): np.copyto(dst[lo:hi], src[lo:hi])
for T in (1, 2, 4, 8):
    t = time.perf_counter()
    for _ in range(3):
        th = [threading.Thread(target=cp, args=(i * n // T, (i + 1) * n // T)) for i in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
    dt = (time.perf_counter() - t) / 3
    print("host memcpy", T, "threads", round(n / dt / 1e9, 1), "GB/s")
