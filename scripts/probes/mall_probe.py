"""Does data written by one kernel get served from the Infinity Cache (MALL, 256 MB) when the next kernel reads it?
write(A) then read(A) against write(A) then read(B) with B cold, for several sizes. torch ops only (fill_ / sum)."""
import torch
dev = torch.device("cuda")
def t_ms(fn, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        fn(None)
        e0.record(); fn(1); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]
big = torch.empty(3 * 1024 ** 3 // 8, dtype=torch.float64, device=dev)      # 3 GB scrubber
for mb in (16, 32, 64, 128, 192, 256, 512, 1024):
    nel = mb * 1024 * 1024 // 8
    A = torch.empty(nel, dtype=torch.float64, device=dev); B = torch.ones(nel, dtype=torch.float64, device=dev)
    def hot(phase):
        if phase is None: big.fill_(0.0); A.fill_(1.0)       # scrub the caches, then write A
        else: A.sum()
    def cold(phase):
        if phase is None: B.sum(); big.fill_(0.0); A.fill_(1.0)   # B was touched before the scrub
        else: B.sum()
    th, tc = t_ms(hot), t_ms(cold)
    print(f"{mb:5d} MB: read-after-write {mb / 1024 / th * 1e3:7.2f} GB/ms... {th:.4f} ms ({mb / 1.024 / th / 1e3:.2f} TB/s)   cold read {tc:.4f} ms ({mb / 1.024 / tc / 1e3:.2f} TB/s)")
