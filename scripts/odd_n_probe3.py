"""Plain J^T J (unit entry) at 128 < n <= 256: the eight-wave ring (n % 16 == 0, m even) against the tile-pair kernel the other
shapes take; and f32 at n % 4 != 0 against n % 4 == 0. m = 400 000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
rng = np.random.default_rng(0)
for m, n in ((400_000, 256), (400_001, 256), (400_000, 250), (400_000, 208), (400_000, 200), (400_000, 129)):
    J = rng.standard_normal((m, n)); y = rng.standard_normal(m)
    t = min(M.jtj(J, y)[3] for _ in range(3))
    print(f"f64 m = {m} n = {n:4d}: plain J^T J {t:.3f} ms ({m * n * (n + 3.0) / t / 1e9:.1f} TFLOP/s, {8 * m * (n + 1) / t / 1e6:.0f} GB/s)", flush=True)
for m, n in ((1_000_000, 128), (1_000_000, 127), (1_000_000, 126), (1_000_000, 64), (1_000_000, 63)):
    J = rng.standard_normal((m, n)).astype(np.float32); y = rng.standard_normal(m).astype(np.float32)
    t = min(M.jtj(J, y, dtype=np.float32)[3] for _ in range(3))
    print(f"f32 m = {m} n = {n:4d}: plain J^T J {t:.3f} ms ({m * n * (n + 3.0) / t / 1e9:.1f} TFLOP/s, {4 * m * (n + 1) / t / 1e6:.0f} GB/s)", flush=True)
