"""Host-only: time of one np.random.choice(n, size, replace=False)-equivalent shuffle in libnpp_hip.so (set NPP_RNG_AVX2 / NPP_RNG_THREADS
= 0 to time the other forms; they are chosen once per process)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd.host_rng import NativeRandomState
g = NativeRandomState(0)
for n in (1 << 20, 245760):
    for _ in range(3):
        g.choice(n, size=[8192], replace=False)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        g.choice(n, size=[8192], replace=False)
        ts.append(time.perf_counter() - t0)
    print(f"AVX2={os.environ.get('NPP_RNG_AVX2', '1')} THREADS={os.environ.get('NPP_RNG_THREADS', '1')} n={n}: median {np.median(ts) * 1e3:.3f} ms  min {min(ts) * 1e3:.3f}  max {max(ts) * 1e3:.3f}")
