"""Throughput mode: two independent image fits interleaved on ONE GPU, each on its own stream (their dependent-launch gaps and
under-filled kernels overlap).  Compares against the same two fits run one after the other."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import synthetic as syn
from npp_amd.fit import CompletionFit
H, K, NF = 512, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 2
a, p, s = syn.synthetic_periodicity(H, K)
fits, pools, streams = [], [], []
for r in range(NF):
    img, mask = syn.synthetic_image(H, seed=r)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f = CompletionFit(img, mask, a, p, syn.SEED0_FREQS, syn.init_params(K, seed=r), N_rand=8192, shifts=s, ksplit=12, seed=r)
        pool = []
        while len(pool) < 20:
            b = f.sample_batch()
            if b is not None:
                pool.append(b)
        for b in pool:
            f.step_from(b)
    fits.append(f); pools.append(pool); streams.append(st)
torch.cuda.synchronize()
n = 200
rows = fits[0].N_rand + fits[0].patch_num * fits[0].patch_size ** 2
t0 = time.perf_counter()
for r in range(NF):
    with torch.cuda.stream(streams[r]):
        for i in range(n):
            fits[r].step_from(pools[r][i % 20])
    torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(n):
    for r in range(NF):
        with torch.cuda.stream(streams[r]):
            fits[r].step_from(pools[r][i % 20])
torch.cuda.synchronize()
t_int = time.perf_counter() - t0
print(f"{NF} fits: one after the other {NF * n * rows / t_seq / 1e6:.1f} M rows/s ({t_seq / (NF * n) * 1e3:.3f} ms/iter); "
      f"interleaved on {NF} streams {NF * n * rows / t_int / 1e6:.1f} M rows/s ({t_int / (NF * n) * 1e3:.3f} ms/iter)")
