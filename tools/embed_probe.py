import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd import ops, EmbedCfg
yy, xx = np.meshgrid(np.arange(1024, dtype=np.int32), np.arange(1024, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).cuda()
a4, p4, _ = oracle.synthetic_periodicity(1024, 3)
cfg = EmbedCfg.make(a4, p4, oracle.SEED0_FREQS, (1024, 1024))
for name, dt, prec, bpe in (("fp32_precise", torch.float32, True, 4), ("fp32_fast", torch.float32, False, 4), ("bf16_fast", torch.bfloat16, False, 2)):
    for _ in range(3): ops.embed_fwd(grid, cfg, dt, precise=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.embed_fwd(grid, cfg, dt, precise=prec)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    nb = grid.shape[0] * (8 + bpe * 3 * 462)
    print(name, f"{t*1e3:.3f} ms  {nb/t/1e9:.0f} GB/s  {grid.shape[0]/t/1e6:.0f} Mpx/s")
