import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from npp_amd import ops, EmbedCfg, synthetic as syn
dev = torch.device("cuda:0")
yy, xx = np.meshgrid(np.arange(1024, dtype=np.int32), np.arange(1024, dtype=np.int32), indexing="ij")
grid = torch.from_numpy(np.stack([yy, xx], -1).reshape(-1, 2)).to(dev)
a4, p4, _ = syn.synthetic_periodicity(1024, 3)
cfg = EmbedCfg.make(a4, p4, syn.SEED0_FREQS, (1024, 1024))
for name, dt, prec, bpe in (("fp32_precise", torch.float32, True, 4), ("fp32", torch.float32, False, 4), ("bf16", torch.bfloat16, False, 2)):
    for _ in range(3): o = ops.embed_fwd(grid, cfg, dt, precise=prec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): o = ops.embed_fwd(grid, cfg, dt, precise=prec)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(name, f"{t*1e3:.3f} ms", f"{grid.shape[0] * (8 + bpe * 3 * 462) / t / 1e12:.2f} TB/s", "checksum", float(o.double().sum()))
