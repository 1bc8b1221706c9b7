"""Round-3 probe: the fused NPP_Net_light chains alone in a tight loop (C candidates x 2048 rows): us per launch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import ops
from npp_amd.light import NPPNetLightBatch, default_light_init
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
C, B, H = int(os.environ.get("R3_C", "9")), int(os.environ.get("R3_B", "2048")), 256
rng = np.random.RandomState(0)
cands = [(np.array([10.0 * i, 90.0 + 5 * i], np.float32), np.array([12.0 + i, 9.0 + 2 * i], np.float32)) for i in range(C)]
nb = NPPNetLightBatch(cands, (rng.randn(10) * 10).astype(np.float32), (H, H), default_light_init(256, 4), device=dev, fused=True)
x_per = torch.randn(C, B, 20, device=dev); x_pos = torch.randn(B, 42, device=dev); gt = torch.rand(B, 3, device=dev)
nb.train_step(x_pos, x_per, gt)
ws = nb._ws[("fused", B)]
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fl_f = 2.0 * C * B * (32 * 256 + 4 * 65536 + 304 * 128 + 384)
print(f"C={C} B={B}: pack {t(lambda: ops.light_pack(nb._desc, nb.params, nb._pack)):.1f} us")
tf = t(lambda: ops.light_fwd(nb._desc, nb.params, nb._pack, x_per, x_pos, ws['stash'], ws['pred']))
tb = t(lambda: ops.light_bwd(nb._desc, nb.params, nb._pack, ws['stash'], ws['pred'], ws['dpred'], ws['draw'], ws['dstash']))
print(f"  fwd {tf:.1f} us = {fl_f / tf * 1e-6:.1f} TF ({fl_f / tf * 1e-6 / 157.3:.2f} of fp32 MFMA peak)   bwd {tb:.1f} us")
print(f"  whole step {t(lambda: nb.train_step(x_pos, x_per, gt)):.1f} us")
