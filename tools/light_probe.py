import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.light import NPPNetLight, default_light_init
from npp_amd import ops
H = 512
img, mask = oracle.synthetic_image(H)
angles, periods, _ = oracle.synthetic_periodicity(H, 1)
net = NPPNetLight(angles[0], periods[0], oracle.SEED0_FREQS, (H, H), default_light_init(256), device="cuda")
c = torch.from_numpy(np.stack([np.random.randint(0, H, 2048), np.random.randint(0, H, 2048)], 1).astype(np.int32)).cuda()
gt = torch.rand(2048, 3, device="cuda")
x_pos, x_per = net.embed(c)
for _ in range(10): net.train_step(x_pos, x_per, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): net.train_step(x_pos, x_per, gt)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("train_step: host enqueue %.3f ms, total %.3f ms per iteration" % ((t1 - t0) * 10, (t2 - t0) * 10))
