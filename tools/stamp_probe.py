import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, npp_amd
from npp_amd import ops
from npp_amd.model import NPPNet
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = 512
angles, periods, _ = oracle.synthetic_periodicity(H, K)
net = NPPNet(angles, periods, oracle.SEED0_FREQS, (H, H), params=oracle.init_params(K), ksplit=4)
rng = np.random.RandomState(0)
L = npp_amd.lib()
L.npp_debug_read_stamps.restype = ctypes.c_int
names = {0: "start", 50: "emb:enter", 51: "emb:warp+bar", 52: "emb:gen0", 53: "emb:bar", 54: "emb:gen1", 55: "emb:mma0-7", 56: "emb:bar",
         1: "L0 mma done", 2: "L0 epi", 3: "L0 bar", 4: "L1 mma", 5: "L1 epi", 6: "L1 bar", 8: "L2 mma", 9: "L2 epi", 10: "L2 bar",
         12: "L3 mma", 13: "L3 epi", 14: "L3 bar", 16: "L4 mma", 17: "L4 epi", 18: "L4 bar", 20: "L5 emb done", 21: "L5 plain mma",
         22: "L5 epi+bar", 40: "P done", 41: "end"}
for mode, n in (("train", 26624), ("render", 262144)):
    c = torch.from_numpy(np.stack([rng.randint(0, H, n), rng.randint(0, H, n)], 1).astype(np.int32)).cuda()
    for rep in range(3):
        if mode == "train": net.forward_train(c)
        else: net.render(c)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    assert L.npp_debug_read_stamps(buf) == 0
    for w in range(2):
        st = {i: buf[w * 64 + i] for i in range(64) if buf[w * 64 + i]}
        t0 = st[0]
        order = sorted(st, key=lambda i: st[i])
        prev = t0
        out = []
        for i in order:
            out.append(f"{names.get(i, i)}:+{(st[i] - prev)}")
            prev = st[i]
        print(mode, "wg", w, "total", st[41] - t0, "|", "  ".join(out))
