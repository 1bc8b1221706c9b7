import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd.fit import CompletionFit
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
angles, periods, _ = oracle.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), N_rand=8192, ksplit=12)
net = fit.net
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
c = fit.i_all_dev[:26624].contiguous()
print(os.environ.get("NPP_LIB_PATH", "in-tree"), "render_512sq us %.1f" % timed(lambda: net.render(fit.i_all_dev)), "fwd_train us %.1f" % timed(lambda: net.forward_train(c)))
