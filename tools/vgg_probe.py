import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd.losses import _Trunk, _VGG19, _VGG16
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for dt in (torch.float32, torch.bfloat16):
        for cl in (False, True):
            for name, cfg, taps, nimg in (("vgg19", _VGG19, (17,), 6), ("vgg16", _VGG16, (3, 8, 15, 22, 29), 2)):
                m = _Trunk(cfg, taps).cuda().to(dt)
                x = torch.rand(nimg, 3, 96, 96, device="cuda", dtype=dt)
                if cl:
                    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
                xr = x.clone().requires_grad_(True)
                def fb():
                    outs = m(xr); sum(o.float().sum() for o in outs).backward()
                try:
                    print(f"bench={bench} {dt} cl={cl} {name}: fwd {t(lambda: m(x)):.3f} ms  fwd+bwd {t(fb):.3f} ms", flush=True)
                except Exception as e:
                    print("fail", bench, dt, cl, name, str(e)[:100])
