"""Round 6: the value-only contextual core at the ranking's size (1 sample, 256 channels, 128 x 128 positions), HIP events around the call."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from npp_amd import ops  # noqa: E402
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
y = torch.relu(torch.randn(1, 256, 128, 128, device=dev, generator=g))
x = torch.relu(0.7 * y + 0.7 * torch.randn(1, 256, 128, 128, device=dev, generator=g))
for _ in range(3):
    loss, _ = ops.cx_fwd_bwd(x, y, want_grad=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    loss, _ = ops.cx_fwd_bwd(x, y, want_grad=False)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('NPP_LIB_PATH', 'default')}: {e0.elapsed_time(e1) / 10:.3f} ms per value-only call, loss {loss.item():.6f}")
