"""Re-create the reference's initial weights deterministically (test helper).

models/networks.py:40-49 (NPP_Net) / :128-140 (NPP_Net_top1) construct their
nn.Linear modules in this order: periodic_linears[0..7], (NPP_Net: scale_linears[0],)
pos_linears[0], feature_linear1, feature_linear2, alpha_linear, rgb_linear.  With
torch.manual_seed(0) on the CPU generator this reproduces the tensors the golden
generator saw, so full-size weights need not be stored in tests/golden/.
"""
import torch


def reference_init(K, W=256, E=462, D=8, skips=(4,), seed=0):
    torch.manual_seed(seed)
    mods = {}
    for i in range(D):
        cin = E if i == 0 else (W + E if (i - 1) in skips else W)
        mods[f"periodic_linears.{i}"] = torch.nn.Linear(cin, W)
    if K > 1:
        mods["scale_linears.0"] = torch.nn.Linear((K - 1) * E + W, W)
        mods["pos_linears.0"] = torch.nn.Linear(2 * W, W // 2)
    else:
        mods["pos_linears.0"] = torch.nn.Linear(W, W // 2)
    mods["feature_linear1"] = torch.nn.Linear(W, W)
    mods["feature_linear2"] = torch.nn.Linear(W, W)
    mods["alpha_linear"] = torch.nn.Linear(W, 1)
    mods["rgb_linear"] = torch.nn.Linear(W // 2, 3)
    P = {}
    for name, m in mods.items():
        if name == "alpha_linear" or (K == 1 and name == "feature_linear2"):
            continue  # built and seeded by the reference, never used
        P[name + ".weight"] = m.weight.detach().numpy().copy()
        P[name + ".bias"] = m.bias.detach().numpy().copy()
    return P
