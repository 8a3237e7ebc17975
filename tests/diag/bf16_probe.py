"""Development probe: which parameter gradients differ between fp32 and bf16-operand MFMA modes (ENDO_OPT_MFMA_BF16)."""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
n, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 256, 320)))
def run(bf):
    torch.manual_seed(10085)
    model = ea.FCDenseNet57(1)
    model.set_kernel_option(4, bf)
    ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
    model = model.to(dev).train()
    x = torch.randn(n, 3, h, w, device=dev)
    y = model(x)
    (y * torch.randn_like(y)).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(1)
y0, g0 = run(0)
torch.manual_seed(1)
y1, g1 = run(mode)
print("mode", mode)
print("forward rel diff", float((y1 - y0).norm() / y0.norm()))
import numpy as np
rels = []
bad = 0
for k in g0:
    a, b = g0[k], g1[k]
    rel = float((b - a).norm() / (a.norm() + 1e-30))
    nan = bool(torch.isnan(b).any())
    if not k.endswith("conv.bias") and not k.endswith("convTrans.1.bias"):
        rels.append(rel)
    if nan or rel > 2e-2:
        bad += 1
        if bad < 4:
            print("%-55s shape %-18s rel %.3e nan %s" % (k, tuple(a.shape), rel, nan))
print("tensors off:", bad, "of", len(g0))
rels = np.array(rels)
print("weight / BN tensors: median rel %.3e  p90 %.3e  max %.3e" % (np.median(rels), np.quantile(rels, 0.9), rels.max()))
