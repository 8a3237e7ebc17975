import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as onet
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
def state_as(state, dtype):
    return {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
def ref_grads(state, x, cot, dtype):
    st = state_as(state, dtype); names = onet.trainable_names()
    for nm in names: st[nm].requires_grad_(True)
    y = onet.forward(st, x.to(dtype), training=True)
    g = torch.autograd.grad((y * cot.to(dtype)).sum(), [st[nm] for nm in names])
    return dict(zip(names, g))
n, h, w = 2, 128, 160
state = onet.perturb_affine(onet.synthetic_state(52), 53)
model = ea.FCDenseNet57(1); model.load_state_dict(state); model = model.to(dev).train()
rng = np.random.default_rng(6)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
g32 = ref_grads(state, x, cot, torch.float32); g64 = ref_grads(state, x, cot, torch.float64)
y = model(x.to(dev)); (y * cot.to(dev)).sum().backward()
params = dict(model.named_parameters())
for nm in ["denseBlocksDown.2.layers.1.norm.bias", "denseBlocksDown.2.layers.1.norm.weight", "transDownBlocks.2.conv.bias", "denseBlocksDown.2.layers.2.norm.bias", "denseBlocksDown.2.layers.0.norm.bias"]:
    r = g64[nm].reshape(-1); hgrad = params[nm].grad.detach().cpu().double().reshape(-1); c = g32[nm].double().reshape(-1)
    eh = (hgrad - r).abs(); ec = (c - r).abs()
    top = torch.argsort(eh, descending=True)[:6]
    print(nm, "max|ref| %.3e" % float(r.abs().max()))
    for i in top:
        print("   idx %4d ref % .5e hip % .5e cpu32 % .5e  errh %.2e errc %.2e" % (int(i), float(r[i]), float(hgrad[i]), float(c[i]), float(eh[i]), float(ec[i])))
