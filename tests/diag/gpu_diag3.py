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
n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 64, 64)
state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(52), 53))
model = ea.FCDenseNet57(1); model.load_state_dict(state); model = model.to(dev).train()
rng = np.random.default_rng(6)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
cot = 0.5 + ea.synthetic.smooth_depth(n, h, w, seed=9)
g32 = ref_grads(state, x, cot, torch.float32); g64 = ref_grads(state, x, cot, torch.float64)
y = model(x.to(dev)); (y * cot.to(dev)).sum().backward()
params = dict(model.named_parameters())
names = onet.trainable_names()
# backward order = reverse parameter order (roughly)
for nm in reversed(names):
    r = g64[nm]; s = float(r.abs().max())
    if nm.endswith(".bias"): s = max(s, float(g64[nm[:-5] + ".weight"].abs().max()))
    s = max(s, 1e-30)
    eh = float((params[nm].grad.detach().cpu().double() - r).abs().max()) / s
    ec = float((g32[nm].double() - r).abs().max()) / s
    print("%-52s max|g| %.3e  hip %.2e cpu %.2e ratio %7.1f" % (nm, float(r.abs().max()), eh, ec, eh / max(ec, 1e-12)))
for nm in ["denseBlocksUp.3.layers.1.norm.bias", "denseBlocksUp.3.layers.1.norm.weight", "denseBlocksUp.2.layers.3.norm.bias"]:
    r = g64[nm].reshape(-1); hg = params[nm].grad.detach().cpu().double().reshape(-1)
    e = (hg - r).abs() / float(r.abs().max())
    bad = [(int(i), float(e[i])) for i in range(e.numel()) if float(e[i]) > 1e-5]
    print(nm, "channels with rel err > 1e-5:", len(bad), bad[:40])
