import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as onet
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
n, h, w = 2, 64, 64
state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(52), 53))
model = ea.FCDenseNet57(1); model.load_state_dict(state); model = model.to(dev).train()
rng = np.random.default_rng(6)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
cot = 0.5 + ea.synthetic.smooth_depth(n, h, w, seed=9)
st = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
for nm in onet.trainable_names(): st[nm].requires_grad_(True)
trace = {}
y64 = onet.forward(st, x.double(), training=True, trace=trace)
keys = ["tu_%d" % l for l in range(5)] + ["skip_%d" % l for l in range(5)] + ["upnew_%d" % l for l in range(5)] + ["bott_in", "bott_new"]
for k in keys: trace[k].retain_grad()
# upnew_L are slices (views) -> need grads of the producing tensors; use autograd.grad on the stored tensors where possible
(y64 * cot.double()).sum().backward()
y = model(x.to(dev)); (y * cot.to(dev)).sum().backward()
lib = ea._lib.load()
hnd, _, _ = model._handle(n, h, w)
gws = model._gradws[(n, h, w)]
for lvl in range(6):
    ch = lib.endo_net_level_channels(lvl); off = lib.endo_net_act_offset(hnd, lvl)
    hh, ww = h >> lvl, w >> lvl
    g = gws[off:off + n * ch * hh * ww].view(n, ch, hh, ww).cpu().double()
    if lvl < 5:
        cl = 48 + 48 * lvl
        parts = [("tu", 0, 48, trace["tu_%d" % lvl].grad), ("skip", 48, 48 + cl + 48, trace["skip_%d" % lvl].grad)]
    else:
        parts = [("bott_in", 0, 288, trace["bott_in"].grad)]
    for name, c0, c1, ref in parts:
        if ref is None:
            print("level", lvl, name, "no ref grad"); continue
        got = g[:, c0:c1]
        err = (got - ref).abs()
        print("level %d %-8s ch[%d,%d) max|ref| %.3e  max err %.3e (rel %.2e)  per-12ch-block rel err: %s" % (
            lvl, name, c0, c1, float(ref.abs().max()), float(err.max()), float(err.max() / ref.abs().max()),
            ["%.1e" % float(err[:, i:i + 12].max() / ref.abs().max()) for i in range(0, c1 - c0, 12)]))
        if name == "tu" and lvl in (1, 2):
            e2 = err.amax(dim=(0, 1))
            print("   spatial pattern of error (rows x cols maxima):")
            print("   rows:", ["%.0e" % float(v) for v in e2.amax(dim=1)])
            print("   cols:", ["%.0e" % float(v) for v in e2.amax(dim=0)])
