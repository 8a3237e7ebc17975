"""Structure of the extra noise in the gradient of denseBlocksUp.4.layers.1's output (development tool).
python tests/diag/gpu_diag7.py [n h w] [--serial]"""
import importlib, os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as onet
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, h, w = (int(a) for a in args[:3]) if len(args) >= 3 else (2, 128, 160)
lib = ea._lib.load()
state = onet.perturb_affine(onet.synthetic_state(52), 53)
rng = np.random.default_rng(6)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
st = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
for nm in onet.trainable_names():
    st[nm].requires_grad_(True)
trace = {}
y = onet.forward(st, x.double(), training=True, trace=trace)
keys = ["conv::denseBlocksUp.4.layers.%d" % j for j in range(4)]
G64 = [t.detach() for t in torch.autograd.grad((y * cot.double()).sum(), [trace[k] for k in keys])]
model = ea.FCDenseNet57(1)
model.load_state_dict(state)
model = model.to(dev).train()
if "--serial" in sys.argv:
    model.set_kernel_option(5, 0)          # ENDO_OPT_WGRAD_OVERLAP
yh = model(x.to(dev))
(yh * cot.to(dev)).sum().backward()
torch.cuda.synchronize()
hnd, _, _ = model._handle(n, h, w, 1)
ws = model._gradws[(n, h, w, 1)]
ch = lib.endo_net_level_channels(0)
off = lib.endo_net_act_offset(hnd, 0)
buf = ws[off:off + n * ch * h * w].view(n, ch, h, w).double().cpu()
for j in range(4):
    c0 = 96 + 48 + 12 * j + 48          # level 0: [0,48) TU, [48,144) skip, [144,192) up-block new maps
    got = buf[:, 96 + 48 + 12 * j:96 + 48 + 12 * j + 12]
    ref = G64[j]
    scale = float(ref.abs().max())
    err = (got - ref) / scale
    print("layer %d: q90 |err| %.3e  max %.3e  mean err (signed) %.3e" % (j, float(torch.quantile(err.abs().reshape(-1)[::5], 0.9)), float(err.abs().max()), float(err.mean())))
    if j == 1:
        print("  per channel  mean signed err:", ["%.1e" % float(err[:, c].mean()) for c in range(12)])
        print("  per channel  rms err       :", ["%.1e" % float(err[:, c].pow(2).mean().sqrt()) for c in range(12)])
        print("  per sample rms:", [float(err[s].pow(2).mean().sqrt()) for s in range(n)])
        rows = err.pow(2).mean((0, 1, 3)).sqrt()
        cols = err.pow(2).mean((0, 1, 2)).sqrt()
        print("  rms by row mod 6 :", ["%.1e" % float(rows[r::6].mean()) for r in range(6)])
        print("  rms first/last rows:", ["%.1e" % float(v) for v in rows[:3]], ["%.1e" % float(v) for v in rows[-3:]])
        print("  rms by col mod 32 (every 4th):", ["%.1e" % float(cols[c::32].mean()) for c in range(0, 32, 4)])
        # is the error affine in x per channel (a P / Q problem)?  least squares err ~ a x + b per channel
        xin = trace[keys[1]].detach()
        for c in (0, 5, 11):
            e = err[:, c].reshape(-1) * scale
            xv = xin[:, c].reshape(-1)
            A = torch.stack([xv, torch.ones_like(xv)], 1)
            sol = torch.linalg.lstsq(A, e.unsqueeze(1)).solution.reshape(-1)
            resid = e - A @ sol
            print("  channel %2d: err ~ %.3e * x + %.3e   explained variance %.3f" % (c, float(sol[0]), float(sol[1]), 1.0 - float(resid.var() / e.var())))
# deferred-term tables of level 0, new-map channels of the up block
off5 = lib.endo_net_act_offset(hnd, 5)
pq_off = off5 + (n * 336 * (h >> 5) * (w >> 5) + 63) // 64 * 64
P = ws[pq_off:pq_off + ch].double().cpu()
Q = ws[pq_off + ch:pq_off + 2 * ch].double().cpu()
print("P[156:168]", P[156:168].numpy())
print("Q[156:168]", Q[156:168].numpy())
