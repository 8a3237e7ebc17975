"""Where does gradient noise enter?  (development tool; run on a GPU box: python tests/diag/gpu_diag6.py [n h w])

For every convolution output of the network, in backward order: the HIP gradient workspace plane (total gradient of those
maps) against the fp64 oracle, beside the fp32 CPU oracle's own distance from fp64; then the same for every parameter
gradient.  ratio = q90(|hip - fp64|) / q90(|cpu32 - fp64|)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as onet
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 128, 160)
smooth = "--smooth" in sys.argv


def state_as(state, dtype):
    return {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}


def oracle_run(state, x, cot, dtype):
    st = state_as(state, dtype)
    names = onet.trainable_names()
    for nm in names:
        st[nm].requires_grad_(True)
    trace = {}
    y = onet.forward(st, x.to(dtype), training=True, trace=trace)
    keys = [k for k in trace if k.startswith("conv::")]
    grads = torch.autograd.grad((y * cot.to(dtype)).sum(), [st[nm] for nm in names] + [trace[k] for k in keys])
    return dict(zip(names, grads[:len(names)])), dict(zip(keys, grads[len(names):]))


def planes():
    """(conv key, level, first channel, channels) in forward order, following the level-buffer layout of net.hip."""
    out = [("conv::firstconv", 0, 48, 48)]
    for l in range(5):
        cl = 48 + 48 * l
        for j in range(4):
            out.append(("conv::denseBlocksDown.%d.layers.%d" % (l, j), l, 48 + cl + 12 * j, 12))
        out.append(("conv::transDownBlocks.%d" % l, l + 1, 48 if l + 1 < 5 else 0, cl + 48))
    for j in range(4):
        out.append(("conv::bottleneck.bottleneck.layers.%d" % j, 5, 288 + 12 * j, 12))
    for i in range(5):
        l = 4 - i
        cl = 48 + 48 * l
        out.append(("conv::transUpBlocks.%d" % i, l, 0, 48))
        for j in range(4):
            out.append(("conv::denseBlocksUp.%d.layers.%d" % (i, j), l, 96 + cl + 12 * j, 12))
    return out


def q90(d):
    d = d.reshape(-1)
    return float(torch.quantile(d, 0.9)) if d.numel() >= 10 else float(d.max())


state = onet.perturb_affine(onet.synthetic_state(52), 53)
model = ea.FCDenseNet57(1)
model.load_state_dict(state)
model = model.to(dev).train()
rng = np.random.default_rng(6)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
if smooth:
    cot = 0.5 + ea.synthetic.smooth_depth(n, h, w, seed=9)
else:
    cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
p32, c32 = oracle_run(state, x, cot, torch.float32)
p64, c64 = oracle_run(state, x, cot, torch.float64)
y = model(x.to(dev))
(y * cot.to(dev)).sum().backward()
torch.cuda.synchronize()
lib = ea._lib.load()
hnd, _, _ = model._handle(n, h, w, 1)
ws = model._gradws[(n, h, w, 1)]
print("%-46s %10s %10s %10s %7s" % ("total gradient of conv output (backward order)", "scale", "q90 hip", "q90 cpu32", "ratio"))
for key, lvl, c0, cnt in reversed(planes()):
    ch = lib.endo_net_level_channels(lvl)
    off = lib.endo_net_act_offset(hnd, lvl)
    hh, ww = h >> lvl, w >> lvl
    buf = ws[off:off + n * ch * hh * ww].view(n, ch, hh, ww)[:, c0:c0 + cnt].double().cpu()
    r64 = c64[key]
    scale = float(r64.abs().max())
    a, b = q90((buf - r64).abs() / scale), q90((c32[key].double() - r64).abs() / scale)
    print("%-46s %10.3e %10.3e %10.3e %7.1f" % (key[6:], scale, a, b, a / max(b, 1e-30)))
print()
print("%-46s %10s %10s %10s %7s" % ("parameter gradient (backward order)", "scale", "q90 hip", "q90 cpu32", "ratio"))
params = dict(model.named_parameters())
for nm in reversed(onet.trainable_names()):
    if nm.endswith("conv.bias") or nm.endswith("convTrans.1.bias"):
        continue
    r64 = p64[nm]
    scale = float(r64.abs().max())
    a = q90((params[nm].grad.double().cpu() - r64).abs() / scale)
    b = q90((p32[nm].double() - r64).abs() / scale)
    print("%-46s %10.3e %10.3e %10.3e %7.1f" % (nm, scale, a, b, a / max(b, 1e-30)))
