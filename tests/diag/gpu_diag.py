"""GPU diagnostic (not a test): per-parameter gradient error table vs the fp64 oracle, and the
train-step fixture components.  Writes gpurun_out/diag.txt."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as onet, train_step as ostep
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
out = open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "w")

def state_as(state, dtype):
    return {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}

def ref_grads(state, x, cot, dtype):
    st = state_as(state, dtype)
    names = onet.trainable_names()
    for nm in names: st[nm].requires_grad_(True)
    y = onet.forward(st, x.to(dtype), training=True)
    g = torch.autograd.grad((y * cot.to(dtype)).sum(), [st[nm] for nm in names])
    return y.detach(), dict(zip(names, g))

for (n, h, w) in [(2, 64, 96), (2, 128, 160)]:
    state = onet.perturb_affine(onet.synthetic_state(52), 53)
    model = ea.FCDenseNet57(1); model.load_state_dict(state); model = model.to(dev).train()
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    y32, g32 = ref_grads(state, x, cot, torch.float32)
    y64, g64 = ref_grads(state, x, cot, torch.float64)
    y = model(x.to(dev)); (y * cot.to(dev)).sum().backward()
    params = dict(model.named_parameters())
    print("shape", n, h, w, "out err hip %.2e cpu %.2e" % (float((y.detach().cpu().double()-y64).abs().max()/y64.abs().max()),
          float((y32.double()-y64).abs().max()/y64.abs().max())), file=out)
    for nm in onet.trainable_names():
        r = g64[nm]; s = float(r.abs().max())
        if nm.endswith(".bias"): s = max(s, float(g64[nm[:-5] + ".weight"].abs().max()))
        s = max(s, 1e-30)
        eh = float((params[nm].grad.detach().cpu().double() - r).abs().max()) / s
        ec = float((g32[nm].double() - r).abs().max()) / s
        # also L2-relative error
        l2h = float((params[nm].grad.detach().cpu().double() - r).norm() / max(float(r.norm()), 1e-30))
        l2c = float((g32[nm].double() - r).norm() / max(float(r.norm()), 1e-30))
        print("%-52s max|g| %.3e  hip %.2e cpu %.2e ratio %6.1f   l2 hip %.2e cpu %.2e" % (nm, float(r.abs().max()), eh, ec, eh / max(ec, 1e-12), l2h, l2c), file=out)

# train-step fixture, iteration 0 components
g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_2x64x96.npz"))
n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
state = onet.perturb_affine(onet.synthetic_state(seed), seed + 1)
model = ea.FCDenseNet57(1); model.load_state_dict(state); model = model.to(dev).train()
opt = ea.optim.FusedClipSGD(model, lr=1e-3)
step = ea.train_step.TrainingStep(model, opt, h, w)
batch = ea.synthetic.make_batch(n, h, w, seed=seed + 10, sparse_points=min(500, h * w // 6))
bd = {k: v.to(dev) for k, v in batch.items()}
loss, dcl, sfl, ex = step.losses(bd)
print("fixture loss %.7f dcl %.7f sfl %.7f" % (float(g["step0_loss"]), float(g["step0_dcl"]), float(g["step0_sfl"])), file=out)
print("hip     loss %.7f dcl %.7f sfl %.7f" % (float(loss), float(dcl), float(sfl)), file=out)
p1 = torch.from_numpy(g["step0_pred_1"])
print("pred_1 rel err %.3e" % float((ex["pred_1"].detach().cpu() - p1).abs().max() / p1.abs().max()), file=out)
st2 = onet.perturb_affine(onet.synthetic_state(seed), seed + 1)
ref = ostep.forward_backward(st2, batch)
for k in ("scaled_1", "scaled_2", "warped_21", "warped_12", "inter_1", "inter_2"):
    a, b = ex[k].detach().cpu(), ref["extras"][k]
    print(k, "max abs diff %.3e  mismatches %d" % (float((a - b).abs().max()), int((a != b).sum())), file=out)
print("oracle loss %.7f dcl %.7f sfl %.7f" % (float(ref["loss"]), float(ref["dcl"]), float(ref["sfl"])), file=out)
out.close()
print(open(os.path.join(ROOT, "gpurun_out", "diag.txt")).read()[-3000:])
