"""Diagnostic: the stages of TrainingStep (forward predictions, d loss / d prediction, flat gradient) in the fp32 and the bf16-storage
family on the same batch and parameters."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
from oracle import network as onet
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 64, 96)
dev = torch.device("cuda:0")
HALF = int(os.environ.get("HALF", "0"))
batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(n, h, w, seed=int(os.environ.get("SEED", "0")), sparse_points=int(os.environ.get("POINTS", "500"))).items()}
out = {}
for mode in ("fp32", "bf16"):
    m = pkg.models.FCDenseNet57(1)
    m.load_state_dict(onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(int(os.environ.get("STATE", "71"))), int(os.environ.get("STATE", "71")) + 1)))
    m = m.to(dev).train()
    opt = pkg.optim.FusedClipSGD(m, lr=0.0, momentum=0.9, max_norm=1.0e9)
    step = pkg.train_step.TrainingStep(m, opt, h, w, bf16_storage=(mode == "bf16" and HALF == 0), fp16_storage=(mode == "bf16" and HALF == 1))
    opt.zero_grad()
    losses_t, x, tape, pred, grad_pred = step._fused_iteration(batch)
    step._fused_backward(x, tape, grad_pred)
    torch.cuda.synchronize()
    out[mode] = (losses_t.clone(), pred.clone(), grad_pred.clone(), m.flat_gradients().detach().clone())
    # the same grad_pred through the OTHER family's autograd entry, for the network backward alone
a, b = out["fp32"], out["bf16"]
rel = lambda u, v: float((u.double() - v.double()).norm() / u.double().norm())
print("losses fp32", a[0].tolist(), "bf16", b[0].tolist())
print("pred      rel L2 diff %.3e   max |pred| %.3e / %.3e   min %.3e / %.3e" % (rel(a[1], b[1]), float(a[1].abs().max()), float(b[1].abs().max()), float(a[1].min()), float(b[1].min())))
print("grad_pred rel L2 diff %.3e   norms %.4e / %.4e   max %.3e / %.3e" % (rel(a[2], b[2]), float(a[2].norm()), float(b[2].norm()), float(a[2].abs().max()), float(b[2].abs().max())))
print("grads     rel L2 diff %.3e   norms %.4e / %.4e" % (rel(a[3], b[3]), float(a[3].norm()), float(b[3].norm())))
# network backward alone: feed the fp32 family's grad_pred to the bf16 network
m = pkg.models.FCDenseNet57(1)
m.load_state_dict(onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(int(os.environ.get("STATE", "71"))), int(os.environ.get("STATE", "71")) + 1)))
m = m.to(dev).train()
with torch.no_grad():
    m._attach_grads() if hasattr(m, "_attach_grads") else None
    m.flat_gradients().zero_()
    y, tape = m._run_forward16(x, 2, bool(HALF))
    m._run_backward16(tuple(x.shape), tape, a[2], True, 2, bool(HALF))
torch.cuda.synchronize()
g = m.flat_gradients().detach()
print("bf16 network backward on the fp32 family's grad_pred: rel L2 diff to fp32 grads %.3e   norm %.4e" % (rel(a[3], g), float(g.norm())))
