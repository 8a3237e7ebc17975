"""Diagnostic (not a test): one training step of the fp32 family and of the bf16-storage family on the same synthetic batch at the
benchmark size: loss, gradient norm, cosine between the two flat gradients.  usage: python tests/diag/bf16_step_compare.py [n h w]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 256, 320)
dev = torch.device("cuda:0")
batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(n, h, w, seed=0).items()}
res = {}
for mode in ("fp32", "bf16"):
    # the tests' parameters (depth kept away from zero: with Kaiming weights and a zero final bias |pre| has zeros everywhere and the
    # scale-normalised losses turn a 1e-2 difference of the predictions into an O(1) difference of the loss, for ANY two implementations)
    from oracle import network as onet
    m = pkg.models.FCDenseNet57(1)
    m.load_state_dict(onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72)))
    m = m.to(dev).train()
    opt = pkg.optim.FusedClipSGD(m, lr=0.0, momentum=0.9, max_norm=1.0e9)
    step = pkg.train_step.TrainingStep(m, opt, h, w, bf16_storage=(mode == "bf16"))
    out = step(batch)
    torch.cuda.synchronize()
    res[mode] = (out["loss"], float(out["grad_norm"]), m.flat_gradients().detach().clone())
    print("%s: loss %.6f  grad norm %.4f" % (mode, res[mode][0], res[mode][1]))
a, b = res["fp32"][2].double(), res["bf16"][2].double()
print("cosine(fp32 grad, bf16 grad) = %.5f   |b - a| / |a| = %.3e" % (float(a @ b / (a.norm() * b.norm())), float((b - a).norm() / a.norm())))
