"""Diagnostic (not collected by pytest): one TrainingStep of the fp32 path and of the two 16-bit-storage modes on the oracle's synthetic
state (the parity tests' model) and the same batch, at growing image sizes: loss terms and gradient norm side by side, plus the
relative L2 distance of the parameter gradients (FusedClipSGD's flat gradient) from the fp32 step's."""
import copy, importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
from oracle import network as onet
dev = torch.device("cuda", 0)
state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(7), 8))
for (n, h, w) in [(1, 64, 96), (2, 128, 160), (2, 256, 320), (8, 256, 320), (2, 512, 640)]:
    m0 = ea.FCDenseNet57(1)
    m0.load_state_dict(state)
    batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=11).items()}
    g32 = None
    for mode in ("fp32", "bf16", "fp16"):
        m = copy.deepcopy(m0).to(dev).train()
        opt = ea.optim.FusedClipSGD(m, lr=0.0)
        step = ea.train_step.TrainingStep(m, opt, h, w, bf16_storage=(mode == "bf16"), fp16_storage=(mode == "fp16"))
        o = step(batch, lr=0.0)
        torch.cuda.synchronize()
        g = m._flat_grad.double().clone() if getattr(m, "_flat_grad", None) is not None else None
        d = ""
        if g is not None:
            if g32 is None: g32 = g
            else: d = "  grad vs fp32: rel L2 %.3e  cos %.5f" % (float((g - g32).norm() / g32.norm()), float((g * g32).sum() / (g.norm() * g32.norm())))
        print("%d x %d x %d  %-5s loss %.6f  sfl %.6f  dcl %.6f  grad_norm %.4f%s" % (n, h, w, mode, o["loss"], o["sfl"], o["dcl"], float(o["grad_norm"]), d))
