"""Diagnostic (not collected by pytest): one TrainingStep of the fp32 path and of the two 16-bit-storage modes on the SAME
Kaiming-initialised model (what bench.py trains) and batch, at growing image sizes: loss terms and gradient norm side by side."""
import copy, importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda", 0)
for (n, h, w) in [(2, 64, 96), (2, 128, 160), (8, 256, 320), (2, 512, 640)]:
    torch.manual_seed(5)
    m0 = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(m0, mode="fan_in", activation_mode="relu", distribution="normal")
    batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=11).items()}
    outs = {}
    for mode in ("fp32", "bf16", "fp16"):
        m = copy.deepcopy(m0).to(dev).train()
        step = ea.train_step.TrainingStep(m, ea.optim.FusedClipSGD(m, lr=1.0e-3), h, w, bf16_storage=(mode == "bf16"), fp16_storage=(mode == "fp16"))
        o = step(batch, lr=1.0e-3)
        torch.cuda.synchronize()
        outs[mode] = o
        print("%d x %d x %d  %-5s loss %.6f  sfl %.6f  dcl %.6f  grad_norm %.4f  skipped %s" % (n, h, w, mode, o["loss"], o["sfl"], o["dcl"], float(o["grad_norm"]), o["skipped"]))
