"""Diagnostic: the bf16-storage training step with the weight gradients on the side stream and in line, A/B in one process."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
n, h, w = 8, 256, 320
dev = torch.device("cuda:0")
torch.manual_seed(10085)
m = pkg.models.FCDenseNet57(1)
pkg.utils.kaiming_weight_zero_bias(m, mode="fan_in", activation_mode="relu", distribution="normal")
m = m.to(dev).train()
opt = pkg.optim.FusedClipSGD(m, lr=1.0e-3, momentum=0.9, max_norm=10.0)
step = pkg.train_step.TrainingStep(m, opt, h, w, bf16_storage=True)
batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(n, h, w, seed=0).items()}
lib = pkg._lib.load()
for _ in range(3):
    step(batch)
hnd = m._handle16(n, h, w, 2)[0]
for rnd in range(3):
    for on in (1, 0):
        lib.endo_net16_set_wgrad_overlap(hnd, on)
        for _ in range(2):
            step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step(batch)
        torch.cuda.synchronize()
        print("round %d  side stream %d: %.3f ms per step" % (rnd, on, (time.perf_counter() - t0) / 20 * 1e3))
