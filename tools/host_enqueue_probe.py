"""How long the HOST needs to enqueue one training step (no synchronisation inside the loop) against the step's wall time: when the
two are equal the step is launch-bound.  usage: host_enqueue_probe.py [config]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
torch.manual_seed(10085)
model = pkg.models.FCDenseNet57(n_classes=1)
pkg.utils.kaiming_weight_zero_bias(model, mode="fan_in", activation_mode="relu", distribution="normal")
model = model.to(dev).train()
opt = pkg.optim.FusedClipSGD(model, lr=1.0e-3)
step = pkg.train_step.TrainingStep(model, opt, 256, 320, bf16_storage=(cfg == 2), fp16_storage=(cfg == 4))
batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(8, 256, 320, seed=0).items()}
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
host = []
for _ in range(n):
    a = time.perf_counter()
    step(batch)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("config %d: host enqueue %.3f ms per step (min %.3f, max %.3f); wall %.3f ms per step; the queue drained %.3f ms after the last enqueue" % (
    cfg, 1e3 * sum(host) / n, 1e3 * min(host), 1e3 * max(host), 1e3 * (t2 - t0) / n, 1e3 * (t2 - t1)))
