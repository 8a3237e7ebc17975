"""Profiling workload (development tool): the bf16-storage network forward + backward of a training step's two batches
(8 x 256 x 320 each, two sample groups of one call), 7 iterations.  usage: rocprofv3 --kernel-trace --stats -- python3 tools/bf16_step_workload.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")

n, h, w = 8, 256, 320
dev = torch.device("cuda:0")
m = ea.FCDenseNet57(1)
ea.utils.kaiming_weight_zero_bias(m, mode="fan_in", activation_mode="relu", distribution="normal")
m = m.to(dev).train()
x1 = torch.rand((n, 3, h, w), device=dev) * 2 - 1
x2 = torch.rand((n, 3, h, w), device=dev) * 2 - 1
g = torch.randn((n, 1, h, w), device=dev)
x = torch.cat([x1, x2])
gg = torch.cat([g, g])
for it in range(7):          # both frames as two sample groups of one call, as TrainingStep(bf16_storage=True) runs them
    with torch.no_grad():
        y, tape = m._run_forward16(x, 2)
        m._run_backward16(tuple(x.shape), tape, gg, True, 2)
torch.cuda.synchronize()
print("done")
