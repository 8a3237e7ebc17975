"""Run a few training steps at a given shape (development aid for rocprofv3): profile_shape.py N H W [steps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
torch.manual_seed(10085)
model = ea.FCDenseNet57(1)
ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
model = model.to(dev).train()
opt = ea.optim.FusedClipSGD(model, lr=1e-3)
step = ea.train_step.TrainingStep(model, opt, h, w)
batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=0).items()}
for _ in range(steps):
    step(batch, lr=1e-3)
torch.cuda.synchronize()
print("done")
