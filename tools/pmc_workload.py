"""Small fixed workload for rocprofv3 --pmc passes: 2 full training steps at N=8, 256x320."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
torch.manual_seed(10085)
model = ea.FCDenseNet57(1)
ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
model = model.to(dev).train()
opt = ea.optim.FusedClipSGD(model, lr=1e-3)
step = ea.train_step.TrainingStep(model, opt, 256, 320)
batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(8, 256, 320, seed=0).items()}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    step(batch, lr=1e-3)
torch.cuda.synchronize()
print("done")
