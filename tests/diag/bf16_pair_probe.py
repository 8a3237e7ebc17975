"""Development probe: bf16-operand mode against fp32 on the benchmark's own first iteration (grouped pair forward, 2 x 8 x 256x320):
predictions, loss terms and d loss / d prediction."""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(8, 256, 320, seed=0).items()}
res = {}
for mode in (0, 1):
    torch.manual_seed(10085)
    model = ea.FCDenseNet57(1)
    model.set_kernel_option(4, mode)
    ea.utils.kaiming_weight_zero_bias(model, mode="fan_in", activation_mode="relu", distribution="normal")
    model = model.to(dev).train()
    step = ea.train_step.TrainingStep(model, ea.optim.FusedClipSGD(model, lr=1e-3), 256, 320, sfl_weight=20.0, dcl_weight=0.1)
    losses_t, x, tape, pred, grad_pred = step._fused_iteration(batch)
    torch.cuda.synchronize()
    res[mode] = (losses_t.tolist(), pred.clone(), grad_pred.clone())
p0, p1 = res[0][1], res[1][1]
print("losses fp32", res[0][0], "bf16", res[1][0])
print("pred rel L2 diff %.3e  max|diff|/max %.3e  pred min %.3e mean %.3e max %.3e" % (
    float((p1 - p0).norm() / p0.norm()), float((p1 - p0).abs().max() / p0.abs().max()), float(p0.min()), float(p0.mean()), float(p0.max())))
print("fraction of pixels with pred < 1e-2 * mean: %.4f" % float((p0 < 1e-2 * p0.mean()).float().mean()))
