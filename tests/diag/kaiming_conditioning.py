"""Diagnostic (not collected by pytest): how well-conditioned is one training step of the Kaiming-initialised model bench.py trains?
The fp32 step on the same batch with (a) the input images rounded to bf16, (b) the parameters rounded to bf16, (c) bf16 MFMA operands
(ENDO_OPT_MFMA_BF16) -- perturbations of 2^-9 relative -- next to the unperturbed step and the bf16-storage step."""
import copy, importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda", 0)
for (n, h, w) in [(2, 64, 96), (8, 256, 320)]:
    torch.manual_seed(5)
    m0 = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(m0, mode="fan_in", activation_mode="relu", distribution="normal")
    batch0 = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=11).items()}
    def run(tag, model, batch, **kw):
        m = copy.deepcopy(model).to(dev).train()
        if kw.pop("bf16_operands", False):
            m.set_kernel_option(4, 1)
        o = ea.train_step.TrainingStep(m, ea.optim.FusedClipSGD(m, lr=1.0e-3), h, w, **kw)(batch, lr=1.0e-3)
        torch.cuda.synchronize()
        print("%d x %d x %d  %-28s loss %.6f  sfl %.6f  dcl %.6f  grad_norm %.4e" % (n, h, w, tag, o["loss"], o["sfl"], o["dcl"], float(o["grad_norm"])))
    run("fp32", m0, batch0)
    b1 = dict(batch0)
    for k in ("colors_1", "colors_2"):
        if k in b1: b1[k] = b1[k].bfloat16().float()
    run("fp32, images rounded to bf16", m0, b1)
    m1 = copy.deepcopy(m0)
    with torch.no_grad():
        for p in m1.parameters(): p.copy_(p.bfloat16().float())
    run("fp32, parameters rounded", m1, batch0)
    run("fp32 storage, bf16 operands", m0, batch0, bf16_operands=True)
    run("bf16 storage", m0, batch0, bf16_storage=True)
    run("fp16 storage", m0, batch0, fp16_storage=True)
print(sorted(batch0.keys()))
