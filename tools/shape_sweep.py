"""Run full training steps at the other BASELINE.json shapes (development check): finite loss, step time."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
for n, h, w in ((1, 256, 320), (4, 512, 640), (8, 256, 320), (2, 64, 96), (3, 96, 160)):
    torch.manual_seed(10085)
    model = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
    model = model.to(dev).train()
    opt = ea.optim.FusedClipSGD(model, lr=1e-3)
    step = ea.train_step.TrainingStep(model, opt, h, w)
    batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=0).items()}
    out = None
    for i in range(5):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = step(batch, lr=1e-3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("N=%d %dx%d: %.2f ms/step, %.1f pairs/s, loss %s" % (n, h, w, ms, n / ms * 1e3, {k: (round(float(v), 5) if hasattr(v, "__float__") else v) for k, v in out.items() if k in ("loss", "skipped")}))
    del model, opt, step, batch
    torch.cuda.empty_cache()
