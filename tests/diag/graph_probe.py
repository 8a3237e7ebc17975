"""Development probe: does capturing one training iteration (forward, loss head, backward, clip + SGD) into a HIP graph and
replaying it beat issuing the ~600 launches from the host?  (The guard of train.py:317 needs the loss on the host, so a graphed
step would need the guard on the device; this probe only times the launches.)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
torch.manual_seed(10085)
model = ea.FCDenseNet57(1)
ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
model = model.to(dev).train()
opt = ea.optim.FusedClipSGD(model, lr=1e-4)
step = ea.train_step.TrainingStep(model, opt, 256, 320)
batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(8, 256, 320, seed=0).items()}

def one():
    opt.zero_grad()
    losses_t, x, tape, pred, grad_pred = step._fused_iteration(batch)
    step._fused_backward(x, tape, grad_pred)
    opt.step(grad_scale=1.0)
    return losses_t

for _ in range(5):
    one()
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager, no host sync: %.3f ms/step" % timeit(one))
print("TrainingStep (host guard): %.3f ms/step" % timeit(lambda: step(batch)))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): one()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = one()
    torch.cuda.synchronize()
    print("graph replay: %.3f ms/step" % timeit(g.replay), "loss", out.tolist())
except Exception as exc:
    print("graph capture failed:", repr(exc)[:500])
