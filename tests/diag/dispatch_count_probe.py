"""Development probe: ways of counting the kernel dispatches of one training iteration in-process (bench.py's dispatches_per_step)."""
import importlib, os, sys, tempfile, torch
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
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
try:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(batch)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if str(e.device_type).endswith("CUDA")]
    names = {}
    for e in evs:
        names[e.name] = names.get(e.name, 0) + 1
    print("torch.profiler: %d device events in one iteration; %d distinct names" % (len(evs), len(names)))
    for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:12]:
        print("   %4d  %s" % (v, k[:100]))
except Exception as exc:
    print("torch.profiler failed:", repr(exc)[:300])
try:
    def one():
        opt.zero_grad()
        losses_t, x, tape, pred, grad_pred = step._fused_iteration(batch)
        step._fused_backward(x, tape, grad_pred)
        opt.step(grad_scale=1.0)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        one()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); g.enable_debug_mode()
    with torch.cuda.graph(g):
        one()
    path = os.path.join(tempfile.gettempdir(), "step_graph.dot")
    g.debug_dump(path)
    print("debug_dump wrote", os.path.exists(path), os.path.getsize(path) if os.path.exists(path) else 0)
    if os.path.exists(path):
        print(open(path).read()[:1500])
except Exception as exc:
    print("graph dump failed:", repr(exc)[:300])
