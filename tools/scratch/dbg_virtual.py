import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import make_model, kernel_options, dev, OPT_WINO_MIN_TILES, OPT_FINAL_VIRTUAL
n, h, w = 2, 64, 96
rng = np.random.default_rng(23)
xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
def run(virtual, wd):
    with kernel_options({OPT_WINO_MIN_TILES: 1}):
        _, model = make_model(66)
    model.set_kernel_option(OPT_FINAL_VIRTUAL, virtual)
    model.set_kernel_option(1, wd)
    model.train()
    y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
    ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
    torch.cuda.synchronize()
    return {nm: p.grad.detach().clone() for nm, p in model.named_parameters()}
res = {(v, wd): run(v, wd) for v in (1, 0) for wd in (1, 3)}
def cmp(a, b, label):
    gmax = max(float(v.abs().max()) for v in b.values())
    rows = []
    for nm in a:
        d = float((a[nm] - b[nm]).abs().max()); sc = max(float(b[nm].abs().max()), 1e-3 * gmax)
        rows.append((d / sc, nm, d))
    rows.sort(reverse=True)
    print(label, "worst:", ["%s %.2e (abs %.2e)" % (nm, r, d) for r, nm, d in rows[:4]], "exactly equal tensors: %d of %d" % (sum(1 for r in rows if r[2] == 0.0), len(rows)))
cmp(res[(1, 1)], res[(0, 1)], "old kernel, virtual on vs off")
cmp(res[(1, 3)], res[(0, 3)], "persistent, virtual on vs off")
cmp(res[(1, 3)], res[(1, 1)], "virtual on, persistent vs old")
cmp(res[(0, 3)], res[(0, 1)], "virtual off, persistent vs old")
r2 = run(1, 3)
cmp(r2, res[(1, 3)], "persistent virtual on, run twice")
