"""Diagnostic: the deferred BatchNorm terms P, Q of one map (denseBlocksUp.4.layers.2's 12 maps at level-0 channels 168..179, whose only
BatchNorm consumer is layer 3 of the block) in the bf16 backward against the oracle's exact values; and the pixel sums of the map."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
from oracle import network as onet
from device_pattern16 import pattern_of, level_buffer
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 256, 320)
dev = torch.device("cuda:0")
state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72))
rng = np.random.default_rng(29)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
g = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
m = ea.FCDenseNet57(1); m.load_state_dict(state); m = m.to(dev).train()
y = m.forward_bf16_storage(x.to(dev))
pattern = pattern_of(y, m, n, h, w)
tape = y.grad_fn.tape
y.backward(g.to(dev)); torch.cuda.synchronize()
lib = ea._lib.load()
hnd = m._handle16(n, h, w, 1)[0]
ws = m._gradws[("bf16", n, h, w, 1)]
t0 = int(lib.endo_net16_offset(hnd, 5, 0))
off = int(lib.endo_net16_offset(hnd, 6, 0))
pq = ws.cpu().numpy()[off:off + 8 * t0].view(np.float32).copy()
P, Q = pq[:t0], pq[t0:2 * t0]
# the oracle, with traces
st64 = {k: (v.double().requires_grad_(k in onet.trainable_names()) if v.is_floating_point() else v.clone()) for k, v in state.items()}
trace = {}
y64 = onet.forward(st64, x.double(), training=True, quant=onet.bf16_ste, trace=trace, pattern=pattern)
for v in trace.values():
    if v.requires_grad: v.retain_grad()
y64.backward(g.double())
want = trace["conv::denseBlocksUp.4.layers.2"].grad             # total gradient of channels 168..179
got = level_buffer(lib, hnd, ws, 4, 0, n, h, w)[:, 168:180].double()
xs = level_buffer(lib, hnd, tape, 3, 0, n, h, w)[:, 168:180].double()
M = n * h * w
print("channel   sum want      sum got     (got-want)/M / rms    P (hip)        Q (hip)       mean x     std x")
for c in range(12):
    d = got[:, c] - want[:, c]
    print("%3d  %12.4f %12.4f   %10.3e   %12.5e %12.5e   %9.4f %9.4f" % (168 + c, float(want[:, c].sum()), float(got[:, c].sum()), float(d.mean() / want[:, c].pow(2).mean().sqrt()),
          P[168 + c], Q[168 + c], float(xs[:, c].mean()), float(xs[:, c].std())))
# regress the error on (1, x - mean): which offset and slope would explain it
for c in range(12):
    d = (got[:, c] - want[:, c]).flatten(); xc = xs[:, c].flatten(); xm = xc - xc.mean()
    a = float(d.mean()); b = float((d * xm).sum() / (xm * xm).sum())
    print("   ch %d: error = %.3e + %.3e (x - mean) + rest; P*std %.3e; rest rms %.3e (total rms %.3e)" % (168 + c, a, b, abs(P[168 + c]) * float(xc.std()), float((d - a - b * xm).pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt())))

# ---- the same P, Q rebuilt in fp64 from the HIP pass's OWN buffers: prepared gradient of layer 3's output (channels 180..191), its
# bf16 weights, the pass's ReLU mask and stored x: what bn_finalize should have produced for channels 168..179
import torch.nn.functional as F
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
pre = "denseBlocksUp.4.layers.3"
G3 = level_buffer(lib, hnd, ws, 4, 0, n, h, w)[:, 180:192].double()
W3 = sd[pre + ".conv.weight"].to(torch.bfloat16).double()                      # [12][180][3][3], reference channel order
dz = F.conv_transpose2d(G3, W3, padding=1)                                      # gradient w.r.t. relu(bn(x)), reference order
mask = pattern["relu::" + pre + ".norm"].double()
da = (dz * mask)[:, 168:180]
saved_off = int(lib.endo_net16_offset(hnd, 1, 48))                              # BN index: 20 down + 5 td + 4 bott + 4*4 up blocks 0..3 + 3 = 48
sv = tape.cpu().numpy()[saved_off:saved_off + 8 * 180].view(np.float32).reshape(180, 2)
gamma = sd[pre + ".norm.weight"].double()
for c in range(12):
    ci = 168 + c
    mean, rstd = float(sv[ci, 0]), float(sv[ci, 1])
    xc = xs[:, c]
    s1 = float(da[:, c].sum()); s2 = float((da[:, c] * (xc - mean) * rstd).sum())
    sc = float(gamma[ci]) * rstd
    k = sc * rstd * s2 / M
    print("   ch %d: P rebuilt %.5e (hip %.5e)   Q rebuilt %.5e (hip %.5e)   S1 %.4e  S2 %.4e  sum|da| %.4e" % (
        ci, -k, P[ci], -sc * s1 / M + k * mean, Q[ci], s1, s2, float(da[:, c].abs().sum())))

# ---- the oracle's own decomposition of the same 12 maps: final-convolution part F, layer 3's masked data gradient D = scale * da, and
# what is left = its BatchNorm correction; against the HIP pass's three parts
G3o = trace["conv::" + pre].grad                                                 # oracle: gradient of layer 3's output
W3o = st64[pre + ".conv.weight"].detach().to(torch.bfloat16).double()
dzo = F.conv_transpose2d(G3o, W3o, padding=1)
dao = (dzo * mask)[:, 168:180]
xo_all = trace["skip_0"].detach() if False else None
# oracle's stored x of these channels = output of layer 2 (quantised)
xo = trace["conv::denseBlocksUp.4.layers.2"].detach()
pre_o = None
gs_h = (g.double() * torch.sign(torch.from_numpy(tape.cpu().numpy()[int(lib.endo_net16_offset(hnd, 0, 0)):int(lib.endo_net16_offset(hnd, 0, 0)) + 4 * n * h * w].view(np.float32).copy()).view(n, 1, h, w).double()))
wf = sd["finalConv.weight"].double().view(-1)                                     # reference order: [TU 48 | skip 96 | new 48] -> new maps at 144..191
for c in (1, 3, 9):
    ci = 168 + c
    mean_o, var_o = float(xo[:, c].mean()), float(xo[:, c].var(unbiased=False))
    rstd_o = (var_o + 1e-5) ** -0.5
    sc_o = float(gamma[ci]) * rstd_o
    Fo = gs_h[:, 0] * float(wf[ci])                                               # same sign pattern and output gradient on both sides
    Do = sc_o * dao[:, c]
    bn_o = want[:, c] - Fo - Do
    xm = (xo[:, c] - mean_o)
    b_o = float((bn_o * xm).sum() / (xm * xm).sum()); a_o = float(bn_o.mean())
    resid = float((bn_o - a_o - b_o * xm).pow(2).mean().sqrt())
    mean_h, rstd_h = float(sv[ci, 0]), float(sv[ci, 1])
    print("   ch %d oracle: BN part = %.5e + %.5e (x - mean) (residual rms %.2e);  hip: Q + P mean = %.5e, P = %.5e;  S1 oracle %.4e  S2 oracle %.4e" % (
        ci, a_o, b_o, resid, Q[ci] + P[ci] * mean_h, P[ci], float(dao[:, c].sum()), float((dao[:, c] * xm * rstd_o).sum())))
