"""Parity of the HIP path (through the C ABI, via the drop-in modules) against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full size --
through size-independent properties.  Tolerance: BASELINE.json north_star = 1e-4 relative (fp32);
each assert states the tolerance it uses.  Run with ``pytest -m gpu`` on an MI355X."""

import ctypes
import importlib

import numpy as np
import pytest
import torch

from oracle import geometry as ogeo, losses as olos, network as onet, schedule as osch, train_step as ostep
from device_pattern import pattern_of

pytestmark = pytest.mark.gpu

ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
synthetic = ea.synthetic


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def rel_err(got, want):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    scale = max(float(want.abs().max()), 1e-30)
    return float((got - want).abs().max()) / scale


def assert_close(got, want, tol, what):
    err = rel_err(got, want)
    assert err <= tol, "%s: max abs err / max |ref| = %.3e > %.1e" % (what, err, tol)


def noise_aware(got, ref32, ref64, what, floor=1e-4, factor=4.0, scale=None):
    """fp32 training-mode BatchNorm over a handful of samples is ill-conditioned, and ReLU masks /
    max-pool argmax / |.| are discontinuous: the reference's own fp32 CPU path differs from an fp64
    evaluation of the same graph by up to ~2e-3 on early-layer gradients at test sizes, with isolated
    elements off by O(1/sqrt(#pixels)) where a single mask bit flipped (measured: tests/diag/gpu_diag2.py).
    So the bar is "as close to the fp64 oracle as the fp32 CPU oracle is":
      * 90th percentile of |hip - fp64| <= max(factor x the same percentile for cpu32, floor)
      * max |hip - fp64| <= max(15 x max for cpu32, 5e-2)          (a few flipped mask bits)
    `floor` = 1e-4 is BASELINE.json's tolerance; nothing is asked to be closer than that."""
    ref64 = ref64.detach().double().cpu().reshape(-1)
    if scale is None:
        scale = max(float(ref64.abs().max()), 1e-30)
    d_hip = (got.detach().double().cpu().reshape(-1) - ref64).abs() / scale
    d_cpu = (ref32.detach().double().cpu().reshape(-1) - ref64).abs() / scale
    q = 0.9 if ref64.numel() >= 10 else 1.0
    q_hip, q_cpu = float(torch.quantile(d_hip, q)), float(torch.quantile(d_cpu, q))
    m_hip, m_cpu = float(d_hip.max()), float(d_cpu.max())
    assert q_hip <= max(factor * q_cpu, floor), "%s: q90 hip-vs-fp64 %.3e, cpu32-vs-fp64 %.3e" % (what, q_hip, q_cpu)
    assert m_hip <= max(15.0 * m_cpu, 5e-2), "%s: max hip-vs-fp64 %.3e, cpu32-vs-fp64 %.3e" % (what, m_hip, m_cpu)
    return m_hip, m_cpu


def state_as(state, dtype):
    return {k: (v.detach().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}


def geometry_inputs(n, h, w, seed):
    batch = synthetic.make_batch(n, h, w, seed=seed, sparse_points=min(500, h * w // 6))
    p1 = synthetic.smooth_depth(n, h, w, seed=seed + 100)
    p2 = synthetic.smooth_depth(n, h, w, seed=seed + 200)
    goal = synthetic.smooth_depth(n, h, w, seed=seed + 300)
    return batch, p1, p2, goal


def to_dev(d):
    return {k: v.to(dev()) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------
# geometry layers and losses, forward + backward
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 16, 20), (3, 64, 96), (1, 256, 320), (2, 37, 53)])
def test_depth_scaling(shape):
    n, h, w = shape
    batch, p1, _, _ = geometry_inputs(n, h, w, 41)
    cot = torch.from_numpy(np.random.default_rng(1).standard_normal((n, 1, h, w)).astype(np.float32))
    pc = p1.clone().requires_grad_(True)
    s_ref, r_ref = ogeo.depth_scaling(pc, batch["sparse_depths_1"], batch["sparse_depth_masks_1"])
    ((s_ref * cot).sum() + 3.0 * r_ref).backward()
    pg = p1.to(dev()).requires_grad_(True)
    s, r = ea.DepthScalingLayer(epsilon=1.0e-8)([pg, batch["sparse_depths_1"].to(dev()), batch["sparse_depth_masks_1"].to(dev())])
    ((s * cot.to(dev())).sum() + 3.0 * r).backward()
    assert_close(s, s_ref, 1e-5, "scaled depth")
    assert_close(r, r_ref, 1e-4, "scale std/mean ratio")
    assert_close(pg.grad, pc.grad, 1e-4, "grad pred")
    # only the scaled output used (the training path): grad_ratio is None
    pg2 = p1.to(dev()).requires_grad_(True)
    s2, _ = ea.DepthScalingLayer()([pg2, batch["sparse_depths_1"].to(dev()), batch["sparse_depth_masks_1"].to(dev())])
    (s2 * cot.to(dev())).sum().backward()
    pc2 = p1.clone().requires_grad_(True)
    s_ref2, _ = ogeo.depth_scaling(pc2, batch["sparse_depths_1"], batch["sparse_depth_masks_1"])
    (s_ref2 * cot).sum().backward()
    assert_close(pg2.grad, pc2.grad, 1e-4, "grad pred (scaled only)")


@pytest.mark.parametrize("shape", [(2, 16, 20), (3, 64, 96), (1, 256, 320), (2, 37, 53)])
def test_flow_from_depth(shape):
    n, h, w = shape
    batch, p1, _, _ = geometry_inputs(n, h, w, 42)
    cot = torch.from_numpy(np.random.default_rng(2).standard_normal((n, 2, h, w)).astype(np.float32))
    args = [batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]]
    dc = p1.clone().requires_grad_(True)
    f_ref = ogeo.flow_from_depth(dc, *args)
    (f_ref * cot).sum().backward()
    dg = p1.to(dev()).requires_grad_(True)
    f = ea.FlowfromDepthLayer()([dg] + [a.to(dev()) for a in args])
    (f * cot.to(dev())).sum().backward()
    assert_close(f, f_ref, 1e-5, "flow")
    assert_close(dg.grad, dc.grad, 1e-4, "grad depth")


@pytest.mark.parametrize("shape", [(2, 16, 20), (3, 64, 96), (1, 256, 320), (2, 37, 53)])
def test_depth_warping(shape):
    n, h, w = shape
    batch, p1, p2, _ = geometry_inputs(n, h, w, 43)
    cot = torch.from_numpy(np.random.default_rng(3).standard_normal((n, 1, h, w)).astype(np.float32))
    args = [batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]]
    c1 = p1.clone().requires_grad_(True)
    c2 = p2.clone().requires_grad_(True)
    w_ref, overlap = ogeo.depth_warping_parts(c1, c2, *args)
    (w_ref * cot).sum().backward()
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    warped, inter = ea.DepthWarpingLayer(epsilon=1.0e-8)([g1, g2] + [a.to(dev()) for a in args])
    (warped * cot.to(dev())).sum().backward()
    assert_close(warped, w_ref, 5e-5, "warped depth")
    assert_close(g1.grad, c1.grad, 1e-4, "grad depth 1")
    assert_close(g2.grad, c2.grad, 1e-4, "grad depth 2")
    # the intersect mask is a threshold at 0.9: compare away from numerically borderline pixels
    clear = (overlap.detach() - 0.9).abs() > 1e-5
    want = (overlap.detach() >= 0.9).float()
    assert torch.equal(inter.cpu()[clear], want[clear]), "intersect mask"
    assert not inter.requires_grad


def test_warp_edge_cases():
    """Masked-out pixels (eps division), points behind the camera, out-of-frame samples."""
    n, h, w = 2, 24, 32
    batch, p1, p2, _ = geometry_inputs(n, h, w, 44)
    t = batch["translations_1_wrt_2"].clone()
    t[0, 2, 0] = 2.0      # pushes z2 <= 0 for sample 0 (depth ~0.3-0.9): the where(z>0) branch
    t[1, 0, 0] = 0.8      # large lateral motion: most samples fall outside the frame
    args = [batch["boundaries"], t, batch["rotations_1_wrt_2"], batch["intrinsics"]]
    cot = torch.ones(n, 1, h, w)
    c1 = p1.clone().requires_grad_(True)
    c2 = p2.clone().requires_grad_(True)
    w_ref, overlap = ogeo.depth_warping_parts(c1, c2, *args)
    (w_ref * cot).sum().backward()
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    warped, inter = ea.DepthWarpingLayer()([g1, g2] + [a.to(dev()) for a in args])
    (warped * cot.to(dev())).sum().backward()
    assert torch.isfinite(warped).all()
    assert_close(warped, w_ref, 5e-5, "warped (edge cases)")
    assert_close(g1.grad, c1.grad, 1e-4, "grad d1 (edge cases)")
    assert_close(g2.grad, c2.grad, 1e-4, "grad d2 (edge cases)")
    f_ref = ogeo.flow_from_depth(p1, *args)
    f = ea.FlowfromDepthLayer()([p1.to(dev())] + [a.to(dev()) for a in args])
    finite = torch.isfinite(f_ref)
    assert torch.equal(torch.isfinite(f.cpu()), finite)            # the flow layer may emit inf/nan; so must we
    assert_close(torch.where(finite, f.cpu(), torch.zeros_like(f_ref)).clamp(-1e6, 1e6),
                 torch.where(finite, f_ref, torch.zeros_like(f_ref)).clamp(-1e6, 1e6), 1e-4, "flow (edge cases)")


@pytest.mark.parametrize("shape", [(2, 16, 20), (3, 64, 96), (1, 256, 320)])
def test_losses(shape):
    n, h, w = shape
    batch, p1, p2, goal = geometry_inputs(n, h, w, 45)
    b = batch["boundaries"]
    rng = np.random.default_rng(4)
    hat = torch.from_numpy(rng.normal(0, 0.03, (n, 2, h, w)).astype(np.float32))
    inter = (torch.from_numpy(rng.uniform(0, 1, (n, 1, h, w)).astype(np.float32)) > 0.3).float() * b
    # sparse flow loss
    hc = hat.clone().requires_grad_(True)
    l_ref = olos.sparse_masked_l1(batch["sparse_flows_1"], hc, batch["sparse_flow_masks_1"])
    l_ref.backward()
    hg = hat.to(dev()).requires_grad_(True)
    loss = ea.SparseMaskedL1Loss()([batch["sparse_flows_1"].to(dev()), hg, batch["sparse_flow_masks_1"].to(dev())])
    loss.backward()
    assert_close(loss, l_ref, 1e-5, "sparse flow loss")
    assert_close(hg.grad, hc.grad, 1e-5, "sparse flow loss grad")
    # depth consistency loss
    c1 = p1.clone().requires_grad_(True)
    c2 = p2.clone().requires_grad_(True)
    l_ref = olos.normalized_distance(c1, c2, inter, batch["intrinsics"])
    l_ref.backward()
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    loss = ea.NormalizedDistanceLoss(height=h, width=w)([g1, g2, inter.to(dev()), batch["intrinsics"].to(dev())])
    loss.backward()
    assert_close(loss, l_ref, 1e-5, "depth consistency loss")
    assert_close(g1.grad, c1.grad, 1e-4, "dcl grad depth")
    assert_close(g2.grad, c2.grad, 1e-4, "dcl grad warped")
    # scale invariant loss
    c1 = p1.clone().requires_grad_(True)
    l_ref = olos.scale_invariant(c1, goal, b)
    l_ref.backward()
    g1 = p1.to(dev()).requires_grad_(True)
    loss = ea.ScaleInvariantLoss(epsilon=1.0e-8)([g1, goal.to(dev()), b.to(dev())])
    loss.backward()
    assert_close(loss, l_ref, 2e-5, "scale invariant loss")
    assert_close(g1.grad, c1.grad, 1e-4, "scale invariant grad")


def test_geometry_golden(golden):
    """HIP geometry + losses against the fixture the REFERENCE produced (tests/golden/make_golden.py)."""
    g = golden("geometry_3x64x96.npz")
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    batch, p1, p2, goal = geometry_inputs(n, h, w, seed)
    batch = to_dev(batch)
    p1 = p1.to(dev()).requires_grad_(True)
    p2 = p2.to(dev()).requires_grad_(True)
    b = batch["boundaries"]
    scaling, flow_layer, warp_layer = ea.DepthScalingLayer(), ea.FlowfromDepthLayer(), ea.DepthWarpingLayer()
    s1, std1 = scaling([p1, batch["sparse_depths_1"], batch["sparse_depth_masks_1"]])
    s2, std2 = scaling([p2, batch["sparse_depths_2"], batch["sparse_depth_masks_2"]])
    f1 = flow_layer([s1, b, batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
    f2 = flow_layer([s2, b, batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
    sfl_fn, dcl_fn, sil_fn = ea.SparseMaskedL1Loss(), ea.NormalizedDistanceLoss(h, w), ea.ScaleInvariantLoss()
    sfl = 0.5 * (sfl_fn([batch["sparse_flows_1"] * b, f1 * b, batch["sparse_flow_masks_1"] * b]) +
                 sfl_fn([batch["sparse_flows_2"] * b, f2 * b, batch["sparse_flow_masks_2"] * b]))
    w21, i1 = warp_layer([s1, s2, b, batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
    w12, i2 = warp_layer([s2, s1, b, batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
    dcl = 0.5 * (dcl_fn([s1, w21, i1, batch["intrinsics"]]) + dcl_fn([s2, w12, i2, batch["intrinsics"]]))
    sil = sil_fn([p1, goal.to(dev()), b])
    total = 20.0 * sfl + 0.1 * dcl + 0.3 * sil + 0.05 * (std1 + std2)
    total.backward()
    for key, val in (("sfl", sfl), ("dcl", dcl), ("sil", sil), ("total", total), ("std_1", std1)):
        assert_close(val, torch.from_numpy(g[key]), 1e-4, key)
    for key, val in (("scaled_1", s1), ("flow_1", f1), ("flow_2", f2), ("warped_21", w21), ("warped_12", w12),
                     ("grad_pred_1", p1.grad), ("grad_pred_2", p2.grad)):
        assert_close(val, torch.from_numpy(g[key]), 1e-4, key)
    assert float((i1.cpu() != torch.from_numpy(g["inter_1"])).float().mean()) < 1e-4


# ---------------------------------------------------------------------------------------------
# network
# ---------------------------------------------------------------------------------------------
MODEL_OPTIONS = {}          # kernel options (include/endo_hip.h ENDO_OPT_*) given to the models make_model builds: see kernel_options


def make_model(seed, positive_depth=False):
    state = onet.perturb_affine(onet.synthetic_state(seed), seed + 1)
    if positive_depth:
        onet.keep_depth_positive(state)
    model = ea.FCDenseNet57(n_classes=1)
    missing = model.load_state_dict(state)
    assert not missing.missing_keys and not missing.unexpected_keys
    for option_id, value in MODEL_OPTIONS.items():
        model.set_kernel_option(option_id, value)
    return state, model.to(dev())


def level_reference(trace, level):
    """Assemble the oracle's view of level buffer `level` (layout: DESIGN.md / net.hip header)."""
    if level == 5:
        return torch.cat([trace["bott_in"], trace["bott_new"]], dim=1)
    return torch.cat([trace["tu_%d" % level], trace["skip_%d" % level], trace["upnew_%d" % level]], dim=1)


@pytest.mark.parametrize("shape", [(2, 32, 32), (2, 64, 96), (1, 128, 160)])
def test_network_forward_levels(shape):
    n, h, w = shape
    state, model = make_model(51)
    x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    trace, trace64 = {}, {}
    state64 = state_as(state, torch.float64)
    y_ref = onet.forward(state, x, training=True, trace=trace)
    y_64 = onet.forward(state64, x.double(), training=True, trace=trace64)
    model.train()
    with torch.no_grad():
        y, levels = model.level_buffers(x.to(dev()))
    for lvl in (0, 1, 2, 3, 4, 5):
        ref, ref64 = level_reference(trace, lvl), level_reference(trace64, lvl)
        got = levels[lvl].cpu()
        for c0 in range(0, ref.shape[1], 12):      # per 12-channel slice: localises a faulty layer
            noise_aware(got[:, c0:c0 + 12], ref[:, c0:c0 + 12], ref64[:, c0:c0 + 12],
                        "level %d channels [%d,%d)" % (lvl, c0, c0 + 12))
    noise_aware(y, y_ref, y_64, "network output")
    sd = model.state_dict()
    for name in ("denseBlocksDown.0.layers.0.norm", "transDownBlocks.2.norm", "bottleneck.bottleneck.layers.3.norm",
                 "denseBlocksUp.4.layers.3.norm"):
        noise_aware(sd[name + ".running_mean"], state[name + ".running_mean"], state64[name + ".running_mean"], name + ".running_mean")
        noise_aware(sd[name + ".running_var"], state[name + ".running_var"], state64[name + ".running_var"], name + ".running_var")
        assert int(sd[name + ".num_batches_tracked"]) == 1
    model.eval()
    with torch.no_grad():
        y_eval = model(x.to(dev()))
        noise_aware(y_eval, onet.forward(state, x, training=False), onet.forward(state64, x.double(), training=False),
                    "eval-mode output")


def reference_grads(state, x, cot, dtype, pattern=None):
    """Oracle parameter gradients; with ``pattern`` on the piecewise-linear branch a HIP forward pass took (device_pattern.py)."""
    st = state_as(state, dtype)
    names = onet.trainable_names()
    for nm in names:
        st[nm].requires_grad_(True)
    y = onet.forward(st, x.to(dtype), training=True, pattern=pattern)
    grads = torch.autograd.grad((y * cot.to(dtype)).sum(), [st[nm] for nm in names])
    return dict(zip(names, grads))


def assert_grads_on_pattern(params, g64, g32, tol, what):
    """Every parameter gradient against the fp64 oracle evaluated on the HIP pass's own activation pattern: max abs error
    <= tol x the tensor's scale.  g32 (the fp32 CPU oracle on the same pattern) is only reported, to show where plain fp32
    rounding sits."""
    report = []
    for nm in onet.trainable_names():
        got = params[nm].grad
        assert got is not None, nm
        scale = grad_scale(g64, nm)
        e_hip = float((got.detach().double().cpu() - g64[nm]).abs().max()) / scale
        e_cpu = float((g32[nm].double() - g64[nm]).abs().max()) / scale if g32 is not None else float("nan")
        report.append((e_hip, e_cpu, nm))
    report.sort(reverse=True)
    print("%s: worst gradient errors on the shared pattern (hip-vs-fp64, cpu32-vs-fp64):" % what,
          ["%.2e %.2e %s" % r for r in report[:5]])
    bad = [r for r in report if not r[0] <= tol]
    assert not bad, "%s: %d tensors over %.1e: %s" % (what, len(bad), tol, ["%.2e %s" % (r[0], r[2]) for r in bad[:8]])
    return report


def grad_scale(ref64, name):
    """Scale a gradient tensor is judged on.  A conv bias that feeds a training-mode BN has an exactly
    zero true gradient (fp32 leaves ~1e-7 noise), so biases are judged on the scale of their weight."""
    own = float(ref64[name].abs().max())
    if name.endswith(".bias"):
        sibling = name[:-5] + ".weight"
        own = max(own, float(ref64[sibling].abs().max()))
    return max(own, 1e-30)


GRAD_TOL = 3e-5          # every parameter gradient: max abs error / the tensor's max, against the fp64 oracle on the HIP pass's
                         # own activation pattern.  Measured: <= 1.0e-5 on every tensor at every size below, 1.0-2x the fp32 CPU
                         # oracle's own distance from fp64 on the same pattern; BASELINE.json's bar is 1e-4.


@pytest.mark.parametrize("shape", [(2, 64, 96)])          # (2, 128, 160): test_network_backward_kernel_forms runs it in three kernel forms
def test_network_backward(shape):
    """All 210 parameter gradients against the fp64 oracle at 1e-4 -- evaluated on the activation pattern the HIP forward
    pass itself took (device_pattern.py).  Without that, the comparison is a lottery: any two finite-precision runs of a
    ReLU network disagree on the few mask bits whose pre-activation lies within rounding of zero, each an O(1) change of
    that pixel's gradient (tests/diag/gpu_diag7.py found exactly the three borderline elements behind round 1's
    "8x noisier" late-layer gradient).  The unconstrained comparison is kept below as a statistic with the round-1 bound."""
    n, h, w = shape
    state, model = make_model(52)
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    model.train()
    y = model(x.to(dev()))
    (pattern,) = pattern_of(y, model, n, h, w)
    (y * cot.to(dev())).sum().backward()
    params = dict(model.named_parameters())
    g64p = reference_grads(state, x, cot, torch.float64, pattern)
    g32p = reference_grads(state, x, cot, torch.float32, pattern)
    assert_grads_on_pattern(params, g64p, g32p, GRAD_TOL, "network backward %s" % (shape,))
    flat64 = torch.cat([g64p[nm].reshape(-1) for nm in onet.trainable_names()])
    first = model.flat_gradients().clone()
    assert_close(first, flat64, GRAD_TOL, "flat gradient vector")
    # how many bits of the pattern differ from the fp64 oracle's own: a handful (the lottery the pattern removes)
    trace = {}
    own = reference_pattern(state_as(state, torch.float64), x.double())
    flips = sum(int((own[k] != pattern[k]).sum()) for k in pattern if k.startswith("relu::"))
    total = sum(pattern[k].numel() for k in pattern if k.startswith("relu::"))
    print("ReLU bits that differ between the HIP pass and the fp64 oracle: %d of %d" % (flips, total))
    assert flips <= 1e-5 * total
    # the unconstrained comparison, as a statistic (round-1 criterion, factor 4 again)
    g32 = reference_grads(state, x, cot, torch.float32)
    g64 = reference_grads(state, x, cot, torch.float64)
    flat32 = torch.cat([g32[nm].reshape(-1) for nm in onet.trainable_names()])
    flat64u = torch.cat([g64[nm].reshape(-1) for nm in onet.trainable_names()])
    noise_aware(first, flat32, flat64u, "flat gradient vector, own patterns", factor=4.0)
    # two backward passes accumulate (train.py:276-277 runs the network twice per step)
    y = model(x.to(dev()))
    (y * cot.to(dev())).sum().backward()
    assert_close(model.flat_gradients(), 2.0 * first, 1e-5, "accumulated gradient")


OPT_WINO_FWD, OPT_WINO_DGRAD, OPT_DGRAD_VEC, OPT_WINO_MIN_TILES, OPT_MFMA_BF16, OPT_WGRAD_OVERLAP, OPT_MFMA_X3, OPT_WGRAD_F34 = 0, 1, 2, 3, 4, 5, 6, 7


class kernel_options(object):
    """``with kernel_options({id: value}):`` -- every model make_model builds inside the block gets these options through
    FCDenseNet57.set_kernel_option (endo_net_set_option on each of its handles, include/endo_hip.h).  Options are per model:
    nothing outside the block, and no model built elsewhere, is affected."""

    def __init__(self, values):
        self.values, self.old = values, None

    def __enter__(self):
        self.old = dict(MODEL_OPTIONS)
        MODEL_OPTIONS.update(self.values)
        return self

    def __exit__(self, *exc):
        MODEL_OPTIONS.clear()
        MODEL_OPTIONS.update(self.old)
        return False


def test_kernel_options_are_per_model():
    """Two models in one process (reference train.py:191: modules are independent objects): switching one to the direct kernels,
    or to bf16 MFMA operands, must not change what the other computes -- bit for bit -- and the switched one must change."""
    n, h, w = 2, 64, 96
    _, a = make_model(81)
    _, b = make_model(81)
    x = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)).to(dev())
    a.train(); b.train()
    with torch.no_grad():
        ya0, yb0 = a(x).clone(), b(x).clone()
    assert torch.equal(ya0, yb0)
    assert b.set_kernel_option(OPT_MFMA_BF16, 1) == 0 and b.kernel_option(OPT_MFMA_BF16) == 1 and a.kernel_option(OPT_MFMA_BF16) == 0
    with torch.no_grad():
        yb1, ya1 = b(x).clone(), a(x).clone()
    assert torch.equal(ya1, ya0), "model a changed when model b's option was set"
    assert not torch.equal(yb1, yb0), "model b did not switch to bf16 operands"
    # a handle created AFTER the option was set (another input size) inherits it; the other model's new handle does not
    x2 = x[:1, :, :32, :64].contiguous()
    with torch.no_grad():
        ya2, yb2 = a(x2), b(x2)
    assert not torch.equal(ya2, yb2)
    b.set_kernel_option(OPT_MFMA_BF16, 0)
    with torch.no_grad():
        assert torch.equal(b(x2), ya2) and torch.equal(b(x), ya0)
    lib = ea._lib.load()
    assert lib.endo_net_set_option(None, OPT_MFMA_BF16, 1) == -1
    hnd = a._handle(n, h, w)[0]
    assert lib.endo_net_set_option(hnd, 99, 1) == -1 and lib.endo_net_get_option(hnd, OPT_WINO_MIN_TILES) == 1024


@pytest.mark.parametrize("which,shape", [("winograd", (2, 64, 96)), ("winograd", (2, 128, 160)), ("winograd", (1, 64, 128)),
                                         ("direct", (2, 64, 96)), ("direct", (2, 128, 160)), ("direct", (1, 64, 128)),
                                         ("winograd4", (2, 64, 96)), ("winograd4", (2, 128, 160)), ("winograd4", (1, 64, 128)),
                                         ("x3", (2, 128, 160))],          # x3: needs >= 2048 row chunks at level 0 to reach the n-split / x3 weight-gradient kernels
                         ids=lambda v: "x".join(str(i) for i in v) if isinstance(v, tuple) else v)
def test_network_backward_kernel_forms(shape, which):
    """The Winograd kernels (dense-layer forward, fused base-channel data gradient, F(3x3, 4x4) weight gradient where the height is a multiple of 16
    and the width of 4) are chosen by launch size and the parity
    tests above are too small to reach them: here they are forced on (ENDO_OPT_WINO_MIN_TILES = 1) or off for every eligible
    level and all 210 gradients are checked on the pass's own activation pattern as in test_network_backward -- so both forms
    of every such layer are held to the same 3e-5, at sizes the fp64 oracle finishes in seconds."""
    n, h, w = shape
    # "x3": the dense weight gradient with its fp32 products as three-term bf16 splits on the bf16 matrix cores (ENDO_OPT_MFMA_X3 bit 0,
    # csrc/wgrad_x3_kernels.h; DESIGN.md 4.15: not the default) -- the SAME function, held to the same fp32 bound
    # "winograd4": the dense-layer forward in F(4x4, 3x3) form (ENDO_OPT_WINO_FWD = 5, csrc/wino4_fwd_kernels.h; the default form of level 0 since
    # round 5) forced onto EVERY level whose height and width allow it.  With the interpolation points 0, +-5/8, +-3/2, inf of round 5 the depth
    # sits at 1.9e-6 .. 2.2e-6 of its maximum (round 4's textbook points: 4e-6 .. 6e-6; the other forms: 0.9e-6 .. 1.1e-6) and is held to the
    # same 1e-5 as every form; the gradients -- taken on the pass's own pattern, from activations that carry the forward's rounding --
    # measured 2.0e-5 .. 3.5e-5 (round 4: 5.8e-5 .. 1.1e-4; the other forms 0.7e-5 .. 2.2e-5) and are held to 5e-5 (round 4: 1.6e-4).
    opts = {OPT_WINO_MIN_TILES: 1, OPT_WINO_FWD: 1} if which == "winograd" else {OPT_WINO_MIN_TILES: 1, OPT_WINO_FWD: 5} if which == "winograd4" else ({OPT_MFMA_X3: 1} if which == "x3" else {OPT_WINO_FWD: 0, OPT_WINO_DGRAD: 0, OPT_DGRAD_VEC: 0, OPT_WGRAD_F34: 0})
    with kernel_options(opts):
        state, model = make_model(62)
        rng = np.random.default_rng(16)
        x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
        cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
        model.train()
        y = model(x.to(dev()))
        (pattern,) = pattern_of(y, model, n, h, w)
        (y * cot.to(dev())).sum().backward()
        torch.cuda.synchronize()
    params = dict(model.named_parameters())
    g64p = reference_grads(state, x, cot, torch.float64, pattern)
    y64 = onet.forward(state_as(state, torch.float64), x.double(), training=True, pattern=pattern)
    print("%s %s: depth max err / max |depth| = %.2e" % (which, shape, rel_err(y, y64)))
    assert_close(y, y64, 1e-5, "depth, %s kernels" % which)
    worst = assert_grads_on_pattern(params, g64p, None, 5e-5 if which == "winograd4" else GRAD_TOL, "network backward %s, %s kernels" % (shape, which))
    print("%s %s: worst gradient tensor %.2e %s" % (which, shape, worst[0][0], worst[0][2]))


def test_network_backward_eval_mode():
    """Backward through the network in eval mode (running statistics, no batch-statistic terms in the BN backward; what a caller
    fine-tuning with frozen BN would run -- endo_net_bwd(training = 0)): all 210 gradients against the fp64 oracle on the pass's
    own activation pattern, the bound of the training-mode test."""
    n, h, w = 2, 64, 96
    state, model = make_model(68)
    rng = np.random.default_rng(19)
    # running statistics that are not the initial (0, 1): take them from one training-mode pass of the oracle
    warm = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    onet.forward(state, warm, training=True)
    _, model = make_model(68)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name in state and buf.dtype.is_floating_point:
                buf.copy_(state[name].to(buf.device))
    model._flatten()          # gather the edited buffers into the flat BN table the library reads
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    model.eval()
    y = model(x.to(dev()))
    (pattern,) = pattern_of(y, model, n, h, w)
    (y * cot.to(dev())).sum().backward()
    torch.cuda.synchronize()
    st64 = state_as(state, torch.float64)
    names = onet.trainable_names()
    for nm in names:
        st64[nm].requires_grad_(True)
    y64 = onet.forward(st64, x.double(), training=False, pattern=pattern)
    g64 = dict(zip(names, torch.autograd.grad((y64 * cot.double()).sum(), [st64[nm] for nm in names])))
    assert_close(y, y64.detach(), 1e-5, "eval-mode depth")
    assert_grads_on_pattern(dict(model.named_parameters()), g64, None, GRAD_TOL, "network backward, eval mode")


BF16_FWD_TOL = 2e-2       # bf16-operand mode (ENDO_OPT_MFMA_BF16): depth against the fp64 oracle on the pass's own pattern, max error / max
BF16_GRAD_TOL = 1e-1      # ... and every parameter gradient, max error / the tensor's max (operands carry 8 significant bits;
                          # measured: depth 6e-3 / 1e-2, gradient tensors median 9.5e-3 / 7e-3, worst 6.0e-2 / 7.2e-2 -- the
                          # bottleneck and first up block, whose BN normalises over 2 x (2 x 3) ... 2 x (16 x 20) values)


@pytest.mark.parametrize("shape", [(2, 64, 96), (2, 128, 160)])          # (a development mode, bench.py --config 5; the second shape has the >= 2048 row chunks at level 0 that select the n-split weight-gradient kernel's bf16 branch, which bench.py --config 5 runs)
def test_bf16_operand_mode_on_pattern(shape):
    """ENDO_OPT_MFMA_BF16 = 1 (the mixed-precision mode behind bench.py --config 5): the dense layers' forward, data-gradient and
    weight-gradient kernels round their MFMA operands to bf16 and accumulate in fp32; tensors in memory, BN statistics, the
    BN / ReLU / pooling arithmetic, reductions and the optimizer stay fp32.  It is a different function from the fp32 path, so it
    has its own stated tolerance: depth and all 210 gradients against the fp64 oracle evaluated on the activation pattern this
    very pass took (the comparison that is meaningful for a piecewise-linear network, see test_network_backward).  The second
    shape is large enough for the n-split weight gradient and the fused data-gradient kernels of the benchmark."""
    n, h, w = shape
    with kernel_options({OPT_MFMA_BF16: 1}):
        state, model = make_model(65)
        rng = np.random.default_rng(17)
        x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
        cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
        model.train()
        y = model(x.to(dev()))
        (pattern,) = pattern_of(y, model, n, h, w)
        (y * cot.to(dev())).sum().backward()
        torch.cuda.synchronize()
    params = dict(model.named_parameters())
    g64p = reference_grads(state, x, cot, torch.float64, pattern)
    y64 = onet.forward(state_as(state, torch.float64), x.double(), training=True, pattern=pattern)
    assert_close(y, y64, BF16_FWD_TOL, "depth, bf16 operands")
    report = assert_grads_on_pattern(params, g64p, None, BF16_GRAD_TOL, "network backward %s, bf16 operands" % (shape,))
    errors = sorted(r[0] for r in report)
    print("bf16 operands %s: depth err %.2e, gradient errors median %.2e, max %.2e" % (
        shape, float((y.detach().double().cpu() - y64).abs().max() / y64.abs().max()), errors[len(errors) // 2], errors[-1]))
    assert errors[len(errors) // 2] <= 1.5e-2
    assert errors[-1] > 1e-4, "the bf16 kernels did not run"


def test_bf16_operand_training_iterations():
    """A few fused training iterations in bf16-operand mode: first-iteration loss within 3 % of the fp32 path on the same batch
    (measured 1.5 %: 57 convolutions deep, every operand rounded to 8 bits), every iteration finite and not skipped."""
    n, h, w = 2, 128, 160
    batch = to_dev(synthetic.make_batch(n, h, w, seed=92, sparse_points=800))
    losses = {}
    for mode in (0, 1):
        with kernel_options({OPT_MFMA_BF16: mode}):
            _, model = make_model(66, positive_depth=True)
            model.train()
            step = ea.train_step.TrainingStep(model, ea.optim.FusedClipSGD(model, lr=1.0e-4), h, w)
            outs = [step(batch) for _ in range(4)]
            torch.cuda.synchronize()
        assert all(not o["skipped"] and np.isfinite(o["loss"]) for o in outs)
        losses[mode] = [o["loss"] for o in outs]
    assert abs(losses[1][0] - losses[0][0]) <= 3e-2 * abs(losses[0][0]), losses
    assert abs(losses[1][0] - losses[0][0]) > 0.0


def reference_pattern(state64, x64):
    """The fp64 oracle's own ReLU pattern (for counting how many bits a HIP pass flips)."""
    import torch.nn.functional as F
    own = {}
    plain = onet._bn_relu

    def recording(state, prefix, xx, training, pattern=None, quant=None):
        yy = F.batch_norm(xx, state[prefix + ".running_mean"].clone(), state[prefix + ".running_var"].clone(),
                          state[prefix + ".weight"], state[prefix + ".bias"], training, onet.BN_MOMENTUM, onet.BN_EPS)
        own["relu::" + prefix] = yy > 0
        return plain(state, prefix, xx, training, pattern, quant)
    onet._bn_relu = recording
    try:
        with torch.no_grad():
            onet.forward(state64, x64, training=True)
    finally:
        onet._bn_relu = plain
    return own


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 32, 64)])          # (an odd group size on the smallest grid the five poolings allow; (3, 64, 64) until round 5, 12 s more for the same launches; (2, 128, 160) went in round 4: the grouped pass at that size is covered on the pattern by test_network_backward_kernel_forms, at 2 x 8 x 256 x 320 by test_full_size_pair_backward_on_pattern)
def test_forward_pair_is_two_calls(shape):
    """forward_pair(x1, x2) -- both frames of a training pair as one grouped batch, every launch covering both, each
    frame with its own BatchNorm batch statistics -- against the oracle's two sequential calls (reference
    train.py:276-277): outputs, running statistics after both updates (order: x1 then x2), summed parameter gradients;
    and against the library's own two separate calls."""
    n, h, w = shape
    state, model = make_model(61)
    rng = np.random.default_rng(12)
    x1 = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    x2 = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    cot1 = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    cot2 = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))
    names = onet.trainable_names()
    # private copies first: state_as(.., float32) aliases its argument and a training-mode oracle forward updates the
    # running statistics in place
    st32 = {k: v.clone() for k, v in state.items()}
    st64 = state_as(state, torch.float64)
    y32 = [onet.forward(st32, x1, training=True), onet.forward(st32, x2, training=True)]
    y64 = [onet.forward(st64, x1.double(), training=True), onet.forward(st64, x2.double(), training=True)]

    _, twin = make_model(61)                   # same weights: the two-call path of the library itself
    twin.train()
    with torch.no_grad():
        t1, t2 = twin(x1.to(dev())), twin(x2.to(dev()))

    model.train()
    y1, y2 = model.forward_pair(x1.to(dev()), x2.to(dev()))
    assert y1.shape == (n, 1, h, w) and y2.shape == (n, 1, h, w)
    pat1, pat2 = pattern_of(y1, model, n, h, w, groups=2)          # each frame's own activation pattern (its own BN statistics)
    ((y1 * cot1.to(dev())).sum() + (y2 * cot2.to(dev())).sum()).backward()
    noise_aware(y1, y32[0], y64[0], "pair output 1")
    noise_aware(y2, y32[1], y64[1], "pair output 2")
    assert_close(y1, t1, 2e-5, "pair output 1 vs separate call")
    assert_close(y2, t2, 2e-5, "pair output 2 vs separate call")
    sd, sd_twin = model.state_dict(), twin.state_dict()
    for name in ("denseBlocksDown.0.layers.0.norm", "transDownBlocks.2.norm", "bottleneck.bottleneck.layers.3.norm",
                 "denseBlocksUp.4.layers.3.norm"):
        for stat in (".running_mean", ".running_var"):
            noise_aware(sd[name + stat], st32[name + stat], st64[name + stat], name + stat)
            assert_close(sd[name + stat], sd_twin[name + stat], 1e-5, name + stat + " vs separate calls")
        assert int(sd[name + ".num_batches_tracked"]) == 2
    params = dict(model.named_parameters())
    # summed parameter gradients of the two frames against the fp64 oracle on the two patterns the grouped pass took
    g64a, g64b = reference_grads(state, x1, cot1, torch.float64, pat1), reference_grads(state, x2, cot2, torch.float64, pat2)
    g32a, g32b = reference_grads(state, x1, cot1, torch.float32, pat1), reference_grads(state, x2, cot2, torch.float32, pat2)
    g64 = {nm: g64a[nm] + g64b[nm] for nm in names}
    g32 = {nm: g32a[nm] + g32b[nm] for nm in names}
    assert_grads_on_pattern(params, g64, g32, GRAD_TOL, "pair backward %s" % (shape,))
    model.eval()
    twin.eval()
    with torch.no_grad():
        e1, e2 = model.forward_pair(x1.to(dev()), x2.to(dev()))
        assert_close(e1, model(x1.to(dev())), 1e-6, "eval pair 1")
        assert_close(e2, model(x2.to(dev())), 1e-6, "eval pair 2")


@pytest.mark.parametrize("shape", [(2, 64, 64), (2, 128, 160)])
def test_network_backward_last_block_exact(shape):
    """The layers that are differentiated FIRST (final conv, last up block, its transition-up) see no
    accumulated mask-flip noise, so the kernels behind them -- dgrad with fused BN/ReLU backward,
    wgrad, the deferred-mean fold, the sum-pool dgrad -- must agree with the fp64 oracle to fp32
    rounding.  A smooth positive cotangent and depth away from zero keep the sums well conditioned;
    the three sizes put the small-tile, mid-tile and big-tile / persistent-dgrad variants on level 0."""
    n, h, w = shape
    state, model = make_model(56, positive_depth=True)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
    cot = 0.5 + synthetic.smooth_depth(n, h, w, seed=9)
    g64 = reference_grads(state, x, cot, torch.float64)
    model.train()
    y = model(x.to(dev()))
    (y * cot.to(dev())).sum().backward()
    params = dict(model.named_parameters())
    checked = 0
    for nm in onet.trainable_names():
        if nm.startswith(("finalConv", "denseBlocksUp.4.", "transUpBlocks.4.")):
            scale = grad_scale(g64, nm)
            err = float((params[nm].grad.detach().double().cpu() - g64[nm]).abs().max()) / scale
            # 1e-4 (north_star); a mask flip inside the block itself moves a sum by ~1/(N*H*W)
            assert err <= max(1e-4, 4.0 / (n * h * w)), "%s: rel err %.3e" % (nm, err)
            checked += 1
    assert checked == 20


def test_network_golden(golden):
    """HIP network against the fixture produced by the REFERENCE FCDenseNet57."""
    g = golden("network_2x64x96.npz")
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    state = onet.perturb_affine(onet.synthetic_state(seed), seed + 1)
    model = ea.FCDenseNet57(1)
    model.load_state_dict(state)
    model = model.to(dev()).train()
    rng = np.random.default_rng(seed + 2)
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)).to(dev())
    cot = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)).to(dev())
    y = model(x)
    assert_close(y, torch.from_numpy(g["output"]), 1e-4, "output vs reference fixture")
    (y * cot).sum().backward()
    params = dict(model.named_parameters())
    names = [str(s) for s in g["grad_names"]]
    assert [nm for nm, _ in model.named_parameters()] == names
    norms = np.array([float(params[nm].grad.double().norm()) for nm in names])
    # the fixture is an fp32 run: its early-layer gradients carry ~2e-3 of fp32 noise themselves
    # (see noise_aware); two fp32 evaluations can therefore differ by a few 1e-3
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-2, atol=1e-4 * float(g["grad_norms"].max()))
    for key in g.files:
        if key.startswith("grad::") and not (key.endswith(".bias") and ("layers" in key or "transDown" in key)):
            assert_close(params[key[6:]].grad, torch.from_numpy(g[key]), 1e-2, key)
        if key.startswith("buf::"):
            assert_close(model.state_dict()[key[5:]], torch.from_numpy(g[key]), 1e-4, key)


def test_optimizer_step():
    model = ea.FCDenseNet57(1).to(dev())
    ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
    opt = ea.optim.FusedClipSGD(model, lr=0.01, momentum=0.9, max_norm=10.0)
    rng = np.random.default_rng(7)
    p_ref = [p.detach().cpu().clone() for p in model.parameters()]
    bufs = [None] * len(p_ref)
    for step, scale in enumerate((5.0, 0.001, 1.0)):       # clipped, not clipped, in between
        flat = model.flat_gradients()
        gnp = (rng.standard_normal(flat.numel()) * scale).astype(np.float32)
        flat.copy_(torch.from_numpy(gnp))
        g_ref = []
        off = 0
        for p in p_ref:
            g_ref.append(torch.from_numpy(gnp[off:off + p.numel()].copy()).view(p.shape))
            off += p.numel()
        norm = opt.step()
        want = osch.clip_and_sgd(p_ref, g_ref, bufs, 0.01)
        assert_close(norm, want, 1e-5, "gradient norm step %d" % step)
        got = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        assert_close(got, torch.cat([p.reshape(-1) for p in p_ref]), 1e-5, "parameters after step %d" % step)


def test_train_step_vs_oracle():
    n, h, w = 2, 64, 96
    state, model = make_model(53, positive_depth=True)
    model.train()
    opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
    step = ea.train_step.TrainingStep(model, opt, h, w, sfl_weight=20.0, dcl_weight=0.1)
    momentum = {}
    for it in range(2):
        batch = synthetic.make_batch(n, h, w, seed=60 + it, sparse_points=500)
        lr = osch.cyclic_lr(it, 1.0e-4, 1.0e-3, 4)
        out = step(to_dev(batch), lr=lr)
        ref = ostep.train_iteration(state, momentum, batch, lr)
        assert not out["skipped"] and not ref["skipped"]
        # iteration 0 starts from identical parameters: 1e-4 (north_star).  Iteration 1 sees the
        # parameters each side updated with its own fp32 gradient (noise ~2e-3, see noise_aware):
        # loss moves by lr*|g|^2 ~ 5e-2 per step, so the two fp32 trajectories separate by ~2e-4.
        tol = 1e-4 if it == 0 else 1e-3
        assert abs(out["loss"] - float(ref["loss"])) <= tol * abs(float(ref["loss"])), (it, out["loss"], float(ref["loss"]))
        assert_close(out["dcl"], ref["dcl"], tol, "dcl")
        assert_close(out["sfl"], ref["sfl"], tol, "sfl")
        assert_close(out["grad_norm"], ref["grad_norm"], 5e-3, "grad norm")
        got = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        want = torch.cat([state[nm].reshape(-1) for nm in onet.trainable_names()])
        assert_close(got, want, 1e-4, "parameters after iteration %d" % it)


def test_train_step_golden(golden):
    g = golden("train_step_2x64x96.npz")
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(seed), seed + 1))
    model = ea.FCDenseNet57(1)
    model.load_state_dict(state)
    model = model.to(dev()).train()
    opt = ea.optim.FusedClipSGD(model, lr=float(g["max_lr"]))
    sched = ea.scheduler.CyclicLR(opt, base_lr=float(g["base_lr"]), max_lr=float(g["max_lr"]), step_size=int(g["step_size"]))
    step = ea.train_step.TrainingStep(model, opt, h, w)
    for it in range(2):
        batch = synthetic.make_batch(n, h, w, seed=seed + 10 + it, sparse_points=min(500, h * w // 6))
        sched.batch_step(batch_iteration=it)
        out = step(to_dev(batch))
        tag = "step%d_" % it
        tol = 1e-4 if it == 0 else 1e-3          # see test_train_step_vs_oracle
        assert abs(out["loss"] - float(g[tag + "loss"])) <= tol * abs(float(g[tag + "loss"]))
        assert_close(out["grad_norm"], torch.from_numpy(g[tag + "grad_norm"]), 5e-3, "grad norm")
        norms = np.array([float(p.double().norm()) for p in model.parameters()])
        np.testing.assert_allclose(norms, g[tag + "param_norms"], rtol=1e-4, atol=1e-5)


def test_level1_dropin_stock_optimizer(golden):
    """INTEGRATION.md level 1: the new modules under the reference's OWN optimizer calls -- torch.optim.SGD(momentum 0.9),
    optimizer.zero_grad() (set_to_none, which detaches the .grad views), loss.backward(), clip_grad_norm_(parameters, 10),
    optimizer.step() (reference train.py:202, 323-328) -- against the two reference iterations of train_step_2x64x96.npz,
    and against FusedClipSGD on a twin model (the parameters of the two optimizers must agree after each step)."""
    g = golden("train_step_2x64x96.npz")
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(seed), seed + 1))

    def build():
        m = ea.FCDenseNet57(1)
        m.load_state_dict(state)
        return m.to(dev()).train()
    model, twin = build(), build()
    stock = torch.optim.SGD(model.parameters(), lr=float(g["max_lr"]), momentum=0.9)
    sched = ea.scheduler.CyclicLR(stock, base_lr=float(g["base_lr"]), max_lr=float(g["max_lr"]), step_size=int(g["step_size"]))
    fused = ea.optim.FusedClipSGD(twin, lr=float(g["max_lr"]))
    fused_sched = ea.scheduler.CyclicLR(fused, base_lr=float(g["base_lr"]), max_lr=float(g["max_lr"]), step_size=int(g["step_size"]))
    step = ea.train_step.TrainingStep(model, fused, h, w)          # only its losses(): the optimizer calls below are the reference's
    twin_step = ea.train_step.TrainingStep(twin, fused, h, w)
    for it in range(2):
        batch = to_dev(synthetic.make_batch(n, h, w, seed=seed + 10 + it, sparse_points=min(500, h * w // 6)))
        sched.batch_step(batch_iteration=it)
        fused_sched.batch_step(batch_iteration=it)
        loss, _, _, _ = step.losses(batch)
        stock.zero_grad()                                           # torch >= 2.0: .grad = None for every parameter
        loss.backward()
        assert all(p.grad is not None for p in model.parameters())
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        stock.step()
        out = twin_step(batch)
        tag = "step%d_" % it
        tol = 1e-4 if it == 0 else 1e-3          # see test_train_step_vs_oracle
        assert abs(float(loss) - float(g[tag + "loss"])) <= tol * abs(float(g[tag + "loss"]))
        assert_close(gnorm, torch.from_numpy(g[tag + "grad_norm"]), 5e-3, "grad norm (stock clip_grad_norm_)")
        assert_close(gnorm, out["grad_norm"], 5e-5, "grad norm, stock vs fused")          # two backward passes, atomically summed statistics: 1.3e-5 seen
        norms = np.array([float(p.detach().double().norm()) for p in model.parameters()])
        np.testing.assert_allclose(norms, g[tag + "param_norms"], rtol=1e-4, atol=1e-5)
        assert_close(model.flat_parameters(), twin.flat_parameters(), 1e-6, "parameters after iteration %d, stock vs fused optimizer" % it)
        assert model._views_intact()


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 256, 320), (3, 32, 64)])
def test_fused_loss_head_matches_modules(shape):
    """endo_loss_head (the whole loss head and its backward as one library call, what TrainingStep runs by default) against the
    same step assembled from the drop-in modules and autograd (fused_head=False): the three loss values, d loss / d prediction
    for both frames, and the parameters after one full iteration.  Same kernels underneath, so the bounds are summation-order
    tight; the oracle comparisons of the step (test_train_step_vs_oracle, the goldens) run through the fused head."""
    n, h, w = shape
    batch = to_dev(synthetic.make_batch(n, h, w, seed=90, sparse_points=min(500, h * w // 6)))
    _, fused_model = make_model(63, positive_depth=True)
    _, module_model = make_model(63, positive_depth=True)
    fused_model.train()
    module_model.train()
    fused = ea.train_step.TrainingStep(fused_model, ea.optim.FusedClipSGD(fused_model, lr=1.0e-3), h, w)
    modular = ea.train_step.TrainingStep(module_model, ea.optim.FusedClipSGD(module_model, lr=1.0e-3), h, w, fused_head=False)
    assert fused.fused_head and not modular.fused_head
    # the head alone, on the same predictions
    losses_t, _, _, pred, grad_pred = fused._fused_iteration(batch)
    b = batch["boundaries"]
    p1 = pred[:n].detach().clone().requires_grad_(True)
    p2 = pred[n:].detach().clone().requires_grad_(True)
    s1, _ = modular.depth_scaling_layer([p1, batch["sparse_depths_1"], batch["sparse_depth_masks_1"]])
    s2, _ = modular.depth_scaling_layer([p2, batch["sparse_depths_2"], batch["sparse_depth_masks_2"]])
    mm = ea.train_step.mask_mul
    f1 = mm(modular.flow_from_depth_layer([s1, b, batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]]), b)
    f2 = mm(modular.flow_from_depth_layer([s2, b, batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]]), b)
    sfl = 20.0 * 0.5 * (modular.sparse_flow_loss_function([mm(batch["sparse_flows_1"], b), f1, mm(batch["sparse_flow_masks_1"], b)]) +
                        modular.sparse_flow_loss_function([mm(batch["sparse_flows_2"], b), f2, mm(batch["sparse_flow_masks_2"], b)]))
    w21, i1 = modular.depth_warping_layer([s1, s2, b, batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
    w12, i2 = modular.depth_warping_layer([s2, s1, b, batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
    dcl = 0.1 * 0.5 * (modular.depth_consistency_loss_function([s1, w21, i1, batch["intrinsics"]]) +
                       modular.depth_consistency_loss_function([s2, w12, i2, batch["intrinsics"]]))
    total = dcl + sfl
    g1, g2 = torch.autograd.grad(total, [p1, p2])
    assert_close(losses_t[0], total, 1e-6, "total loss, fused head vs modules")
    assert_close(losses_t[1], dcl, 1e-6, "depth consistency loss")
    assert_close(losses_t[2], sfl, 1e-6, "sparse flow loss")
    assert_close(grad_pred[:n], g1, 2e-6, "d loss / d prediction 1")
    assert_close(grad_pred[n:], g2, 2e-6, "d loss / d prediction 2")
    # a whole iteration each way (fresh forward passes)
    fused.optimizer.zero_grad()
    out_f = fused(batch)
    out_m = modular(batch)
    assert not out_f["skipped"] and not out_m["skipped"]
    assert abs(out_f["loss"] - out_m["loss"]) <= 1e-6 * abs(out_m["loss"])
    assert_close(out_f["grad_norm"], out_m["grad_norm"], 5e-5, "gradient norm")          # two backward passes (atomically summed statistics)
    assert_close(fused_model.flat_parameters(), module_model.flat_parameters(), 1e-6, "parameters after one iteration")


def test_nonfinite_guard():
    n, h, w = 1, 32, 32
    _, model = make_model(54)
    model.train()
    opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
    step = ea.train_step.TrainingStep(model, opt, h, w)
    batch = to_dev(synthetic.make_batch(n, h, w, seed=70, sparse_points=100))
    batch["sparse_flows_1"][0, 0, 5, 5] = float("nan")
    batch["sparse_flow_masks_1"][0, 0, 5, 5] = 1.0
    batch["boundaries"][0, 0, 5, 5] = 1.0
    before = model.flat_parameters().clone()
    out = step(batch, lr=1.0e-3)
    assert out["skipped"]
    assert torch.equal(before, model.flat_parameters())


@pytest.mark.parametrize("mode", ["fp32", "fp32-winograd", "bf16", "fp16"])
def test_clean_step_after_a_skipped_step(mode):
    """The guard is decided on the device: a step whose loss is NaN still runs the whole backward pass -- on NaN
    gradients -- before the optimizer kernel skips the update, so afterwards every gradient plane, partial-sum scratch and BN table of
    the workspace holds NaN.  The NEXT step must not see any of it (a kernel that multiplies a stale plane by a zero weight, or
    accumulates into a buffer it assumes clean, would carry the NaN on).  A model that ran [NaN step, clean step] must therefore
    end where a twin that ran only the clean step ends: same loss, gradient norm to the fp32 atomic-order bound, same parameters,
    same BN running statistics of the clean step, everything finite -- fp32 (default kernels and the Winograd forms forced on at
    this size) and both 16-bit-storage modes."""
    n, h, w = 2, 64, 96
    kw = {"bf16": {"bf16_storage": True}, "fp16": {"fp16_storage": True}}.get(mode, {})
    opts = {OPT_WINO_MIN_TILES: 1} if mode == "fp32-winograd" else {}
    bad = synthetic.make_batch(n, h, w, seed=72, sparse_points=300)
    bad["sparse_flows_1"][0, 0, 9, 11] = float("nan")
    bad["sparse_flow_masks_1"][0, 0, 9, 11] = 1.0
    bad["boundaries"][0, 0, 9, 11] = 1.0
    good = synthetic.make_batch(n, h, w, seed=73, sparse_points=300)
    results = []
    with kernel_options(opts):
        for with_nan_step in (True, False):
            _, model = make_model(58, positive_depth=True)
            model.train()
            opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
            step = ea.train_step.TrainingStep(model, opt, h, w, **kw)
            running0 = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
            if with_nan_step:
                before = model.flat_parameters().clone()
                out = step(to_dev(bad), lr=1.0e-3)
                assert out["skipped"] and torch.equal(before, model.flat_parameters())
                # the skipped step's forward did update the running statistics (the reference's two forward calls do too, train.py:276-277);
                # put them back so that both models enter the clean step in the same state
                model.load_state_dict(running0, strict=False)
            out = step(to_dev(good), lr=1.0e-3)
            torch.cuda.synchronize()
            assert not out["skipped"]
            results.append((out["loss"], float(out["grad_norm"]), model.flat_parameters().clone(), model.flat_gradients().clone(),
                            {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    (l_a, g_a, p_a, gr_a, r_a), (l_b, g_b, p_b, gr_b, r_b) = results
    assert np.isfinite(l_a) and np.isfinite(g_a) and torch.isfinite(p_a).all() and torch.isfinite(gr_a).all()
    tol = 5e-5 if mode.startswith("fp32") else 2e-3          # two backward passes: atomically summed statistics (fp32); the 16-bit modes round partial sums in another order
    assert abs(l_a - l_b) <= 1e-6 * abs(l_b), (l_a, l_b)
    assert abs(g_a - g_b) <= tol * g_b, (g_a, g_b)
    assert_close(gr_a, gr_b, 20 * tol, "gradients of the clean step, with and without a skipped step before it")
    assert_close(p_a, p_b, 1e-6 if mode.startswith("fp32") else 1e-5, "parameters after the clean step")
    for k in r_b:
        assert torch.isfinite(r_a[k]).all() and rel_err(r_a[k], r_b[k]) <= 1e-6, k


def test_checkpoint_resume_matches_reference(golden):
    """SURVEY 8(f2) on the device: the checkpoint the REFERENCE wrote after two iterations (its DataParallel 'module.' keys, its
    torch.optim.SGD state; tests/golden/checkpoint_2x64x96.npz, make_golden.py checkpoint_case) is loaded through
    utils.load_model_state + FusedClipSGD.load_state_dict, and iteration 3 -- which the reference ran from the same file -- must end
    where the reference's ends: loss 1e-4, gradient norm 5e-3 (the bounds of test_train_step_golden), every parameter tensor's norm
    within 1e-4 and, the sharp part, every tensor's UPDATE (lr x (0.9 x loaded momentum + clipped gradient)) within 2 % of the
    reference's in norm.  The same iteration from the same weights WITHOUT the optimizer state must miss that by a wide margin:
    the momentum really transferred."""
    from conftest import checkpoint_from_fixture
    g = golden("checkpoint_2x64x96.npz")
    blob = checkpoint_from_fixture(g)
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    batch = to_dev(synthetic.make_batch(n, h, w, seed=seed + 12, sparse_points=min(500, h * w // 6)))
    updates = {}
    for with_momentum in (True, False):
        model = ea.FCDenseNet57(1)
        res = ea.utils.load_model_state(model, blob["model"])
        assert not res.missing_keys and not res.unexpected_keys
        model = model.to(dev()).train()
        opt = ea.optim.FusedClipSGD(model, lr=float(g["max_lr"]))
        if with_momentum:
            opt.load_state_dict(blob["optimizer"])
        sched = ea.scheduler.CyclicLR(opt, base_lr=float(g["base_lr"]), max_lr=float(g["max_lr"]), step_size=int(g["step_size"]))
        step = ea.train_step.TrainingStep(model, opt, h, w)
        before = [p.detach().double().clone() for p in model.parameters()]
        sched.batch_step(batch_iteration=int(blob["step"]))
        assert abs(opt.param_groups[0]["lr"] - float(g["step2_lr"])) <= 1e-12
        out = step(batch)
        torch.cuda.synchronize()
        assert not out["skipped"]
        assert abs(out["loss"] - float(g["step2_loss"])) <= 1e-4 * abs(float(g["step2_loss"])), (out["loss"], float(g["step2_loss"]))
        assert_close(out["grad_norm"], torch.from_numpy(g["step2_grad_norm"]), 5e-3, "grad norm of the resumed iteration")
        updates[with_momentum] = np.array([float((p.detach().double() - q).norm()) for p, q in zip(model.parameters(), before)])
        if with_momentum:
            norms = np.array([float(p.detach().double().norm()) for p in model.parameters()])
            np.testing.assert_allclose(norms, g["step2_param_norms"], rtol=1e-4, atol=1e-5)
            sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
            assert np.abs(sums - g["step2_param_sums"]).max() <= 2e-4
            sd = opt.state_dict()          # and the state goes back out in torch.optim.SGD's layout, on the device
            assert sorted(sd["state"]) == list(range(210)) and all(v["momentum_buffer"].is_cuda for v in sd["state"].values())
    want = g["step2_update_norms"]
    big = want > 1e-3 * want.max()          # conv biases in front of a training-mode BN have a true gradient of zero: their "update" is rounding noise
    err_with = np.abs(updates[True][big] - want[big]) / want[big]
    err_without = np.abs(updates[False][big] - want[big]) / want[big]
    print("update-norm error per tensor: with the loaded momentum median %.2e max %.2e; without it median %.2e" % (
        np.median(err_with), err_with.max(), np.median(err_without)))
    assert err_with.max() <= 2e-2, "updates of the resumed iteration differ from the reference's: max %.3e" % err_with.max()
    assert np.median(err_without) >= 0.2, "the check is not sensitive to the momentum (median %.3e without it)" % np.median(err_without)


OPT_FINAL_VIRTUAL = 8
OPT_TD_PERSIST = 9


@pytest.mark.parametrize("extra", [{}, {4: 1}, {2: 0}, {2: 1}], ids=["fp32", "bf16-operands", "dgrad-vec-0", "dgrad-vec-1"])
@pytest.mark.parametrize("forced", [False, True], ids=["default-forms", "winograd-forms"])
def test_final_conv_fusions_are_transparent(forced, extra):
    """ENDO_OPT_FINAL_VIRTUAL (round 5, default on): the final 1x1 convolution's forward sum over the last dense layer's 180 input channels is
    formed by that layer's F(4x4,3x3) launch, its rank-one data gradient g * w[c] by the last up block's kernels (never written to the 192
    level-0 planes), and the first convolution's prep_dy is folded into its weight-gradient kernel.  With the option off the separate kernels
    run (final_fwd_kernel over 192 planes, final_bwd_data_kernel, prep_dy).  Same function: depth to 2e-6, all 210 gradients to 2e-5 of each
    tensor's maximum, one grouped pair pass each.  "winograd-forms" forces the Winograd kernels on at this size (the forms the benchmark-size
    launches take: the fused forward and the virtual BASE channels need them); "default-forms" leaves the new-map passes and prep_dy virtual
    and the base channels materialised."""
    n, h, w = 2, 64, 96
    rng = np.random.default_rng(23)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    results = []
    for virtual in (1, 0):
        with kernel_options({OPT_WINO_MIN_TILES: 1} if forced else {}):
            _, model = make_model(66)
        model.set_kernel_option(OPT_FINAL_VIRTUAL, virtual)
        for option_id, value in extra.items():          # ENDO_OPT_MFMA_BF16 = 1: the virtual old gradient goes through dgrad_block_kernel<..., BF = 1>;
            model.set_kernel_option(option_id, value)   # ENDO_OPT_DGRAD_VEC = 0 / 1: the per-tile new-map kernels with the virtual gradient
        model.train()
        y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
        ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
        torch.cuda.synchronize()
        results.append((y1.detach().clone(), y2.detach().clone(), {nm: p.grad.detach().clone() for nm, p in model.named_parameters()}))
    (a1, a2, ga), (b1, b2, gb) = results
    assert_close(a1, b1, 2e-6, "depth of frame 1, fused vs separate final convolution")
    assert_close(a2, b2, 2e-6, "depth of frame 2, fused vs separate final convolution")
    worst = 0.0
    for nm in ga:
        scale = max(float(gb[nm].abs().max()), 1e-3 * max(float(v.abs().max()) for v in gb.values()))
        worst = max(worst, float((ga[nm] - gb[nm]).abs().max()) / scale)
        assert float((ga[nm] - gb[nm]).abs().max()) <= 2e-5 * scale, "gradient of %s differs between the fused and the separate final-convolution kernels" % nm
    print("final-conv fusions on vs off (%s): worst gradient difference %.2e" % ("winograd forms" if forced else "default forms", worst))


@pytest.mark.parametrize("shape", [(2, 64, 96), (2, 96, 160), (3, 128, 160)], ids=lambda v: "x".join(str(i) for i in v))
def test_persistent_new_map_passes_match_the_per_tile_blocks(shape):
    """ENDO_OPT_DGRAD_VEC = 2 (round 5, default): the new-map passes of a dense block's backward run as persistent blocks that walk a run
    of 32 x 6 tiles (csrc/dgrad_newmap_kernels.h) instead of one block per tile (= 1, csrc/dgrad_block_kernels.h).  Same MFMA order per
    pixel, so the data gradients agree to the last bits; the BN-backward sums are added up in another order (fp32 per lane over the run,
    fp64 across blocks; sum of dz (x - mean) scaled by rstd once per tile instead of per term).  Bound: 5e-5 of each tensor's maximum (BatchNorm
    weight gradients are such sums with cancellation: 1.4e-5 measured), with a floor of 1e-2 of the largest gradient for the tensors whose true
    gradient is ZERO (the bias of a convolution that only BatchNorms read: what any kernel form returns there is the rounding residue of
    sums of O(1e3) terms -- 2e-5 absolute here -- and it changes with every summation order; two runs of ONE form differ by 5e-6 there).
    Shapes: 64 x 96 has a bottom rim at level 0 (64 = 10 * 6 + 4) and a right rim at level 1 (48 = 32 + 16); 96 x 160 tiles level 0 exactly and
    has right rims at levels 1 and 2 (80, 40); 3 x 128 x 160 has runs that cross the samples of a group.  The last up block's passes
    form the final convolution's virtual gradient (ENDO_OPT_FINAL_VIRTUAL) in both forms."""
    n, h, w = shape
    rng = np.random.default_rng(29)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    results = []
    for form in (2, 1):
        _, model = make_model(67)
        model.set_kernel_option(OPT_DGRAD_VEC, form)
        model.train()
        y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
        ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
        torch.cuda.synchronize()
        results.append({nm: p.grad.detach().clone() for nm, p in model.named_parameters()})
    ga, gb = results
    gmax = max(float(v.abs().max()) for v in gb.values())
    rows = []
    for nm in ga:
        assert torch.isfinite(ga[nm]).all(), nm
        scale = max(float(gb[nm].abs().max()), 1e-2 * gmax)
        rows.append((float((ga[nm] - gb[nm]).abs().max()) / scale, nm, float(gb[nm].abs().max())))
    rows.sort(reverse=True)
    worst = rows[0][0]
    assert worst <= 5e-5, "gradients differ between the persistent and the per-tile new-map passes: " + "; ".join("%s %.2e (max %.2e)" % (nm, d, mx) for d, nm, mx in rows[:6])
    print("persistent vs per-tile new-map passes %s: worst gradient difference %.2e of the tensor's maximum" % (shape, worst))


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 96, 160), (8, 128, 160)], ids=lambda v: "x".join(str(i) for i in v))
def test_persistent_base_pass_matches_the_per_tile_kernel(shape):
    """ENDO_OPT_WINO_DGRAD = 3 (round 6, default): the fused base-channel data gradient of a dense block runs as persistent blocks that walk
    a run of 32 x 8 tiles (csrc/dgrad_wino3p_kernels.h: dY maps by 16-byte LDS-DMA refilled layer by layer for the next tile, counted
    waits, per-tile flush of the BN-backward sums) instead of one block per tile (= 1, csrc/dgrad_wino3_kernels.h), and for the last up block
    it also forms the final convolution's weight gradient of the 144 base channels (final_bwd_weight_kernel reads the other 48).  Same
    arithmetic per pixel; sums are added up in another order.  Bounds as for the persistent new-map passes: 5e-5 of each tensor's maximum
    with a floor of 1e-2 of the largest gradient for the tensors whose true gradient is zero.  The Winograd forms are forced on
    (ENDO_OPT_WINO_MIN_TILES = 1) so that the kernel runs at these sizes: blocks with 48, 96 and 144 base channels take it (odd and even
    group counts), the 192-channel block keeps the per-tile kernel in both runs.  3 x 96 x 160: runs that cross the samples of a group;
    8 x 128 x 160: several tiles per block (320 tiles per group on 128 blocks)."""
    n, h, w = shape
    rng = np.random.default_rng(31)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    results = []
    for form in (3, 1):
        with kernel_options({OPT_WINO_MIN_TILES: 1}):
            _, model = make_model(68)
        model.set_kernel_option(OPT_WINO_DGRAD, form)
        model.train()
        y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
        ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
        torch.cuda.synchronize()
        results.append({nm: p.grad.detach().clone() for nm, p in model.named_parameters()})
    ga, gb = results
    gmax = max(float(v.abs().max()) for v in gb.values())
    rows = []
    for nm in ga:
        assert torch.isfinite(ga[nm]).all(), nm
        scale = max(float(gb[nm].abs().max()), 1e-2 * gmax)
        rows.append((float((ga[nm] - gb[nm]).abs().max()) / scale, nm, float(gb[nm].abs().max())))
    rows.sort(reverse=True)
    worst = rows[0][0]
    fin = [r for r in rows if r[1] == "finalConv.weight"]
    assert worst <= 5e-5, "gradients differ between the persistent and the per-tile base pass: " + "; ".join("%s %.2e (max %.2e)" % (nm, d, mx) for d, nm, mx in rows[:6])
    print("persistent vs per-tile base pass %s: worst gradient difference %.2e of the tensor's maximum; final convolution's weight %.2e" % (
        shape, worst, fin[0][0] if fin else float("nan")))


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 96, 160), (8, 128, 160)], ids=lambda v: "x".join(str(i) for i in v))
def test_persistent_transition_down_dgrad_matches_the_per_tile_kernel(shape):
    """ENDO_OPT_TD_PERSIST bit 0 (round 6, default on): the data gradient of the transition-down layers with 96 / 144 channels (levels 0 / 1) runs
    as persistent blocks (csrc/td_dgrad_kernels.h: weights LDS-resident, pooled gradient + argmax codes by 16-byte DMA through a swizzled
    source, x / old gradient requested ahead of a pass's MFMAs) where the level has whole 32 x 8 tiles, instead of one block per
    (tile, 32 output channels) (= 2, conv_dma_kernel<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN>).  Same products per pixel in the same k order;
    BN-backward sums in another order.  Bounds as for the other persistent forms.  64 x 96: level 0 only (level 1 is 32 x 48: a partial
    tile, per-tile kernel in both runs); 96 x 160 and 128 x 160: level 0 (96 channels, two tile buffers) and, at 128 x 160, level 1
    (144 channels, one buffer: 64 x 80 has partial tiles -- per-tile kernel)."""
    n, h, w = shape
    rng = np.random.default_rng(37)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    results = []
    for form in (3, 2):          # bit 1 (the forward) on in both
        _, model = make_model(69)
        model.set_kernel_option(OPT_TD_PERSIST, form)
        model.train()
        y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
        ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
        torch.cuda.synchronize()
        results.append({nm: p.grad.detach().clone() for nm, p in model.named_parameters()})
    ga, gb = results
    gmax = max(float(v.abs().max()) for v in gb.values())
    rows = []
    for nm in ga:
        assert torch.isfinite(ga[nm]).all(), nm
        scale = max(float(gb[nm].abs().max()), 1e-2 * gmax)
        rows.append((float((ga[nm] - gb[nm]).abs().max()) / scale, nm, float(gb[nm].abs().max())))
    rows.sort(reverse=True)
    worst = rows[0][0]
    assert worst <= 5e-5, "gradients differ between the persistent and the per-tile transition-down data gradient: " + "; ".join("%s %.2e (max %.2e)" % (nm, d, mx) for d, nm, mx in rows[:6])
    print("persistent vs per-tile transition-down data gradient %s: worst gradient difference %.2e of the tensor's maximum" % (shape, worst))


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 96, 160), (8, 128, 160)], ids=lambda v: "x".join(str(i) for i in v))
def test_persistent_transition_down_forward_matches_the_per_tile_kernel(shape):
    """ENDO_OPT_TD_PERSIST bit 1 (round 6, default on): the training-mode forward of the transition-down layers with 96 / 144 channels on whole
    32 x 8 tiles runs as persistent blocks (csrc/td_fwd_kernels.h: transposed weights LDS-resident, all output channels per tile, the input
    planes once through three LDS stages of 16-byte DMA with a swizzled source) instead of one block per (tile, 48 output channels)
    (conv_dma_kernel<1, 8, 3, IN_BNRELU, EPI_FWD_POOL>).  The 1x1 products are summed in the same channel order (k = 0 .. C - 1 through the
    same MFMA), so the pooled activations agree to the last bits and the depth to 1e-6 of its maximum; running statistics and the saved
    (mean, rstd) -- written by the kernel's first blocks -- agree to 1e-6."""
    n, h, w = shape
    rng = np.random.default_rng(41)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    outs = []
    for form in (3, 1):
        _, model = make_model(70)
        model.set_kernel_option(OPT_TD_PERSIST, form)
        model.train()
        y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
        torch.cuda.synchronize()
        outs.append((y1.detach().clone(), y2.detach().clone(), {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k}))
    (a1, a2, ra), (b1, b2, rb) = outs
    assert_close(a1, b1, 1e-6, "depth of frame 1, persistent vs per-tile transition-down forward")
    assert_close(a2, b2, 1e-6, "depth of frame 2, persistent vs per-tile transition-down forward")
    for k in ra:
        assert_close(ra[k], rb[k], 1e-6, "running statistic %s" % k)


@pytest.mark.parametrize("shape", [(2, 64, 96), (4, 128, 160)])
def test_wgrad_overlap_is_transparent(shape):
    """endo_net_bwd runs the weight gradients on a side stream, overlapped with the data-gradient chain.
    With the overlap switched off the same kernels run in line on one stream; the gradients after one backward pass, and
    the parameters after three training iterations, must agree to the noise of fp32 atomic accumulation order -- a race
    between the streams (a weight gradient reading a buffer the chain is rewriting) would show up as an O(1) difference
    in some tensor."""
    n, h, w = shape
    results = []
    if True:
        for overlap in (1, 0):
            _, model = make_model(57, positive_depth=True)
            model.set_kernel_option(OPT_WGRAD_OVERLAP, overlap)
            model.train()
            opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
            step = ea.train_step.TrainingStep(model, opt, h, w)
            batch = to_dev(synthetic.make_batch(n, h, w, seed=71))
            # one backward pass without the optimizer: raw gradients
            loss, _, _, _ = step.losses(batch)
            loss.backward()
            torch.cuda.synchronize()
            grads = model.flat_gradients().clone()
            for _ in range(3):
                step(batch, lr=1.0e-3)
            torch.cuda.synchronize()
            results.append((grads, model.flat_parameters().clone()))
    (g_on, p_on), (g_off, p_off) = results
    assert torch.isfinite(g_on).all() and torch.isfinite(p_on).all()
    assert_close(g_on, g_off, 2e-5, "gradients, overlap on vs off")
    assert_close(p_on, p_off, 2e-5, "parameters after 3 iterations, overlap on vs off")
    # per tensor, so that a small tensor cannot hide behind the largest one
    for (nm, prm), o in zip(model.named_parameters(), model._offsets):
        a, b = g_on[o:o + prm.numel()], g_off[o:o + prm.numel()]
        scale = max(float(b.abs().max()), 1e-3 * float(g_off.abs().max()))
        assert float((a - b).abs().max()) <= 1e-3 * scale, "gradient of %s differs between overlap on and off" % nm


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: 512 x 640 (network_downsampling 64), and the LDS source tiles of the warp kernels
# ---------------------------------------------------------------------------------------------
WARP_TILES = [(0, 0), (8, 32), (16, 32), (16, 64), (32, 32), (32, 64)]


@pytest.mark.parametrize("tile", WARP_TILES)
def test_depth_warping_tiles_512x640(tile):
    """Every LDS source-tile shape of the warp kernels (and the L2-gather kernels, 0 x 0) at the configs[3] frame size
    against the oracle evaluated in fp64: warped depth and both gradients 1e-4; the intersect mask away from the threshold.
    (fp64 because at 640 pixels the oracle's own fp32 sample coordinates -- K R^T K^-1 formed in fp32 -- are 1e-4 pixel
    off, which alone moves the fp32 oracle's warped depth 1.1e-4 from this kernel; the kernels form the camera maps in fp64.)"""
    n, h, w = 1, 512, 640
    batch, p1, p2, _ = geometry_inputs(n, h, w, 47)
    cot = torch.from_numpy(np.random.default_rng(3).standard_normal((n, 1, h, w)).astype(np.float32))
    args = [batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]]
    c1 = p1.double().requires_grad_(True)
    c2 = p2.double().requires_grad_(True)
    w_ref, overlap = ogeo.depth_warping_parts(c1, c2, *[a.double() for a in args])
    (w_ref * cot.double()).sum().backward()
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    warped, inter = ea.DepthWarpingLayer(epsilon=1.0e-8, tile=tile)([g1, g2] + [a.to(dev()) for a in args])
    (warped * cot.to(dev())).sum().backward()
    assert_close(warped, w_ref, 1e-4, "warped depth, tile %s" % (tile,))
    assert_close(g2.grad, c2.grad, 1e-4, "grad depth 2, tile %s" % (tile,))
    # the d1 gradient is the slope of a bilinear patch: piecewise constant in the sample position, so where fp32 and fp64 put
    # a sample on different sides of a pixel boundary (|position - integer| < ~1e-4 pixel at 640 columns) that one pixel's
    # gradient differs by O(1).  Everywhere else: 1e-4; such pixels: at most 0.1 % of the frame.
    scale = float(c1.grad.abs().max())
    err = (g1.grad.detach().double().cpu() - c1.grad).abs() / scale
    off = err > 1e-4
    assert float(off.double().mean()) <= 1e-3, "grad depth 1, tile %s: %.3e of the pixels beyond 1e-4" % (tile, float(off.double().mean()))
    assert float(torch.quantile(err.reshape(-1)[::3], 0.99)) <= 1e-5, "grad depth 1, tile %s: 99th percentile" % (tile,)
    clear = (overlap.detach() - 0.9).abs() > 1e-4
    assert torch.equal(inter.cpu()[clear], (overlap.detach() >= 0.9).float()[clear]), "intersect mask"


def test_warp_tiles_agree_and_fall_back():
    """The tile shape changes speed only: forward bit-identical to the gather kernels for every shape, d2 gradient equal up to
    the order of its atomic additions -- on ordinary motion, on a batch whose poses are scaled x10 (source boxes larger than
    the staging buffers: blocks fall back to gathers), with points behind the camera and samples that leave the frame
    (the edge cases of test_warp_edge_cases), and on a size that is not a multiple of any tile (37 x 53)."""
    for (n, h, w, scale) in ((2, 256, 320, 1.0), (2, 256, 320, 10.0), (2, 37, 53, 1.0), (2, 64, 96, 3.0)):
        batch, p1, p2, _ = geometry_inputs(n, h, w, 48)
        t = batch["translations_1_wrt_2"].clone() * scale
        t[0, 2, 0] = 2.0                     # z2 <= 0 for sample 0
        args = [batch["boundaries"].to(dev()), t.to(dev()), batch["rotations_1_wrt_2"].to(dev()), batch["intrinsics"].to(dev())]
        cot = torch.randn(n, 1, h, w, device=dev(), generator=torch.Generator(device=dev()).manual_seed(2))
        ref = None
        for tile in WARP_TILES:
            g1 = p1.to(dev()).requires_grad_(True)
            g2 = p2.to(dev()).requires_grad_(True)
            warped, inter = ea.DepthWarpingLayer(tile=tile)([g1, g2] + args)
            (warped * cot).sum().backward()
            if ref is None:
                ref = (warped.detach().clone(), inter.clone(), g1.grad.clone(), g2.grad.clone())
                assert torch.isfinite(ref[0]).all()
                continue
            what = "tile %s at %dx%d, motion x%g" % (tile, h, w, scale)
            assert torch.equal(warped.detach(), ref[0]), "warped differs from the gather kernel: " + what
            assert torch.equal(inter, ref[1]), "intersect differs: " + what
            assert torch.equal(g1.grad, ref[2]), "d1 gradient differs: " + what
            assert_close(g2.grad, ref[3], 1e-5, "d2 gradient, " + what)
    with pytest.raises(RuntimeError):
        ea.DepthWarpingLayer(tile=(24, 48))([g1, g2] + args)


def test_geometry_and_losses_512x640():
    """The whole geometry + loss chain of a training step at 1 x 512 x 640 (configs[3] frame size) against the oracle."""
    n, h, w = 1, 512, 640
    batch, p1, p2, _ = geometry_inputs(n, h, w, 49)
    p1, p2 = p1 + 2.0, p2 + 2.0
    c1 = p1.double().requires_grad_(True)          # fp64 oracle: see test_depth_warping_tiles_512x640
    c2 = p2.double().requires_grad_(True)
    l_ref, dcl_ref, sfl_ref, _ = ostep.losses_from_depths(c1, c2, {k: v.double() for k, v in batch.items()})
    l_ref.backward()
    dbatch = to_dev(batch)
    b = dbatch["boundaries"]
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    scaling, flow_layer, warp_layer = ea.DepthScalingLayer(), ea.FlowfromDepthLayer(), ea.DepthWarpingLayer()
    sfl_fn, dcl_fn = ea.SparseMaskedL1Loss(), ea.NormalizedDistanceLoss(h, w)
    s1, _ = scaling([g1, dbatch["sparse_depths_1"], dbatch["sparse_depth_masks_1"]])
    s2, _ = scaling([g2, dbatch["sparse_depths_2"], dbatch["sparse_depth_masks_2"]])
    f1 = flow_layer([s1, b, dbatch["translations_1_wrt_2"], dbatch["rotations_1_wrt_2"], dbatch["intrinsics"]]) * b
    f2 = flow_layer([s2, b, dbatch["translations_2_wrt_1"], dbatch["rotations_2_wrt_1"], dbatch["intrinsics"]]) * b
    sfl = 20.0 * 0.5 * (sfl_fn([dbatch["sparse_flows_1"] * b, f1, dbatch["sparse_flow_masks_1"] * b]) +
                        sfl_fn([dbatch["sparse_flows_2"] * b, f2, dbatch["sparse_flow_masks_2"] * b]))
    w21, i1 = warp_layer([s1, s2, b, dbatch["translations_1_wrt_2"], dbatch["rotations_1_wrt_2"], dbatch["intrinsics"]])
    w12, i2 = warp_layer([s2, s1, b, dbatch["translations_2_wrt_1"], dbatch["rotations_2_wrt_1"], dbatch["intrinsics"]])
    dcl = 0.1 * 0.5 * (dcl_fn([s1, w21, i1, dbatch["intrinsics"]]) + dcl_fn([s2, w12, i2, dbatch["intrinsics"]]))
    (dcl + sfl).backward()
    assert_close(sfl, sfl_ref, 1e-4, "sparse flow loss at 512x640")
    assert_close(dcl, dcl_ref, 1e-4, "depth consistency loss at 512x640")
    for got, want, what in ((g1.grad, c1.grad, "grad pred 1 at 512x640"), (g2.grad, c2.grad, "grad pred 2 at 512x640")):
        err = (got.detach().double().cpu() - want).abs() / float(want.abs().max())          # see test_depth_warping_tiles_512x640
        assert float((err > 1e-4).double().mean()) <= 1e-3, what
        assert float(torch.quantile(err.reshape(-1)[::3], 0.99)) <= 1e-5, what


def test_pair_backward_on_pattern_512x640():
    """configs[3] AT ITS OWN GRID: forward_pair at 2 x (4 x 512 x 640) -- 8 samples per launch, exactly the launches
    ``bench.py --config 3`` times (other grids select other kernel variants and workspace sizes: round 3's partial-buffer overrun
    was such a case) -- all 210 parameter gradients against the fp32 CPU oracle on the pass's own activation pattern (as
    test_full_size_pair_backward_on_pattern), 1e-4."""
    n, h, w = 4, 512, 640
    state, model = make_model(59)
    rng = np.random.default_rng(15)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    model.train()
    y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
    patterns = pattern_of(y1, model, n, h, w, groups=2)
    ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    names = onet.trainable_names()
    total = None
    for x, cot, pat, got in zip(xs, cots, patterns, (y1, y2)):
        st = {k: v.clone() for k, v in state.items()}
        for nm in names:
            st[nm].requires_grad_(True)
        y = onet.forward(st, x, training=True, pattern=pat)
        assert_close(got, y.detach(), 1e-4, "depth at 512x640 vs the oracle on the same pattern")
        grads = torch.autograd.grad((y * cot).sum(), [st[nm] for nm in names])
        total = list(grads) if total is None else [a + b for a, b in zip(total, grads)]
        del y, grads, st
    assert_grads_on_pattern(params, dict(zip(names, total)), None, 1e-4, "512x640 pair backward")


# ---------------------------------------------------------------------------------------------
# the benchmark configuration itself (BASELINE.json configs[1]: N = 8, 256 x 320, grouped pair forward) against the
# fixture the REFERENCE produced at that size
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("forward_form", ["default", "f23"])
def test_train_step_full_size_golden(golden, forward_form):
    """forward_form: "default" = the kernel forms bench.py runs (since round 5 the level-0 dense layers' forward in F(4x4, 3x3), ENDO_OPT_WINO_FWD = 5);
    "f23" = the same with F(2x2, 3x3) there (ENDO_OPT_WINO_FWD = 1), the form every bound below was set on in round 4.  The bounds are the
    SAME for both and unchanged from round 4.  (With round 4's textbook interpolation points the F(4x4, 3x3) forward sat 5e-6 from fp64 on the
    depth, flipped about five times as many ReLU / max-pool decisions and put 8 of the 185 gradient tensors further than 1e-2 from fp64 --
    over the bound of 5 -- which is why it was not the default then.  With the points 0, +-5/8, +-3/2, inf of round 5 the same run gives:
    depth 9.6e-7, 0 of 185 tensors beyond 1e-2, medians 8.7e-5 / 1.0e-3 against the reference's own 5.1e-5 / 6.2e-4; the F(2x2, 3x3) form:
    7.1e-7, 1 of 185, 6.1e-5 / 8.2e-4.)

    One training iteration at the size and through the code path bench.py times -- TrainingStep(pair_forward=True):
    16 samples per launch, the 32x16 / split-K / n-split / 8-wave fused-dgrad variants that only these grids select --
    against tests/golden/train_step_8x256x320.npz, which make_golden.py wrote by running the reference's own modules,
    torch.optim.SGD and clip_grad_norm_ on the same seeded batch (reference train.py:272-328).

    Forward (loss terms, depth, scaled depth, warped depth, BN running statistics): 1e-4 relative, BASELINE.json's bar
    (measured 1e-7 on the losses, 8e-7 on the depth).
    Gradients: a fixture cannot carry the activation pattern of a 2 x 8 x 256 x 320 pass (2 G mask bits), so here the HIP
    gradient and the reference's fp32 gradient are two evaluations on slightly different patterns, and the fixture's fp64
    evaluation (a third) is the yardstick: the reference's own fp32 gradients sit up to 2.6e-3 (elementwise) / 4.4e-3
    (projection) from fp64, the coarse levels worst, where one flipped ReLU bit is one of only 1280 pixels of a channel.
    Bounds: per tensor, norm / seeded random projection (sees every element) / the 23 tensors kept in full within 5e-2 of
    the tensor's scale from fp64 and at most 5 of the 185 tensors beyond 1e-2 (which tensor a flipped bit of a 16 x 20
    level lands in changes with every rounding difference: two builds of this library already move it); over all
    tensors, the median distance within 3x the reference's median.  The TIGHT
    statement about these kernel variants is test_full_size_pair_backward_on_pattern below (same grids, same pattern,
    1e-4 on every tensor)."""
    g = golden("train_step_8x256x320.npz")
    n, h, w, seed, sub = (int(g[k]) for k in ("n", "h", "w", "seed", "subsample"))
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(seed), seed + 1), bias=float(g["final_bias_shift"]))
    model = ea.FCDenseNet57(1)
    model.load_state_dict(state)
    model = model.to(dev()).train()
    if forward_form == "f23":
        model.set_kernel_option(OPT_WINO_FWD, 1)
    else:
        assert model.kernel_option(OPT_WINO_FWD) == 5
    opt = ea.optim.FusedClipSGD(model, lr=float(g["lr"]))
    step = ea.train_step.TrainingStep(model, opt, h, w, sfl_weight=float(g["sfl_weight"]), dcl_weight=float(g["dcl_weight"]),
                                      pair_forward=True)
    assert step.pair_forward
    batch = to_dev(synthetic.make_batch(n, h, w, seed=seed + 10, sparse_points=500))
    loss, dcl, sfl, ex = step.losses(batch)
    problems = []

    def check(ok, msg):
        if not ok:
            problems.append(msg)

    for key, val in (("loss", loss), ("dcl", dcl), ("sfl", sfl)):
        err = abs(float(val) - float(g[key])) / abs(float(g[key]))
        print("%-5s hip %.8f  reference %.8f  fp64 oracle %.8f  rel %.2e" % (key, float(val), float(g[key]), float(g["o64_" + key]), err))
        check(err <= 1e-4, "%s: rel err %.3e > 1e-4" % (key, err))
    for key in ("pred_1", "pred_2", "scaled_1", "warped_21"):
        full = ex[key].detach()
        got = full[:, :, ::sub, ::sub].cpu()
        err = rel_err(got, torch.from_numpy(g[key]))
        check(err <= 1e-4, "%s (every %dth pixel): %.3e > 1e-4" % (key, sub, err))
        s_err = abs(float(full.double().sum()) - float(g[key + "_sum"])) / float(g[key + "_abs"])
        check(s_err <= 1e-5, "%s plane sum: %.3e > 1e-5 of the absolute sum" % (key, s_err))
        print("%-10s max rel err %.2e  sum rel err %.2e" % (key, err, s_err))
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    names = [str(s) for s in g["grad_names"]]
    params = dict(model.named_parameters())
    assert [nm for nm, _ in model.named_parameters()] == names
    n32, n64, p32, p64 = g["grad_norms"], g["o64_grad_norms"], g["grad_probes"], g["o64_grad_probes"]
    top = float(n64.max())
    report = []
    for i, nm in enumerate(names):
        got = params[nm].grad.detach().double().cpu().reshape(-1)
        if n64[i] <= 1e-7 * top:
            # a conv bias in front of training-mode BNs only: the true gradient is exactly zero, fp32 leaves rounding noise
            # (the reference's own value is ~1e-5 of the weight gradient); judged on the scale of its weight (listed just before)
            check(float(got.norm()) <= 1e-4 * n64[i - 1], "grad %s: |g| = %.3e, expected ~0 (weight gradient norm %.3e)" % (nm, float(got.norm()), n64[i - 1]))
            continue
        scale = float(n64[i])
        e_norm = abs(float(got.norm()) - n64[i]) / scale
        r_norm = abs(n32[i] - n64[i]) / scale
        probe = np.random.default_rng(seed * 1000 + i).standard_normal(got.numel())
        e_probe = abs(float((got.numpy() * probe).sum()) - p64[i]) / scale
        r_probe = abs(p32[i] - p64[i]) / scale
        report.append((max(e_norm, e_probe), nm, e_norm, r_norm, e_probe, r_probe))
        # per-tensor bound 3e-2, at most 2 tensors beyond 1e-2, medians within 2x the reference's own (round 6: the bounds follow what is
        # measured -- 0 / 1 tensors beyond 1e-2 and medians 1.7x in the two forward forms; if one trips on another box, report the numbers)
        check(e_norm <= 3e-2, "grad norm %s: hip-vs-fp64 %.3e, reference-vs-fp64 %.3e" % (nm, e_norm, r_norm))
        check(e_probe <= 3e-2, "grad projection %s: hip-vs-fp64 %.3e, reference-vs-fp64 %.3e" % (nm, e_probe, r_probe))
    beyond = [r[1] for r in report if r[0] > 1e-2]
    print("forward form %s: %d of %d tensors further than 1e-2 from fp64" % (forward_form, len(beyond), len(report)))
    check(len(beyond) <= 2, "%d tensors further than 1e-2 from fp64: %s" % (len(beyond), beyond[:10]))
    med = [float(np.median([r[k] for r in report])) for k in (2, 3, 4, 5)]
    print("median over %d tensors: norm err hip %.2e / reference %.2e, projection err hip %.2e / reference %.2e" % (len(report), *med))
    check(med[0] <= max(2.0 * med[1], 1e-4), "median gradient-norm distance from fp64: hip %.3e, reference %.3e" % (med[0], med[1]))
    check(med[2] <= max(2.0 * med[3], 1e-4), "median gradient-projection distance from fp64: hip %.3e, reference %.3e" % (med[2], med[3]))
    report.sort(reverse=True)
    print("worst gradient tensors (name, norm err hip / ref32, projection err hip / ref32):")
    for row in report[:8]:
        print("   %-46s %.2e %.2e %.2e %.2e" % row[1:])
    for nm in (str(s) for s in g["keep"]):
        r32, r64 = torch.from_numpy(g["grad::" + nm]).double(), torch.from_numpy(g["o64_grad::" + nm]).double()
        scale = max(float(r64.abs().max()), 1e-30)
        e = float((params[nm].grad.detach().double().cpu() - r64).abs().max()) / scale
        r = float((r32 - r64).abs().max()) / scale
        print("   kept %-46s hip-vs-fp64 %.2e  reference-vs-fp64 %.2e" % (nm, e, r))
        if not nm.endswith("conv.bias") and not nm.endswith("convTrans.1.bias"):
            check(e <= 3e-2, "grad tensor %s: hip-vs-fp64 %.3e, reference-vs-fp64 %.3e" % (nm, e, r))
    norm = opt.step()
    e = abs(float(norm) - float(g["o64_grad_norm"])) / float(g["o64_grad_norm"])
    r = abs(float(g["grad_norm"]) - float(g["o64_grad_norm"])) / float(g["o64_grad_norm"])
    print("total gradient norm: hip %.6f reference %.6f fp64 %.6f" % (float(norm), float(g["grad_norm"]), float(g["o64_grad_norm"])))
    check(e <= max(4.0 * r, 1e-4), "total gradient norm: hip-vs-fp64 %.3e, reference-vs-fp64 %.3e" % (e, r))
    # parameters after clip_grad_norm_(10) + SGD: the update has norm lr * 10 = 1e-2 in total, so agreement of the
    # per-tensor norms and sums to 2e-5 absolute pins the update to ~1e-3 of itself
    norms = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
    check(np.abs(norms - g["param_norms"]).max() <= 2e-5, "parameter norms after the step: %.3e" % np.abs(norms - g["param_norms"]).max())
    check(np.abs(sums - g["param_sums"]).max() <= 2e-4, "parameter sums after the step: %.3e" % np.abs(sums - g["param_sums"]).max())
    sd = model.state_dict()
    for name in (str(s) for s in g["buffers"]):
        for stat in (".running_mean", ".running_var"):
            err = rel_err(sd[name + stat], torch.from_numpy(g["buf::" + name + stat]))
            check(err <= 1e-4, "%s: %.3e > 1e-4" % (name + stat, err))
        check(int(sd[name + ".num_batches_tracked"]) == 2, name + ".num_batches_tracked")
    assert not problems, "\n".join(problems)


def test_train_step_full_size_golden_fused_path(golden):
    """The code path bench.py TIMES, at the size it times: ``TrainingStep.__call__`` -> ``_fused_iteration`` (mask, grouped pair
    forward, ``endo_loss_head``) -> host guard -> ``_fused_backward`` -> all-reduce hook -> ``FusedClipSGD.step`` -- against the
    reference-generated fixture tests/golden/train_step_8x256x320.npz (reference train.py:272-328 run by make_golden.py with the
    reference's modules, torch.optim.SGD and clip_grad_norm_).  test_train_step_full_size_golden drives the same kernels through
    ``step.losses()`` + autograd + a separate ``opt.step()``; here nothing but ``step(batch)`` is called, so workspace sizing, the
    ``grad_pred`` hand-over and the zero-before-sync ordering of the fused path are what is checked: loss terms 1e-4, total
    gradient norm as in the module-path test, parameters after the update (per-tensor norms 2e-5 / sums 2e-4 absolute: the
    update has norm lr * 10 = 1e-2, so that is ~1e-3 of it), BN running statistics 1e-4, two statistic updates per BN layer --
    and the gradients left in the flat buffer against a twin model stepped through the module path (same kernels, same inputs:
    fp32 atomic-order noise only)."""
    g = golden("train_step_8x256x320.npz")
    n, h, w, seed = (int(g[k]) for k in ("n", "h", "w", "seed"))
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(seed), seed + 1), bias=float(g["final_bias_shift"]))
    models, steps, opts = [], [], []
    for _ in range(2):
        model = ea.FCDenseNet57(1)
        model.load_state_dict(state)
        model = model.to(dev()).train()
        opt = ea.optim.FusedClipSGD(model, lr=float(g["lr"]))
        steps.append(ea.train_step.TrainingStep(model, opt, h, w, sfl_weight=float(g["sfl_weight"]), dcl_weight=float(g["dcl_weight"]),
                                                pair_forward=True))
        models.append(model); opts.append(opt)
    assert steps[0].fused_head and steps[0].pair_forward
    batch = to_dev(synthetic.make_batch(n, h, w, seed=seed + 10, sparse_points=500))
    out = steps[0](batch)                                   # THE timed path
    torch.cuda.synchronize()
    assert not out["skipped"]
    problems = []

    def check(ok, msg):
        if not ok:
            problems.append(msg)

    for key in ("loss", "dcl", "sfl"):
        err = abs(float(out[key]) - float(g[key])) / abs(float(g[key]))
        print("%-5s fused path %.8f  reference %.8f  rel %.2e" % (key, float(out[key]), float(g[key]), err))
        check(err <= 1e-4, "%s: rel err %.3e > 1e-4" % (key, err))
    e = abs(float(out["grad_norm"]) - float(g["o64_grad_norm"])) / float(g["o64_grad_norm"])
    r = abs(float(g["grad_norm"]) - float(g["o64_grad_norm"])) / float(g["o64_grad_norm"])
    print("total gradient norm: fused path %.6f reference %.6f fp64 %.6f" % (float(out["grad_norm"]), float(g["grad_norm"]), float(g["o64_grad_norm"])))
    check(e <= max(4.0 * r, 1e-4), "total gradient norm: hip-vs-fp64 %.3e, reference-vs-fp64 %.3e" % (e, r))
    model = models[0]
    norms = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
    check(np.abs(norms - g["param_norms"]).max() <= 2e-5, "parameter norms after the step: %.3e" % np.abs(norms - g["param_norms"]).max())
    check(np.abs(sums - g["param_sums"]).max() <= 2e-4, "parameter sums after the step: %.3e" % np.abs(sums - g["param_sums"]).max())
    sd = model.state_dict()
    for name in (str(s) for s in g["buffers"]):
        for stat in (".running_mean", ".running_var"):
            err = rel_err(sd[name + stat], torch.from_numpy(g["buf::" + name + stat]))
            check(err <= 1e-4, "%s: %.3e > 1e-4" % (name + stat, err))
        check(int(sd[name + ".num_batches_tracked"]) == 2, name + ".num_batches_tracked")
    # the twin through the module path: same kernels on the same inputs
    loss, _, _, _ = steps[1].losses(batch)
    opts[1].zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    g_fused, g_modules = models[0].flat_gradients(), models[1].flat_gradients()
    check(abs(float(loss) - float(out["loss"])) <= 1e-6 * abs(float(loss)), "fused head loss vs modules: %.9f / %.9f" % (float(out["loss"]), float(loss)))
    worst = 0.0
    for (nm, pf), (_, pm) in zip(models[0].named_parameters(), models[1].named_parameters()):
        scale = max(float(pm.grad.abs().max()), 1e-30)
        worst = max(worst, float((pf.grad - pm.grad).abs().max()) / scale)
    print("gradients, fused path vs module path: worst tensor %.2e of its maximum; flat buffers %.2e" % (worst, rel_err(g_fused, g_modules)))
    check(rel_err(g_fused, g_modules) <= 1e-4, "flat gradient buffers, fused vs module path: %.3e" % rel_err(g_fused, g_modules))
    opts[1].step()
    assert_close(models[0].flat_parameters(), models[1].flat_parameters(), 1e-6, "parameters after the step, fused vs module path")
    assert not problems, "\n".join(problems)


def test_gap_scaled_pose_regime():
    """BASELINE.json configs[4] ("adjacent range 5-30"): per-sample frame gaps U{5..30}, poses scaled by gap / 10
    (synthetic.make_batch(gap_scale=(5, 30)), reference dataset.py:384-404 with train.py --adjacent_range 5 30) at the benchmark
    batch 8 x 256 x 320.  Up to 3x the motion of configs[1]: source boxes of the LDS-staged warp kernels outgrow their staging
    buffers and blocks fall back to the gather path -- asserted through endo_warp_fallback_blocks, so the fallback is known to
    have run.  (a) The geometry + loss chain on smooth depths, values and input gradients, against the fp64 oracle (bounds of
    test_geometry_and_losses_512x640).  (b) One full TrainingStep iteration against oracle.train_step: loss terms 1e-4."""
    n, h, w = 8, 256, 320
    lib = ea._lib.load()
    batch = synthetic.make_batch(n, h, w, seed=120, sparse_points=500, gap_scale=(5, 30))
    p1 = synthetic.smooth_depth(n, h, w, seed=220)
    p2 = synthetic.smooth_depth(n, h, w, seed=320)
    c1 = p1.double().requires_grad_(True)
    c2 = p2.double().requires_grad_(True)
    l_ref, dcl_ref, sfl_ref, _ = ostep.losses_from_depths(c1, c2, {k: v.double() for k, v in batch.items()})
    l_ref.backward()
    dbatch = to_dev(batch)
    b = dbatch["boundaries"]
    g1 = p1.to(dev()).requires_grad_(True)
    g2 = p2.to(dev()).requires_grad_(True)
    fwd, bwd = ctypes.c_longlong(), ctypes.c_longlong()
    assert lib.endo_warp_fallback_blocks(None, None, 1) == 0
    scaling, flow_layer, warp_layer = ea.DepthScalingLayer(), ea.FlowfromDepthLayer(), ea.DepthWarpingLayer()
    sfl_fn, dcl_fn = ea.SparseMaskedL1Loss(), ea.NormalizedDistanceLoss(h, w)
    s1, _ = scaling([g1, dbatch["sparse_depths_1"], dbatch["sparse_depth_masks_1"]])
    s2, _ = scaling([g2, dbatch["sparse_depths_2"], dbatch["sparse_depth_masks_2"]])
    f1 = flow_layer([s1, b, dbatch["translations_1_wrt_2"], dbatch["rotations_1_wrt_2"], dbatch["intrinsics"]]) * b
    f2 = flow_layer([s2, b, dbatch["translations_2_wrt_1"], dbatch["rotations_2_wrt_1"], dbatch["intrinsics"]]) * b
    sfl = 20.0 * 0.5 * (sfl_fn([dbatch["sparse_flows_1"] * b, f1, dbatch["sparse_flow_masks_1"] * b]) +
                        sfl_fn([dbatch["sparse_flows_2"] * b, f2, dbatch["sparse_flow_masks_2"] * b]))
    w21, i1 = warp_layer([s1, s2, b, dbatch["translations_1_wrt_2"], dbatch["rotations_1_wrt_2"], dbatch["intrinsics"]])
    w12, i2 = warp_layer([s2, s1, b, dbatch["translations_2_wrt_1"], dbatch["rotations_2_wrt_1"], dbatch["intrinsics"]])
    dcl = 0.1 * 0.5 * (dcl_fn([s1, w21, i1, dbatch["intrinsics"]]) + dcl_fn([s2, w12, i2, dbatch["intrinsics"]]))
    (dcl + sfl).backward()
    assert lib.endo_warp_fallback_blocks(ctypes.byref(fwd), ctypes.byref(bwd), 1) == 0
    print("gap-scaled poses: %d forward / %d backward blocks of the tiled warp kernels took the gather fallback (of %d per pass)" % (
        fwd.value, bwd.value, 2 * n * ((h + 15) // 16) * ((w + 31) // 32)))
    assert fwd.value > 0 and bwd.value > 0, "the large-motion batch did not exercise the gather fallback of the tiled warp kernels"
    assert_close(sfl, sfl_ref, 1e-4, "sparse flow loss, gap-scaled poses")
    assert_close(dcl, dcl_ref, 1e-4, "depth consistency loss, gap-scaled poses")
    for got, want, what in ((g1.grad, c1.grad, "grad pred 1, gap-scaled poses"), (g2.grad, c2.grad, "grad pred 2, gap-scaled poses")):
        err = (got.detach().double().cpu() - want).abs() / float(want.abs().max())          # see test_depth_warping_tiles_512x640
        assert float((err > 1e-4).double().mean()) <= 1e-3, what
        assert float(torch.quantile(err.reshape(-1)[::3], 0.99)) <= 1e-5, what
    # (b) the whole iteration on the same batch
    state, model = make_model(83, positive_depth=True)
    model.train()
    opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
    step = ea.train_step.TrainingStep(model, opt, h, w, sfl_weight=20.0, dcl_weight=0.1)
    out = step(dbatch, lr=1.0e-3)
    torch.cuda.synchronize()
    assert lib.endo_warp_fallback_blocks(ctypes.byref(fwd), ctypes.byref(bwd), 1) == 0
    ref = ostep.train_iteration(state, {}, batch, 1.0e-3)
    assert not out["skipped"] and not ref["skipped"]
    print("gap-scaled iteration: loss %.7f / oracle %.7f, dcl %.7f / %.7f, sfl %.7f / %.7f, grad norm %.5f / %.5f; fallback blocks %d / %d" % (
        out["loss"], float(ref["loss"]), float(out["dcl"]), float(ref["dcl"]), float(out["sfl"]), float(ref["sfl"]),
        float(out["grad_norm"]), float(ref["grad_norm"]), fwd.value, bwd.value))
    assert abs(out["loss"] - float(ref["loss"])) <= 1e-4 * abs(float(ref["loss"]))
    assert_close(out["dcl"], ref["dcl"], 1e-4, "dcl, gap-scaled iteration")
    assert_close(out["sfl"], ref["sfl"], 1e-4, "sfl, gap-scaled iteration")
    assert_close(out["grad_norm"], ref["grad_norm"], 5e-3, "grad norm, gap-scaled iteration")


def test_warp_consistency_call():
    """endo_warp_consistency (losses.warp_consistency): depth warp both ways + NormalizedDistanceLoss both ways, forward and backward
    in one call -- the chain of BASELINE.json's second metric -- against the modules under autograd (same kernels: 1e-6) and against
    the fp64 oracle (reference models.py:454-554, losses.py:112-146)."""
    for (n, h, w, seed) in ((2, 64, 96, 50), (8, 256, 320, 51), (2, 37, 53, 52)):
        batch, p1, p2, _ = geometry_inputs(n, h, w, seed)
        db = to_dev(batch)
        g1 = p1.to(dev()).requires_grad_(True)
        g2 = p2.to(dev()).requires_grad_(True)
        warp, dcl = ea.DepthWarpingLayer(), ea.NormalizedDistanceLoss(h, w)
        w21, i1 = warp([g1, g2, db["boundaries"], db["translations_1_wrt_2"], db["rotations_1_wrt_2"], db["intrinsics"]])
        w12, i2 = warp([g2, g1, db["boundaries"], db["translations_2_wrt_1"], db["rotations_2_wrt_1"], db["intrinsics"]])
        want = 0.7 * 0.5 * (dcl([g1, w21, i1, db["intrinsics"]]) + dcl([g2, w12, i2, db["intrinsics"]]))
        want.backward()
        loss, d1, d2 = ea.losses.warp_consistency(g1.detach(), g2.detach(), db["boundaries"], db["translations_1_wrt_2"],
                                                  db["rotations_1_wrt_2"], db["translations_2_wrt_1"], db["rotations_2_wrt_1"],
                                                  db["intrinsics"], dcl_weight=0.7)
        assert_close(loss, want.detach(), 1e-6, "warp_consistency loss vs modules %s" % ((n, h, w),))
        assert_close(d1, g1.grad, 1e-6, "warp_consistency grad 1 vs modules")
        assert_close(d2, g2.grad, 1e-5, "warp_consistency grad 2 vs modules")          # atomics: order of the scatter-adds
        if h * w <= 64 * 96:
            c1, c2 = p1.double().requires_grad_(True), p2.double().requires_grad_(True)
            b64 = {k: v.double() for k, v in batch.items()}
            r21, j1 = ogeo.depth_warping(c1, c2, b64["boundaries"], b64["translations_1_wrt_2"], b64["rotations_1_wrt_2"], b64["intrinsics"], 1e-8)
            r12, j2 = ogeo.depth_warping(c2, c1, b64["boundaries"], b64["translations_2_wrt_1"], b64["rotations_2_wrt_1"], b64["intrinsics"], 1e-8)
            ref = 0.7 * 0.5 * (olos.normalized_distance(c1, r21, j1, b64["intrinsics"]) + olos.normalized_distance(c2, r12, j2, b64["intrinsics"]))
            ref.backward()
            assert_close(loss, ref.detach(), 1e-5, "warp_consistency loss vs fp64 oracle")
            assert_close(d1, c1.grad, 1e-4, "warp_consistency grad 1 vs fp64 oracle")
            assert_close(d2, c2.grad, 1e-4, "warp_consistency grad 2 vs fp64 oracle")
    lib = ea._lib.load()
    assert lib.endo_warp_consistency(*([None] * 8), 1.0, 1e-8, *([None] * 4), 1, 8, 8, None) == -1




def test_full_size_pair_backward_on_pattern():
    """The grouped 2 x 8 x 256 x 320 pass bench.py times (forward_pair: 16 samples per launch -- the 32x16 forward tiles,
    split-K coarse levels, n-split weight gradients and 8-wave fused data gradients that only these grids select), all 210
    parameter gradients against the CPU oracle evaluated on the activation pattern the pass itself took
    (device_pattern.py), frame by frame with each frame's own BatchNorm statistics.  The oracle runs in fp32 here (fp64 at
    this size costs minutes and 40 GB): both sides then carry fp32 rounding, ~1e-5 each at the smaller sizes, and the bound
    is BASELINE.json's 1e-4 on every tensor."""
    n, h, w = 8, 256, 320
    state, model = make_model(58)
    rng = np.random.default_rng(14)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)) for _ in range(2)]
    model.train()
    y1, y2 = model.forward_pair(xs[0].to(dev()), xs[1].to(dev()))
    patterns = pattern_of(y1, model, n, h, w, groups=2)
    ((y1 * cots[0].to(dev())).sum() + (y2 * cots[1].to(dev())).sum()).backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    names = onet.trainable_names()
    total = None
    for x, cot, pat, got in zip(xs, cots, patterns, (y1, y2)):
        st = {k: v.clone() for k, v in state.items()}
        for nm in names:
            st[nm].requires_grad_(True)
        y = onet.forward(st, x, training=True, pattern=pat)
        assert_close(got, y.detach(), 1e-4, "depth of one frame of the pair vs the oracle on the same pattern")
        grads = torch.autograd.grad((y * cot).sum(), [st[nm] for nm in names])
        total = list(grads) if total is None else [a + b for a, b in zip(total, grads)]
        del y, grads, st
    ref = dict(zip(names, total))
    assert_grads_on_pattern(params, ref, None, 1e-4, "full-size pair backward")


def test_wgrad_f34_against_the_direct_kernels_at_benchmark_size():
    """The F(3x3, 4x4) Winograd weight gradient (csrc/wgrad_f34_kernels.h, ENDO_OPT_WGRAD_F34; levels 0-4 at this size, the last strip of levels 3 and 4 partly outside the image) against the direct
    kernels it replaces, on the SAME model, inputs and forward pass -- the grouped 2 x 8 x 256 x 320 launch bench.py times, so both forms see
    identical activations, BatchNorm statistics of both sample groups and prepared gradients: every dense layer's weight gradient within
    2e-5 of its maximum (measured 8e-6: the two forms' fp32 roundings, 3-6e-6 and 5e-7 against fp64 in tools/x3_bench), every other
    gradient (BN parameters, the other layers' weights: same kernels, same inputs) within the atomics' summation order, 1e-5."""
    n, h, w = 8, 256, 320
    state, model = make_model(61)
    rng = np.random.default_rng(23)
    xs = [torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)).to(dev()) for _ in range(2)]
    cots = [torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32)).to(dev()) for _ in range(2)]
    model.train()
    grads = {}
    for form in (1, 0):
        model.set_kernel_option(OPT_WGRAD_F34, form)
        model.zero_grad(set_to_none=True)
        y1, y2 = model.forward_pair(xs[0], xs[1])
        ((y1 * cots[0]).sum() + (y2 * cots[1]).sum()).backward()
        torch.cuda.synchronize()
        grads[form] = {nm: p.grad.detach().clone() for nm, p in model.named_parameters()}
    model.set_kernel_option(OPT_WGRAD_F34, 1)
    worst_dense, worst_other = (0.0, ""), (0.0, "")
    gmax = max(float(b.abs().max()) for b in grads[0].values())          # (conv biases in front of a BatchNorm have gradients that are rounding noise around 0)
    for nm, a in grads[1].items():
        b = grads[0][nm]
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-4 * gmax))
        dense_w = ("denseBlocks" in nm or "bottleneck" in nm) and nm.endswith("conv.weight")
        if dense_w:
            worst_dense = max(worst_dense, (err, nm))
        elif ".norm." in nm or not nm.endswith(".bias"):          # (a conv bias in front of a BatchNorm: its gradient is the rounding noise of sums that cancel)
            worst_other = max(worst_other, (err, nm))
    print("F(3x3, 4x4) vs direct weight gradients: worst dense-layer weight %.2e (%s), worst other tensor %.2e (%s)" % (worst_dense + worst_other))
    assert worst_dense[0] <= 2e-5, worst_dense
    assert worst_dense[0] > 0.0, "the option did not change the kernel"
    assert worst_other[0] <= 1e-5, worst_other


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json config 2: N = 8, 256 x 320)
# ---------------------------------------------------------------------------------------------
def test_full_size_properties():
    n, h, w = 8, 256, 320
    batch = to_dev(synthetic.make_batch(n, h, w, seed=80))
    b = batch["boundaries"]
    eye = torch.eye(3, device=dev()).expand(n, 3, 3).contiguous()
    zero_t = torch.zeros(n, 3, 1, device=dev())
    d = synthetic.smooth_depth(n, h, w, seed=81).to(dev())
    ones = torch.ones_like(d)
    # identity pose: zero flow; warp = half-pixel-shifted bilinear resample; mask zero on row/col 0
    flow = ea.FlowfromDepthLayer()([d, ones, zero_t, eye, batch["intrinsics"]])
    assert float(flow.abs().max()) < 1e-5
    warped, inter = ea.DepthWarpingLayer()([d, d, ones, zero_t, eye, batch["intrinsics"]])
    assert float(inter[:, :, 0, :].sum()) == 0 and float(inter[:, :, :, 0].sum()) == 0
    assert float(inter[:, :, 1:, 1:].min()) == 1
    want = 0.25 * (d[:, :, 1:, 1:] + d[:, :, :-1, 1:] + d[:, :, 1:, :-1] + d[:, :, :-1, :-1])
    assert_close(warped[:, :, 1:, 1:], want, 1e-5, "identity warp")
    # depth scaling is invariant to the scale of the prediction
    s1, _ = ea.DepthScalingLayer()([d, batch["sparse_depths_1"], batch["sparse_depth_masks_1"]])
    s2, _ = ea.DepthScalingLayer()([3.0 * d, batch["sparse_depths_1"], batch["sparse_depth_masks_1"]])
    assert_close(s2, s1, 1e-5, "scale invariance of depth scaling")
    # losses: zero for identical inputs, per-sample means (shard property used by data parallelism)
    dcl = ea.NormalizedDistanceLoss(h, w)
    assert float(dcl([d, d, b, batch["intrinsics"]])) == 0.0
    full = dcl([d, 1.1 * d, b, batch["intrinsics"]])
    halves = 0.5 * (dcl([d[:4], 1.1 * d[:4], b[:4], batch["intrinsics"][:4]]) +
                    dcl([d[4:], 1.1 * d[4:], b[4:], batch["intrinsics"][4:]]))
    assert_close(full, halves, 1e-5, "mean of shard means")
    # network: output non-negative, gradient linear in the cotangent, BN batch statistics per call
    _, model = make_model(55)
    model.train()
    x = (batch["colors_1"] * b)
    y = model(x)
    assert y.shape == (n, 1, h, w) and float(y.min()) >= 0.0 and torch.isfinite(y).all()
    cot = torch.randn(n, 1, h, w, device=dev(), generator=torch.Generator(device=dev()).manual_seed(1))
    opt = ea.optim.FusedClipSGD(model, lr=0.0)
    opt.zero_grad()
    (y * cot).sum().backward()
    g1 = model.flat_gradients().clone()
    opt.zero_grad()
    y = model(x)
    (y * (2.0 * cot)).sum().backward()
    assert_close(model.flat_gradients(), 2.0 * g1, 1e-3, "gradient linearity at full size")
    assert torch.isfinite(g1).all()
