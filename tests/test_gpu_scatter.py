"""GPU sparse scatter (endo_sparse_scatter through the C ABI) against the golden fixture generated from the reference
(tests/golden/make_golden.py, utils.get_torch_training_data on the shipped example sequence) and against the CPU
oracle on seeded synthetic clouds with many pixel collisions.  Index / mask / flow planes: bit exact.  Depth values
are fp64 dot products cast to fp32: bit exact on these inputs as well (asserted), see csrc/scatter.hip."""
import numpy as np
import pytest
import torch

from oracle import scatter as oracle_scatter

pytestmark = pytest.mark.gpu


def _dense(g, tag, name):
    shape = tuple(int(v) for v in g[tag + name + "_shape"])
    want = np.zeros((2, shape[1] * shape[2], shape[3]), np.float32)
    i = g[tag + name + "_idx"]
    want[i[0], i[1]] = g[tag + name + "_val"]
    return want.reshape(shape)


def _to_hwc(t, b):
    return t[:, b].permute(0, 2, 3, 1).contiguous().cpu().numpy()


def test_scatter_golden_batched(pkg, golden):
    g = golden("scatter_example.npz")
    n_pairs = len(g["pairs"])
    vis = np.concatenate([g["pair%d_visibility" % i] for i in range(n_pairs)], axis=1)          # (P, 2 * pairs)
    seq = pkg.scatter.SequenceScatter(g["points"], g["mask"], vis, g["clean"], list(range(2 * n_pairs)))
    ext = np.stack([g["pair%d_extrinsics" % i] for i in range(n_pairs)])
    proj = np.stack([g["pair%d_projections" % i] for i in range(n_pairs)])
    out = seq.planes(ext, proj, [[2 * i, 2 * i + 1] for i in range(n_pairs)])
    torch.cuda.synchronize()
    for i in range(n_pairs):
        for name in ("depth_masks", "depths", "flow_masks", "flows"):
            np.testing.assert_array_equal(_to_hwc(out[name], i), _dense(g, "pair%d_" % i, name), err_msg="%s pair %d" % (name, i))


def test_scatter_dropin_signature(pkg, golden):
    g = golden("scatter_example.npz")
    got = pkg.scatter.get_torch_training_data(g["pair1_extrinsics"], g["pair1_projections"], g["pairs"][1], g["points"], g["mask"],
                                              g["pair1_visibility"], g["clean"], list(g["pairs"][1]))
    for name, arr in zip(("depth_masks", "depths", "flow_masks", "flows"), got):
        assert arr.dtype == np.float32
        np.testing.assert_array_equal(arr, _dense(g, "pair1_", name))


def _synthetic(seed, n_points, height, width, batch):
    rng = np.random.default_rng(seed)
    k = np.array([[0.55 * width, 0, 0.5 * width], [0, 0.55 * width, 0.45 * height], [0, 0, 1.0]])
    points = np.concatenate([rng.uniform(-1.2, 1.2, (n_points, 2)), rng.uniform(0.6, 3.0, (n_points, 1)), np.ones((n_points, 1))], axis=1)
    points[: n_points // 50, 2] *= -1.0                      # some points behind the camera
    ext = np.zeros((batch, 2, 4, 4))
    proj = np.zeros((batch, 2, 3, 4))
    for b in range(batch):
        for i in range(2):
            ang = rng.normal(0, 0.05, 3)
            rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
            ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
            rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
            e = np.eye(4)
            e[:3, :3] = rx @ ry @ rz
            e[:3, 3] = rng.normal(0, 0.08, 3)
            ext[b, i] = e
            proj[b, i] = k @ e[:3]
    yy, xx = np.mgrid[0:height, 0:width]
    mask = np.where(((xx - width / 2) / (0.48 * width)) ** 2 + ((yy - height / 2) / (0.48 * height)) ** 2 < 1.0, 255, 0).astype(np.uint8)
    vis = (rng.uniform(size=(n_points, 2 * batch)) > 0.2).astype(np.float32)
    clean = (rng.uniform(size=n_points) > 0.1).astype(np.float32)
    return points, ext, proj, mask, vis, clean


@pytest.mark.parametrize("n_points,height,width,batch,use_clean", [(6000, 32, 40, 3, True), (500, 256, 320, 2, False), (1, 8, 8, 1, True)])
def test_scatter_vs_oracle_collisions(pkg, n_points, height, width, batch, use_clean):
    points, ext, proj, mask, vis, clean = _synthetic(7 + n_points, n_points, height, width, batch)
    clean_arg = clean if use_clean else np.zeros((0,), np.float32)
    seq = pkg.scatter.SequenceScatter(points, mask, vis, clean_arg, list(range(2 * batch)))
    out = seq.planes(ext, proj, [[2 * b, 2 * b + 1] for b in range(batch)], depth_multiplier=1.0)
    torch.cuda.synchronize()
    hits = 0
    for b in range(batch):
        want = oracle_scatter.sparse_planes(ext[b], proj[b], vis[:, 2 * b:2 * b + 2], clean_arg, points, mask)
        hits += int(want[0].sum())
        for name, arr in zip(("depth_masks", "depths", "flow_masks", "flows"), want):
            np.testing.assert_array_equal(_to_hwc(out[name], b), arr, err_msg="%s pair %d" % (name, b))
    if n_points >= 500:
        assert hits > 50          # the case is not vacuous
    if n_points == 6000:
        assert hits < 2 * batch * height * width and hits > 0.3 * 2 * batch * (mask == 255).sum()    # heavy collisions


def test_scatter_empty_cloud_and_multiplier(pkg):
    points, ext, proj, mask, vis, clean = _synthetic(3, 64, 16, 20, 1)
    seq = pkg.scatter.SequenceScatter(np.zeros((0, 4)), mask, np.zeros((0, 2), np.float32), [], [0, 1])
    out = seq.planes(ext, proj, [[0, 1]])
    assert all(float(v.abs().sum()) == 0.0 for v in out.values())
    seq = pkg.scatter.SequenceScatter(points, mask, vis, [], [0, 1])
    a = seq.planes(ext, proj, [[0, 1]])
    b = seq.planes(ext, proj, [[0, 1]], depth_multiplier=0.25)
    torch.testing.assert_close(b["depths"], a["depths"] * 0.25, rtol=0, atol=0)
    torch.testing.assert_close(b["flows"], a["flows"], rtol=0, atol=0)


def test_point_cloud_golden_and_oracle(pkg, golden):
    """endo_point_cloud against the fixture generated from reference utils.point_cloud_from_depth (bit exact: the kernel
    keeps numpy's three separate float32 roundings) and against the oracle at 256 x 320 with the synthetic mask."""
    from oracle import pointcloud as oracle_pc
    g = golden("point_cloud.npz")
    cases = (("all", dict(point_cloud_downsampling=1)), ("ds2", dict(point_cloud_downsampling=2)),
             ("thr", dict(point_cloud_downsampling=1, min_threshold=60, max_threshold=180)))
    for tag, kw in cases:
        got = pkg.utils.point_cloud_from_depth(g["depth"], g["color"], g["mask"], g["intrinsics"], **kw)
        assert got.dtype == np.float32 and got.shape[1] == 6
        np.testing.assert_array_equal(got, g["points_" + tag], err_msg=tag)
    rng = np.random.default_rng(5)
    h, w = 256, 320
    batch = pkg.synthetic.make_batch(1, h, w, seed=3)
    mask = batch["boundaries"][0, 0].numpy()
    depth = (rng.uniform(0.01, 2.0, (h, w)).astype(np.float32) * mask).astype(np.float32)
    color = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    k = batch["intrinsics"][0].numpy().astype(np.float32)
    for ds in (1, 3, 4):
        got = pkg.utils.point_cloud_from_depth(torch.from_numpy(depth), color, mask, k, ds)
        want = oracle_pc.point_cloud_from_depth(depth, color, mask, k, ds)
        np.testing.assert_array_equal(got, want, err_msg="downsampling %d" % ds)
    empty = pkg.utils.point_cloud_from_depth(depth, color, np.zeros_like(mask), k, 1)
    assert empty.shape == (0, 6)


def test_training_batch_on_device(pkg, golden):
    """SequenceScatter.training_batch: the 14 non-image tensors of a training batch assembled on the device from the resident
    example sequence -- sparse planes as the reference fixture (depths divided by the sequence's scale as dataset.py:391-392
    does, in fp32), relative poses bit-exact against oracle/poses.py (numpy restatement of dataset.py:384-399), boundary
    plane and intrinsics as dataset.py:421-430 -- and a TrainingStep consumes them as they are."""
    from oracle import poses as oposes
    g = golden("scatter_example.npz")
    n_pairs = len(g["pairs"])
    vis = np.concatenate([g["pair%d_visibility" % i] for i in range(n_pairs)], axis=1)
    ext = np.concatenate([g["pair%d_extrinsics" % i] for i in range(n_pairs)])           # views: a0, b0, a1, b1, ...
    proj = np.concatenate([g["pair%d_projections" % i] for i in range(n_pairs)])
    scale = float(g["scale"])
    seq = pkg.scatter.SequenceScatter(g["points"], g["mask"], vis, g["clean"], list(range(2 * n_pairs)), extrinsics=ext,
                                      projections=proj, intrinsic_matrix=g["intrinsics"], estimated_scale=scale)
    positions = [(2 * i, 1) for i in range(n_pairs)]
    batch = seq.training_batch(positions)
    torch.cuda.synchronize()
    assert sorted(batch) == sorted(k for k in pkg.synthetic.BATCH_KEYS if not k.startswith("colors"))
    h, w = seq.height, seq.width
    for i in range(n_pairs):
        tag = "pair%d_" % i
        want_depth = _dense(g, tag, "depths")
        want_depth /= scale                                              # float32 array /= python float, as the reference
        for f, sfx in ((0, "_1"), (1, "_2")):
            np.testing.assert_array_equal(batch["sparse_depths" + sfx][i, 0].cpu().numpy(), want_depth[f, :, :, 0])
            np.testing.assert_array_equal(batch["sparse_depth_masks" + sfx][i, 0].cpu().numpy(), _dense(g, tag, "depth_masks")[f, :, :, 0])
            np.testing.assert_array_equal(batch["sparse_flow_masks" + sfx][i, 0].cpu().numpy(), _dense(g, tag, "flow_masks")[f, :, :, 0])
            np.testing.assert_array_equal(batch["sparse_flows" + sfx][i].permute(1, 2, 0).cpu().numpy(), _dense(g, tag, "flows")[f])
        r12, r21, t12, t21 = oposes.relative_poses(ext[2 * i], ext[2 * i + 1], scale)
        np.testing.assert_array_equal(batch["rotations_1_wrt_2"][i].cpu().numpy(), r12)
        np.testing.assert_array_equal(batch["rotations_2_wrt_1"][i].cpu().numpy(), r21)
        np.testing.assert_array_equal(batch["translations_1_wrt_2"][i].cpu().numpy(), t12)
        got_t21 = batch["translations_2_wrt_1"][i].cpu().numpy()
        # the last product is a float32 3x3 . 3x1 whose summation order / fusion BLAS is free to choose: one ulp
        assert np.abs(got_t21 - t21).max() <= 1.2e-7 * max(np.abs(t21).max(), 1e-30), (got_t21, t21)
    np.testing.assert_array_equal(batch["boundaries"][0, 0].cpu().numpy(), oposes.boundary_plane(g["mask"]))
    np.testing.assert_array_equal(batch["intrinsics"][1].cpu().numpy(), np.asarray(g["intrinsics"])[:3, :3].astype(np.float32))
    assert batch["boundaries"].shape == (n_pairs, 1, h, w) and batch["translations_2_wrt_1"].shape == (n_pairs, 3, 1)
    # straight into a training step (the example frames are 256 x 320)
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = pkg.models.FCDenseNet57(1)
    pkg.utils.kaiming_weight_zero_bias(model, distribution="normal")
    with torch.no_grad():
        model.finalConv.bias.add_(4.0)
    model = model.to(dev).train()
    opt = pkg.optim.FusedClipSGD(model, lr=1.0e-4)
    step = pkg.train_step.TrainingStep(model, opt, h, w)
    gen = torch.Generator(device=dev).manual_seed(2)
    batch["colors_1"] = torch.rand(n_pairs, 3, h, w, device=dev, generator=gen) * 2 - 1
    batch["colors_2"] = torch.rand(n_pairs, 3, h, w, device=dev, generator=gen) * 2 - 1
    out = step(batch)
    assert not out["skipped"] and np.isfinite(out["loss"]) and float(out["grad_norm"]) > 0.0
