"""Timing PRINTS of the 16-bit-storage family (no pass / fail gate on speed): marker `bench`, not `gpu` -- the round-end `pytest -m gpu`
run does not spend its minutes here.  Run by hand on a GPU box: `python -m pytest tests/test_bench_prints.py -m bench -s`.
Without a GPU they skip."""

import importlib

import pytest
import torch

pytestmark = [pytest.mark.bench, pytest.mark.skipif(not torch.cuda.is_available(), reason="timing prints need an MI355X")]
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")

from test_gpu_bf16 import dev, reference          # noqa: E402 -- the brick test's helpers


def test_bf16_conv_dense_layer_timing():
    """Not a pass / fail performance gate: prints the time of the widest dense layer of the network (level 0, Cin = 180 -> 12, 16
    samples of 256 x 320) in the bf16-storage brick next to the fp32 Winograd kernel's 470 us (profiles/r03_j_kernel_stats_by_grid.txt),
    and checks the result against fp64 on a sub-block."""
    lib = ea._lib.load()
    n, h, w, t, ic0, cin, cout, oc0 = 16, 256, 320, 192, 0, 180, 12, 180
    g = torch.Generator(device=dev()).manual_seed(3)
    blk = 32                                       # the level buffers' layout: [n][t / 32][h][w][32]
    xin = (torch.rand(n * h * w * t, device=dev(), generator=g) * 2 - 1).to(torch.bfloat16)
    weight = torch.randn((cout, cin, 3, 3), device=dev(), generator=g) * (2.0 / (cin * 9)) ** 0.5
    bn = torch.stack([torch.rand(cin, device=dev(), generator=g) + 0.5, torch.rand(cin, device=dev(), generator=g) * 0.2 - 0.1], dim=1).contiguous()
    bias = torch.zeros(cout, device=dev())
    wl = torch.empty(int(lib.endo_bf16_conv_weight_elems(cout, cin, 3)), dtype=torch.bfloat16, device=dev())
    assert lib.endo_bf16_conv_weights(weight.data_ptr(), cout, cin, 3, wl.data_ptr(), None) == 0
    sums = torch.zeros((cout, 2), dtype=torch.float64, device=dev())

    def run():
        return lib.endo_bf16_conv(xin.data_ptr(), t, blk, ic0, cin, bn.data_ptr(), wl.data_ptr(), bias.data_ptr(), xin.data_ptr(), t, blk, oc0, cout,
                                  sums.data_ptr(), n, h, w, 3, 0, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = n * h * w * (cin + cout) * 2 / 1e9
    print("bf16-storage dense layer, level 0, Cin 180 -> 12, 16 x 256 x 320: %.1f us = %.2f TB/s of algorithmic bytes, %.1f TFLOP/s" % (
        us, gb / us * 1e3, 2.0 * n * h * w * cin * cout * 9 / us / 1e6))
    # correctness on the first sample's top-left block
    y = torch.empty((1, cout, h, w), dtype=torch.float32, device=dev())
    assert lib.endo_bf16_unpack_nhwc(xin.data_ptr(), y.data_ptr(), 1, cout, h, w, t, blk, oc0, None) == 0
    xa = torch.empty((1, cin, h, w), dtype=torch.float32, device=dev())
    assert lib.endo_bf16_unpack_nhwc(xin.data_ptr(), xa.data_ptr(), 1, cin, h, w, t, blk, ic0, None) == 0
    x0 = xa[:, :, :40, :40].cpu()
    ref = reference(x0, 0, cin, weight.cpu(), bias.cpu(), bn.cpu(), 3, 0)[0, :, :32, :32]
    err = float((y[0, :, :32, :32].cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= 6e-3, err


def test_bf16_storage_forward_timing():
    """Prints (no gate) the forward time of the benchmark batch -- 16 frames of 256 x 320, one call -- over bf16 level buffers next to
    the fp32 path's grouped pair forward (5.9 ms of a training step, profiles/r03_i_stream_timeline.txt)."""
    n, h, w = 16, 256, 320
    m = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(m, mode="fan_in", activation_mode="relu", distribution="normal")
    m = m.to(dev()).train()
    x = torch.rand((n, 3, h, w), device=dev()) * 2 - 1
    res = {}
    for name, fn in (("bf16 storage", lambda: m.forward_bf16_storage(x)), ("fp32", lambda: m.forward_pair(x[:8], x[8:]))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    print("forward of 16 x 256 x 320, training mode: bf16 storage %.3f ms, fp32 (grouped pair) %.3f ms" % (res["bf16 storage"], res["fp32"]))


def test_bf16_storage_step_timing():
    """Prints (no gate) forward + backward of the benchmark batch (two calls of 8 frames of 256 x 320, as a training step makes them)
    over bf16 level buffers next to the fp32 path's grouped pair."""
    n, h, w = 8, 256, 320
    m = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(m, mode="fan_in", activation_mode="relu", distribution="normal")
    m = m.to(dev()).train()
    x1 = torch.rand((n, 3, h, w), device=dev()) * 2 - 1
    x2 = torch.rand((n, 3, h, w), device=dev()) * 2 - 1
    g = torch.randn((n, 1, h, w), device=dev())

    def step16():
        y1 = m.forward_bf16_storage(x1); y2 = m.forward_bf16_storage(x2)
        torch.autograd.backward([y1, y2], [g, g])

    def step32():
        y1, y2 = m.forward_pair(x1, x2)
        torch.autograd.backward([y1, y2], [g, g])
    res = {}
    for name, fn in (("bf16 storage", step16), ("fp32", step32)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 5
    print("network forward + backward, two batches of 8 x 256 x 320, training mode: bf16 storage %.3f ms, fp32 (grouped pair) %.3f ms" % (
        res["bf16 storage"], res["fp32"]))


