"""Development aid: the bf16-storage convolution brick on the widest dense layer of the network (level 0, Cin = 180 -> 12, 16 samples of
256 x 320) and on a few other layer shapes; prints time, algorithmic TB/s and direct-count TFLOP/s.  Run it under rocprofv3 --pmc for
the SQ counters (tools/pmc_sq_report.py)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
lib = ea._lib.load()
dev = torch.device("cuda:0")


def bench(n, h, w, t, ic0, cin, cout, oc0, ks, bn_on, reps=20, out_hw=None):
    g = torch.Generator(device=dev).manual_seed(3)
    xin = (torch.rand((n, h, w, t), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    weight = torch.randn((cout, cin, ks, ks), device=dev, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    bn = torch.stack([torch.rand(cin, device=dev, generator=g) + 0.5, torch.rand(cin, device=dev, generator=g) * 0.2 - 0.1], dim=1).contiguous()
    bias = torch.zeros(cout, device=dev)
    wl = torch.empty(int(lib.endo_bf16_conv_weight_elems(cout, cin, ks)), dtype=torch.bfloat16, device=dev)
    assert lib.endo_bf16_conv_weights(weight.data_ptr(), cout, cin, ks, wl.data_ptr(), None) == 0
    sums = torch.zeros((cout, 2), dtype=torch.float64, device=dev)
    out = xin if oc0 + cout <= t and oc0 >= ic0 + cin else torch.empty((n, h, w, cout), dtype=torch.bfloat16, device=dev)
    out_t = t if out is xin else cout
    oc = oc0 if out is xin else 0

    def run():
        return lib.endo_bf16_conv(xin.data_ptr(), t, ic0, cin, bn.data_ptr() if bn_on else None, wl.data_ptr(), bias.data_ptr(), out.data_ptr(), out_t,
                                  oc, cout, sums.data_ptr(), n, h, w, ks, 0, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gbytes = n * h * w * (cin + cout) * 2 / 1e9
    print("ks %d  %2d x %3d x %3d  cin %3d -> %3d  bn %d : %7.1f us  %5.2f TB/s algorithmic  %6.1f TFLOP/s" % (
        ks, n, h, w, cin, cout, int(bn_on), us, gbytes / us * 1e3, 2.0 * n * h * w * cin * cout * ks * ks / us / 1e6))


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bench(16, 256, 320, 192, 0, 180, 12, 180, 3, True, reps)
bench(16, 256, 320, 192, 0, 180, 12, 180, 3, False, reps)
bench(16, 256, 320, 192, 48, 48, 12, 96, 3, True, reps)
bench(16, 128, 160, 240, 0, 228, 12, 228, 3, True, reps)
bench(16, 256, 320, 192, 48, 96, 96, 0, 1, True, reps)
