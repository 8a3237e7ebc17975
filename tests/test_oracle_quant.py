"""CPU checks of the oracle's storage-rounding hooks (oracle.network.forward(quant=...)): the restatement of WHERE the 16-bit-storage
kernel families round (tests/test_gpu_bf16.py compares endo_net16_* / endo_net16h_* against it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import network as onet


def test_ste_rounds_forward_and_passes_gradients_through():
    t = torch.linspace(-3.0, 3.0, 1001, dtype=torch.float64, requires_grad=True)
    for ste, dtype, ulp in ((onet.bf16_ste, torch.bfloat16, 2.0 ** -8), (onet.fp16_ste, torch.float16, 2.0 ** -11)):
        q = ste(t)
        assert torch.equal(q.detach(), t.detach().to(dtype).double())                 # the forward value is the rounded one
        big = t.detach().abs() > 1e-3
        assert float(((q - t).abs()[big] / t.detach().abs()[big]).max()) <= ulp          # within half a unit of the last place
        (g,) = torch.autograd.grad(q.sum(), t)
        assert torch.equal(g, torch.ones_like(t))                                      # identity in the backward direction
        assert torch.equal(ste(q.detach()), q.detach())                                # idempotent


def test_quantised_forward_is_a_small_perturbation_with_exact_gradients():
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(5), 6))
    x = torch.rand((2, 3, 64, 64), dtype=torch.float64) * 2 - 1
    outs = {}
    for name, quant in (("plain", None), ("bf16", onet.bf16_ste), ("fp16", onet.fp16_ste)):
        st = {k: (v.double().requires_grad_(k in onet.trainable_names()) if v.is_floating_point() else v.clone()) for k, v in state.items()}
        y = onet.forward(st, x, training=True, quant=quant)
        y.sum().backward()
        outs[name] = (y.detach(), {k: st[k].grad for k in onet.trainable_names()})
    y0 = outs["plain"][0]
    e16 = float((outs["bf16"][0] - y0).abs().max() / y0.abs().max())
    eh = float((outs["fp16"][0] - y0).abs().max() / y0.abs().max())
    assert 1e-4 < e16 < 3e-2 and 1e-5 < eh < 4e-3 and eh < e16, (e16, eh)
    for k, g in outs["bf16"][1].items():
        assert g is not None and torch.isfinite(g).all(), k                             # every parameter still receives a gradient


def test_quant_and_pattern_compose():
    """quant + pattern: the ReLU decisions come from the pattern, the values are still rounded"""
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(5), 6))
    x = torch.rand((1, 3, 32, 32), dtype=torch.float64) * 2 - 1
    st = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    trace = {}
    y = onet.forward({k: v.clone() for k, v in st.items()}, x, training=False, quant=onet.bf16_ste, trace=trace)
    # every stored convolution output is representable in bf16
    for name, t in trace.items():
        if name.startswith("conv::"):
            assert torch.equal(t, t.to(torch.bfloat16).double()), name
    assert torch.isfinite(y).all()
