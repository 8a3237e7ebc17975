"""Fused ``clip_grad_norm_(params, 10.0)`` + ``SGD(momentum=0.9)`` over the flat buffers of
``models.FCDenseNet`` (reference train.py:202, 327-328): two HIP kernels per step instead of
~850 small launches.  Subclasses ``torch.optim.Optimizer`` so ``scheduler.CyclicLR`` (which insists
on an Optimizer, reference scheduler.py:84-86) and ``zero_grad`` keep working.
"""

import torch
from torch.optim import Optimizer

from . import _lib


class FusedClipSGD(Optimizer):
    def __init__(self, model, lr, momentum=0.9, max_norm=10.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, momentum=momentum, max_norm=max_norm))
        self._momentum = None
        self._norm = None
        self._steps = 0

    def zero_grad(self, set_to_none=False):
        """Keeps the gradient views attached and clears the flat buffer with one memset."""
        grads = self.model.flat_gradients(create=False)
        if grads is not None:
            grads.zero_()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """grad_scale: 1/world_size after a summed all-reduce.  Returns the pre-clip gradient norm
        as a 0-dim fp64 device tensor (no host sync)."""
        lib = _lib.load()
        group = self.param_groups[0]
        params = self.model.flat_parameters()
        grads = self.model.flat_gradients()
        if self._momentum is None or self._momentum.device != params.device:
            self._momentum = torch.zeros_like(params)
            self._norm = torch.zeros(2, dtype=torch.float64, device=params.device)
            self._steps = 0
        _lib.check(lib.endo_sgd_clip_step(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(self._momentum), _lib.ptr(self._norm),
                                          params.numel(), float(group['lr']), float(group['momentum']), float(group['max_norm']),
                                          float(grad_scale), 1 if self._steps == 0 else 0, _lib.stream()), "endo_sgd_clip_step")
        self._steps += 1
        return self._norm[1]

    def state_dict(self):
        return {"momentum": self._momentum, "steps": self._steps,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, state):
        self._momentum = state["momentum"]
        self._steps = state["steps"]
        if self._momentum is not None:
            self._norm = torch.zeros(2, dtype=torch.float64, device=self._momentum.device)
        for g, s in zip(self.param_groups, state["param_groups"]):
            g.update(s)
