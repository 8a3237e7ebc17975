"""Fused ``clip_grad_norm_(params, 10.0)`` + ``SGD(momentum=0.9)`` over the flat buffers of
``models.FCDenseNet`` (reference train.py:202, 327-328): two HIP kernels per step instead of
~850 small launches.  Subclasses ``torch.optim.Optimizer`` so ``scheduler.CyclicLR`` (which insists
on an Optimizer, reference scheduler.py:84-86) and ``zero_grad`` keep working.

``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.SGD``'s layout
(``{'state': {i: {'momentum_buffer': tensor}}, 'param_groups': [...]}``), the one stored under the ``'optimizer'`` key of
the reference's checkpoints (reference utils.py:674-682), so optimizer state moves both ways between the stacks.
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

    def _ensure_state(self):
        params = self.model.flat_parameters()
        if self._momentum is None or self._momentum.device != params.device or self._momentum.numel() != params.numel():
            self._momentum = torch.zeros_like(params)
            self._steps = 0
        if self._norm is None or self._norm.device != params.device:
            self._norm = torch.zeros(2, dtype=torch.float64, device=params.device)

    @torch.no_grad()
    def step(self, grad_scale=1.0, skip_flag=None):
        """grad_scale: 1/world_size after a summed all-reduce.  skip_flag: None, or a 1-element fp32 device tensor -- when it is
        non-zero the kernels leave parameters and momentum untouched (the non-finite-loss guard of reference train.py:317-322,
        decided on the device so that the host never waits for the loss before the backward pass).  Returns the pre-clip gradient
        norm as a 0-dim fp64 device tensor of its own (no host sync; later steps do not overwrite it)."""
        lib = _lib.load()
        group = self.param_groups[0]
        params = self.model.flat_parameters()
        grads = self.model.flat_gradients()
        self._ensure_state()
        # (first_step: the momentum buffer starts as zeros, so mu * buf + g IS g on the first step; the flag is kept for callers of the
        # C entry point that hand over an uninitialised buffer)
        _lib.check(lib.endo_sgd_clip_step(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(self._momentum), _lib.ptr(self._norm),
                                          params.numel(), float(group['lr']), float(group['momentum']), float(group['max_norm']),
                                          float(grad_scale), 0, _lib.ptr(skip_flag) if skip_flag is not None else None,
                                          _lib.stream()), "endo_sgd_clip_step")
        self._steps += 1
        return self._norm[1].clone()

    # ---- torch.optim.SGD wire format ------------------------------------------------------------
    def state_dict(self):
        params = list(self.model.parameters())
        offsets = self.model._offsets
        state = {}
        if self._momentum is not None and self._steps > 0:
            for i, (p, off) in enumerate(zip(params, offsets)):
                state[i] = {"momentum_buffer": self._momentum[off:off + p.numel()].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        # the keys torch.optim.SGD expects in a param group, so that SGD.load_state_dict takes the dict as is
        for key, val in (("dampening", 0), ("weight_decay", 0), ("nesterov", False), ("maximize", False), ("foreach", None),
                         ("differentiable", False), ("fused", None)):
            group.setdefault(key, val)
        group["params"] = list(range(len(params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        params = list(self.model.parameters())
        offsets = self.model._offsets
        groups = state_dict["param_groups"]
        index = [i for g in groups for i in g["params"]]
        if len(index) != len(params):
            raise ValueError("optimizer state covers %d parameters, the model has %d" % (len(index), len(params)))
        for key in ("lr", "momentum", "max_norm"):
            if key in groups[0]:
                self.param_groups[0][key] = groups[0][key]
        state = state_dict.get("state", {})
        flat = self.model.flat_parameters()
        momentum = torch.zeros_like(flat)
        loaded = 0
        for pos, (p, off) in enumerate(zip(params, offsets)):
            entry = state.get(index[pos], state.get(str(index[pos])))
            buf = None if entry is None else entry.get("momentum_buffer")
            if buf is None:
                continue
            if buf.numel() != p.numel():
                raise ValueError("momentum buffer %d has %d elements, parameter has %d" % (pos, buf.numel(), p.numel()))
            momentum[off:off + p.numel()].copy_(buf.reshape(-1).to(device=flat.device, dtype=torch.float32))
            loaded += 1
        if loaded not in (0, len(params)):
            raise ValueError("optimizer state has momentum for %d of %d parameters" % (loaded, len(params)))
        self._momentum = momentum
        self._steps = 1 if loaded else 0          # 0: the next step initialises momentum from the gradient, as torch does
        self._norm = torch.zeros(2, dtype=torch.float64, device=flat.device)
