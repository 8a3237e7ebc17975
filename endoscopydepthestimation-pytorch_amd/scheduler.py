"""CyclicLR with the interface of reference ``scheduler.py:16-161`` (the 'triangular' policy is what
train.py:203 uses; 'triangular2' and 'exp_range' are kept for API parity).  Host-side only."""

import math

from torch.optim import Optimizer


class CyclicLR(object):
    def __init__(self, optimizer, base_lr=1e-3, max_lr=6e-3, step_size=2000, mode='triangular', gamma=1.,
                 scale_fn=None, scale_mode='cycle', last_batch_iteration=-1):
        if not isinstance(optimizer, Optimizer):
            raise TypeError('{} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer = optimizer
        groups = len(optimizer.param_groups)

        def per_group(value, name):
            if isinstance(value, (list, tuple)):
                if len(value) != groups:
                    raise ValueError("expected {} {}, got {}".format(groups, name, len(value)))
                return list(value)
            return [value] * groups

        self.base_lrs = per_group(base_lr, "base_lr")
        self.max_lrs = per_group(max_lr, "max_lr")
        self.step_size = step_size
        if mode not in ('triangular', 'triangular2', 'exp_range') and scale_fn is None:
            raise ValueError('mode is invalid and scale_fn is None')
        self.mode, self.gamma = mode, gamma
        if scale_fn is None:
            self.scale_fn = {'triangular': lambda x: 1.,
                             'triangular2': lambda x: 1 / (2. ** (x - 1)),
                             'exp_range': lambda x: self.gamma ** x}[mode]
            self.scale_mode = 'iterations' if mode == 'exp_range' else 'cycle'
        else:
            self.scale_fn, self.scale_mode = scale_fn, scale_mode
        self.batch_step(last_batch_iteration + 1)
        self.last_batch_iteration = last_batch_iteration

    def batch_step(self, batch_iteration=None):
        if batch_iteration is None:
            batch_iteration = self.last_batch_iteration + 1
        self.last_batch_iteration = batch_iteration
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr

    def get_lr(self):
        size = float(self.step_size)
        cycle = math.floor(1 + self.last_batch_iteration / (2 * size))
        x = abs(self.last_batch_iteration / size - 2 * cycle + 1)
        arg = cycle if self.scale_mode == 'cycle' else self.last_batch_iteration
        return [base + (peak - base) * max(0., 1 - x) * self.scale_fn(arg)
                for base, peak in zip(self.base_lrs, self.max_lrs)]
