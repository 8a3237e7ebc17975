"""Oracle (test infrastructure): host-side schedule pieces of the training step.

  * cyclic_lr     <- reference scheduler.py:146-161 (CyclicLR.get_lr, 'triangular' policy, which is
                     what train.py:203 instantiates and train.py:251 steps with the global step)
  * dcl_weight    <- reference train.py:239-242
  * clip_and_sgd  <- reference train.py:327-328 (clip_grad_norm_(10.0) then SGD(momentum=0.9)),
                     i.e. torch.nn.utils.clip_grad_norm_ + torch.optim.SGD semantics of torch 2.10
"""

import math

import torch


def cyclic_lr(step, base_lr, max_lr, step_size):
    cycle = math.floor(1 + step / (2.0 * step_size))
    x = abs(step / float(step_size) - 2 * cycle + 1)
    return base_lr + (max_lr - base_lr) * max(0.0, 1.0 - x)


def dcl_weight(epoch, configured):
    return 0.1 if epoch <= 20 else configured


def clip_and_sgd(params, grads, momentum_bufs, lr, max_norm=10.0, momentum=0.9):
    """In place on lists of tensors.  Returns the pre-clip global L2 norm.

    clip coefficient = min(1, max_norm / (norm + 1e-6)); buf = momentum * buf + g (buf = g on the
    first step); p -= lr * buf.  No weight decay, no dampening, no nesterov (train.py:202).
    """
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1.0e-6), max=1.0)
    for i, (p, g) in enumerate(zip(params, grads)):
        g.mul_(coef)
        if momentum_bufs[i] is None:
            momentum_bufs[i] = g.clone()
        else:
            momentum_bufs[i].mul_(momentum).add_(g)
        p.sub_(lr * momentum_bufs[i])
    return total
