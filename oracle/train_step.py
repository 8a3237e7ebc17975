"""Oracle (test infrastructure): one training iteration, CPU torch fp32.

Restates the batch loop body of reference train.py:272-328:
  mask colours by the boundary, two network forwards (BN statistics per call), depth scaling,
  flow-from-depth both ways, masking by the boundary, sparse-flow loss, depth warping both ways,
  depth-consistency loss, weighted sum, non-finite guard, backward, clip_grad_norm_(10), SGD(0.9).
"""

import math

import torch

from . import geometry, losses, network, schedule

BATCH_KEYS = ("colors_1", "colors_2", "sparse_depths_1", "sparse_depths_2",
              "sparse_depth_masks_1", "sparse_depth_masks_2", "sparse_flows_1", "sparse_flows_2",
              "sparse_flow_masks_1", "sparse_flow_masks_2", "boundaries",
              "rotations_1_wrt_2", "rotations_2_wrt_1", "translations_1_wrt_2",
              "translations_2_wrt_1", "intrinsics")


def losses_from_depths(pred_1, pred_2, batch, sfl_weight=20.0, dcl_weight=0.1, epsilon=1.0e-8):
    """train.py:279-315 given the two predicted depth maps.  Returns (loss, dcl, sfl, extras)."""
    b = batch["boundaries"]
    scaled_1, std_1 = geometry.depth_scaling(pred_1, batch["sparse_depths_1"],
                                             batch["sparse_depth_masks_1"], epsilon)
    scaled_2, std_2 = geometry.depth_scaling(pred_2, batch["sparse_depths_2"],
                                             batch["sparse_depth_masks_2"], epsilon)
    raw_1 = geometry.flow_from_depth(scaled_1, b, batch["translations_1_wrt_2"],
                                     batch["rotations_1_wrt_2"], batch["intrinsics"])
    raw_2 = geometry.flow_from_depth(scaled_2, b, batch["translations_2_wrt_1"],
                                     batch["rotations_2_wrt_1"], batch["intrinsics"])
    flow_1, flow_2 = raw_1 * b, raw_2 * b
    sfl = sfl_weight * 0.5 * (
        losses.sparse_masked_l1(batch["sparse_flows_1"] * b, flow_1, batch["sparse_flow_masks_1"] * b) +
        losses.sparse_masked_l1(batch["sparse_flows_2"] * b, flow_2, batch["sparse_flow_masks_2"] * b))
    warped_21, inter_1 = geometry.depth_warping(scaled_1, scaled_2, b, batch["translations_1_wrt_2"],
                                                batch["rotations_1_wrt_2"], batch["intrinsics"], epsilon)
    warped_12, inter_2 = geometry.depth_warping(scaled_2, scaled_1, b, batch["translations_2_wrt_1"],
                                                batch["rotations_2_wrt_1"], batch["intrinsics"], epsilon)
    dcl = dcl_weight * 0.5 * (
        losses.normalized_distance(scaled_1, warped_21, inter_1, batch["intrinsics"]) +
        losses.normalized_distance(scaled_2, warped_12, inter_2, batch["intrinsics"]))
    extras = {"scaled_1": scaled_1, "scaled_2": scaled_2, "flow_1": raw_1, "flow_2": raw_2,
              "masked_flow_1": flow_1, "masked_flow_2": flow_2,
              "warped_21": warped_21, "warped_12": warped_12, "inter_1": inter_1, "inter_2": inter_2,
              "std_1": std_1, "std_2": std_2}
    return dcl + sfl, dcl, sfl, extras


def forward_backward(state, batch, sfl_weight=20.0, dcl_weight=0.1, epsilon=1.0e-8):
    """Forward + backward of one iteration.  ``state`` is the network dict of ``oracle.network``;
    trainable entries become autograd leaves.  Returns dict(loss, dcl, sfl, pred_1, pred_2, grads)."""
    names = network.trainable_names()
    for n in names:
        state[n] = state[n].detach().requires_grad_(True)
    b = batch["boundaries"]
    pred_1 = network.forward(state, b * batch["colors_1"], training=True)
    pred_2 = network.forward(state, b * batch["colors_2"], training=True)
    loss, dcl, sfl, extras = losses_from_depths(pred_1, pred_2, batch, sfl_weight, dcl_weight, epsilon)
    grads = torch.autograd.grad(loss, [state[n] for n in names], allow_unused=True)
    for n in names:
        state[n] = state[n].detach()
    return {"loss": loss.detach(), "dcl": dcl.detach(), "sfl": sfl.detach(),
            "pred_1": pred_1.detach(), "pred_2": pred_2.detach(),
            "grads": dict(zip(names, grads)), "extras": {k: v.detach() for k, v in extras.items()}}


def train_iteration(state, momentum, batch, lr, sfl_weight=20.0, dcl_weight=0.1, epsilon=1.0e-8):
    """Full iteration including the non-finite guard and the optimizer (train.py:317-328).

    In the non-finite branch torch >= 2.0 (zero_grad(set_to_none=True)) makes ``optimizer.step()``
    a no-op, so nothing is updated.  ``momentum`` is {name: tensor or None}.
    """
    out = forward_backward(state, batch, sfl_weight, dcl_weight, epsilon)
    value = float(out["loss"])
    if math.isnan(value) or math.isinf(value):
        out["skipped"] = True
        return out
    names = network.trainable_names()
    params = [state[n] for n in names]
    grads = [out["grads"][n].clone() for n in names]
    bufs = [momentum.get(n) for n in names]
    out["grad_norm"] = schedule.clip_and_sgd(params, grads, bufs, lr)
    for n, buf in zip(names, bufs):
        momentum[n] = buf
    out["skipped"] = False
    return out
