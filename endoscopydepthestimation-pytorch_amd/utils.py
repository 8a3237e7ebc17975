"""Host-side helpers of the training path with the reference's names (reference utils.py:615-682)."""

import torch


def kaiming_weight_zero_bias(model, mode="fan_in", activation_mode="relu", distribution="uniform"):
    """reference utils.py:655-671: Kaiming init of every non-BN weight, BN weight = 1, biases 0."""
    if activation_mode == "leaky_relu":
        raise ValueError("Leaky relu is not supported yet")
    with torch.no_grad():
        for module in model.modules():
            weight = getattr(module, "weight", None)
            if isinstance(weight, torch.Tensor):
                if 'BatchNorm' not in module.__class__.__name__:
                    init = torch.nn.init.kaiming_uniform_ if distribution == "uniform" else torch.nn.init.kaiming_normal_
                    init(weight, mode=mode, nonlinearity=activation_mode)
                else:
                    weight.fill_(1)
            bias = getattr(module, "bias", None)
            if isinstance(bias, torch.Tensor):
                bias.zero_()


def init_net(net, type="kaiming", mode="fan_in", activation_mode="relu", distribution="normal"):
    """reference utils.py:619-626 (called train.py:193): move to the GPU and initialise."""
    if not torch.cuda.is_available():
        raise RuntimeError("init_net needs a HIP device (reference utils.py:620 asserts the same)")
    net = net.cuda()
    if type != "kaiming":
        raise ValueError("only the Kaiming initialisation used by train.py is provided")
    kaiming_weight_zero_bias(net, mode=mode, activation_mode=activation_mode, distribution=distribution)
    return net


def generating_pos_and_increment(idx, visible_view_indexes, adjacent_range, rng=None):
    """reference utils.py:412-438 (called dataset.py:346-350): position of the first frame of a training pair inside the
    sequence's visible views and the signed gap to its partner, drawn from ``adjacent_range`` = (min gap, max gap) -- the
    "adjacent range 5-30" of BASELINE.json configs[4].  Host-side pair selection: it consumes Python's ``random`` exactly as the
    reference does (same calls in the same order), so a seeded run picks the same pairs.  ``rng``: a ``random.Random`` to draw
    from instead of the module-level generator (same calls; lets an iterator own its seed -- dataset.TrainingBatches)."""
    import random
    if rng is not None:
        random = rng
    visible_view_idx = idx % len(visible_view_indexes)
    low, high = adjacent_range[0], adjacent_range[1]
    count = len(visible_view_indexes)
    if count <= 2 * low:
        low = count // 2
    if visible_view_idx <= low - 1:
        increment = random.randint(low, min(high, count - 1 - visible_view_idx))
    elif visible_view_idx >= count - low:
        increment = -random.randint(low, min(high, visible_view_idx))
    else:
        if random.randint(0, 1) == 1:
            increment = random.randint(low, min(high, count - 1 - visible_view_idx))
        else:
            increment = -random.randint(low, min(high, visible_view_idx))
    return [visible_view_idx, increment]


def save_model(model, optimizer, epoch, step, model_path, validation_loss, module_prefix=True):
    """reference utils.py:674-682 wire format {model, optimizer, epoch, step, validation}; keys carry
    the 'module.' prefix the reference's DataParallel wrapper adds (train.py:197)."""
    state = model.state_dict()
    if module_prefix:
        state = {"module." + k: v for k, v in state.items()}
    torch.save({'model': state, 'optimizer': optimizer.state_dict(), 'epoch': epoch, 'step': step,
                'validation': validation_loss}, str(model_path))


def load_checkpoint(model_path, model=None, optimizer=None, map_location=None):
    """Read a checkpoint file in the reference's wire format (utils.py:674-682) and, when given, restore ``model`` (keys with or
    without 'module.', train.py:222 / evaluate.py:150) and ``optimizer`` (torch.optim.SGD's layout: FusedClipSGD or SGD).  Returns
    the dictionary {model, optimizer, epoch, step, validation}.  The file is a pickle and is read as one (the param group's lr is a
    numpy scalar once scheduler.CyclicLR has stepped, which torch's weights-only loader refuses): load files you trust, exactly as
    the reference's own ``torch.load(path)`` (train.py:218) assumes."""
    state = torch.load(str(model_path), map_location=map_location, weights_only=False)
    if model is not None:
        load_model_state(model, state["model"])
    if optimizer is not None:
        optimizer.load_state_dict(state["optimizer"])
    return state


def load_model_state(model, state):
    """Load a reference checkpoint's ``state['model']`` with or without the 'module.' prefix."""
    cleaned = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
    return model.load_state_dict(cleaned)


def point_cloud_from_depth(depth_map, color_img, mask_img, intrinsic_matrix, point_cloud_downsampling,
                           min_threshold=None, max_threshold=None, device="cuda"):
    """Drop-in for reference utils.py:823-852 on the GPU (endo_point_cloud): (P, 6) float32 numpy array of
    (x, y, z, r, g, b), row-major over the kept pixels.  The reference walks 81 920 pixels in a Python double loop per
    frame (evaluate.py:272,340); this is three small kernels.  Inputs may be numpy arrays or tensors (any device);
    there is no CPU fallback."""
    import numpy as np
    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("point_cloud_from_depth needs a GPU device: the MI355X path has no CPU fallback")
    depth = torch.as_tensor(np.asarray(depth_map.detach().cpu() if torch.is_tensor(depth_map) else depth_map), dtype=torch.float32)
    height, width = int(depth.shape[0]), int(depth.shape[1])
    depth = depth.contiguous().to(dev)
    color = torch.as_tensor(np.ascontiguousarray(np.asarray(color_img.detach().cpu() if torch.is_tensor(color_img) else color_img)
                                                 .reshape(height, width, 3)).astype(np.uint8)).to(dev)
    mask = torch.as_tensor(np.asarray(mask_img.detach().cpu() if torch.is_tensor(mask_img) else mask_img), dtype=torch.float32)
    mask = mask.reshape(height, width).contiguous().to(dev)
    k = torch.as_tensor(np.asarray(intrinsic_matrix.detach().cpu() if torch.is_tensor(intrinsic_matrix) else intrinsic_matrix),
                        dtype=torch.float32).reshape(3, 3).contiguous().to(dev)
    use_thr = max_threshold is not None and min_threshold is not None
    points = torch.empty((height * width, 6), dtype=torch.float32, device=dev)
    offsets = torch.empty(height + 1, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.endo_point_cloud(_lib.ptr(depth), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(k), height, width,
                                  int(point_cloud_downsampling), 1 if use_thr else 0,
                                  float(min_threshold) if use_thr else 0.0, float(max_threshold) if use_thr else 0.0,
                                  _lib.ptr(offsets), _lib.ptr(points), _lib.ptr(count), _lib.stream())
    _lib.check(rc, "endo_point_cloud")
    n = int(count.item())
    return points[:n].cpu().numpy().reshape(-1, 6)
