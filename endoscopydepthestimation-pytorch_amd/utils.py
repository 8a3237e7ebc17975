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


def save_model(model, optimizer, epoch, step, model_path, validation_loss, module_prefix=True):
    """reference utils.py:674-682 wire format {model, optimizer, epoch, step, validation}; keys carry
    the 'module.' prefix the reference's DataParallel wrapper adds (train.py:197)."""
    state = model.state_dict()
    if module_prefix:
        state = {"module." + k: v for k, v in state.items()}
    torch.save({'model': state, 'optimizer': optimizer.state_dict(), 'epoch': epoch, 'step': step,
                'validation': validation_loss}, str(model_path))


def load_model_state(model, state):
    """Load a reference checkpoint's ``state['model']`` with or without the 'module.' prefix."""
    cleaned = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
    return model.load_state_dict(cleaned)
