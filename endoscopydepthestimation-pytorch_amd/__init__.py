"""MI355X-native training hot path of EndoscopyDepthEstimation-Pytorch.

Drop-in ``nn.Module`` replacements (``models``, ``losses``), the reference's LR schedule
(``scheduler``), a fused clip+SGD optimizer (``optim``), one-collective data parallelism
(``distributed``) and the training-iteration glue (``train_step``), all on hand-written HIP kernels
for gfx950 behind the C ABI of ``include/endo_hip.h``.  See DESIGN.md / INTEGRATION.md.

The directory name carries a hyphen, so import it with
``importlib.import_module("endoscopydepthestimation-pytorch_amd")`` or ``import endo_amd``.
"""

from . import _lib, dataset, distributed, losses, models, optim, reader, scatter, scheduler, synthetic, train_step, utils  # noqa: F401
from .models import FCDenseNet57, DepthScalingLayer, DepthWarpingLayer, FlowfromDepthLayer  # noqa: F401
from .losses import SparseMaskedL1Loss, NormalizedDistanceLoss, ScaleInvariantLoss  # noqa: F401
