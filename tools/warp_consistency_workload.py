"""The second BASELINE metric's call (endo_warp_consistency, batch 8 x 256 x 320) 50 times: a rocprofv3 workload."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
dev = torch.device("cuda:0")
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 256, 320)
batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(n, h, w, seed=0).items()}
d1 = pkg.synthetic.smooth_depth(n, h, w, seed=1).to(dev)
d2 = pkg.synthetic.smooth_depth(n, h, w, seed=2).to(dev)
for _ in range(50):
    with torch.no_grad():
        pkg.losses.warp_consistency(d1, d2, batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"],
                                    batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"], dcl_weight=2.0)
torch.cuda.synchronize()
