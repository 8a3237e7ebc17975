"""``import endo_amd`` -> the package in ``endoscopydepthestimation-pytorch_amd/`` (whose directory
name is not a valid Python identifier)."""

import importlib
import sys

sys.modules[__name__] = importlib.import_module("endoscopydepthestimation-pytorch_amd")
