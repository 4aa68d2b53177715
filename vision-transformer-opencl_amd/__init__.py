"""MI355X-native ViT inference hot path (drop-in for the reference's ViT_opencl path).

The directory name is not a Python identifier; load it with
``importlib.import_module("vision-transformer-opencl_amd")`` (tests/conftest.py and
__graft_entry__.py register it under the alias ``vit_amd``).
"""
from . import dp, launch, synth  # noqa: F401
from .synth import ModelConfig, VIT_B16, VIT_L16_384, VIT_SMALL, VIT_TINY  # noqa: F401
