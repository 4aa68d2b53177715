import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the package directory name is not an identifier: import by string and alias it
vit_amd = importlib.import_module("vision-transformer-opencl_amd")
sys.modules.setdefault("vit_amd", vit_amd)
binding = importlib.import_module("vision-transformer-opencl_amd.binding")
sys.modules.setdefault("vit_amd.binding", binding)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import subprocess
    if not os.path.exists(binding.LIB_PATH):
        subprocess.run(["make", "-C", os.path.dirname(binding.LIB_PATH), "-j4"], check=True)
    from oracle import pyoracle
    if not os.path.exists(pyoracle.LIB_PATH):
        pyoracle.build(ref=True)


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


def oracle_config(cfg):
    from oracle import pyoracle as po
    return po.Config(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.num_classes, cfg.embed_dim,
                     cfg.depth, cfg.num_heads, cfg.hidden_dim)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as po
    po.set_threads(min(16, os.cpu_count() or 1))
    return po
