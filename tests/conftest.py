import importlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def load_pair(golden, tag):
    """returns (rgb planar f32 [3,R,C], W f32 [R,C]) for a committed fixture pair"""
    info = golden[tag]["files"]
    R, C = info["rows"], info["cols"]
    # binary PPM (P6): header, then interleaved RGB u8
    rgb = np.fromfile(os.path.join(GOLDEN, info["rgb"]), np.uint8)[-R * C * 3:].reshape(R, C, 3)
    W = np.fromfile(os.path.join(GOLDEN, info["w"]), np.float32).reshape(R, C)
    return np.ascontiguousarray(rgb.transpose(2, 0, 1)).astype(np.float32), W


@pytest.fixture(scope="session")
def pair512(golden):
    return load_pair(golden, "512")


@pytest.fixture(scope="session")
def pair_crop(golden):
    return load_pair(golden, "720p_crop")


@pytest.fixture(scope="session")
def wm():
    """the product package (directory name has a hyphen, hence importlib)"""
    return importlib.import_module("watermarking-gpu_amd")
