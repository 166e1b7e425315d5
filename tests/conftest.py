import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def program():
    import yolo_v3_tf2_amd  # noqa: F401
    from yolo_v3_tf2_amd.graph import load_program
    return load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)


@pytest.fixture(scope="session")
def weights(program):
    from yolo_v3_tf2_amd.weights import synthetic_weights
    return synthetic_weights(program, seed=4321)


@pytest.fixture(scope="session")
def anchors():
    from yolo_v3_tf2_amd.core.utils import get_anchors
    return get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype("float32")
