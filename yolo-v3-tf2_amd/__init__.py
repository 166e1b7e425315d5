"""MI355X-native YOLOv3 inference hot path (Darknet-53 + 3 heads -> decode -> NMS).

Host-side mirror of the reference's Python plugin surface over a C-ABI HIP library.
Import as `yolo_v3_tf2_amd` (see the shim at the repo root)."""
import os as _os

PACKAGE_DIR = _os.path.dirname(_os.path.abspath(__file__))
REPO_ROOT = _os.path.dirname(PACKAGE_DIR)
__version__ = "0.1.0"
