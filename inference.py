#!/usr/bin/env python3
"""`python inference.py --config config/detect_config_coco.yaml` -- same entry point as the reference's
inference.py, running the MI355X-native path (see yolo-v3-tf2_amd/inference.py)."""
import yolo_v3_tf2_amd.inference as _inf

if __name__ == "__main__":
    _inf.main()
