"""Host mirror of reference core/yolo_decode_layer.py -- same name and signature, HIP underneath."""
from ..runtime import yolo_decode as _decode


def yolo_decode(model_output_grids, anchors_table, nclasses):
    """model_output_grids: 3 x [B,g,g,3,5+nclasses] CUDA tensors; anchors_table [3,3,2] (normalised w,h,
    row s for grid s).  Returns (all_grids_bboxes [B,N,4], all_grids_confidence [B,N,1],
    all_grids_class_probs [B,N,nclasses]) -- reference: core/yolo_decode_layer.py:15-36."""
    return _decode([_to_device(g) for g in model_output_grids], anchors_table, nclasses)


def _to_device(t):
    """Host arrays are accepted like the reference accepts NumPy inputs; they are copied to the GPU."""
    import numpy as np
    import torch
    if isinstance(t, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(t, np.float32)).cuda()
    return t.contiguous()
