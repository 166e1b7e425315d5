"""Host mirror of reference core/yolo_nms.py -- same name and signature, HIP underneath."""
from ..runtime import class_scores as _class_scores, nms_padded as _nms_padded


def yolo_nms(outputs, yolo_max_boxes, nms_iou_threshold, nms_score_threshold):
    """outputs = (bboxes [B,N,4], confidence [B,N,1], class_probs [B,N,nc]) CUDA tensors.
    Returns (bboxes [B,N,4] f32, class_indices [B,N] i64, scores [B,N] f32 (unfiltered),
    selected_indices_padded [B,max] i32, num_valid_detections [B] i32) -- reference: core/yolo_nms.py:16-34.
    Class-agnostic NMS with TF's `non_max_suppression_padded` semantics (IoU >= threshold suppresses,
    score > threshold passes, ties by lower index)."""
    from .yolo_decode_layer import _to_device
    bboxes, confidence, class_probs = (_to_device(t) for t in outputs)
    class_indices, scores = _class_scores(confidence.contiguous(), class_probs.contiguous())
    bboxes = bboxes.reshape(bboxes.shape[0], -1, 4).contiguous()
    selected_indices_padded, num_valid_detections = _nms_padded(bboxes, scores, yolo_max_boxes, nms_iou_threshold,
                                                                nms_score_threshold)
    return (bboxes, class_indices, scores, selected_indices_padded, num_valid_detections)
