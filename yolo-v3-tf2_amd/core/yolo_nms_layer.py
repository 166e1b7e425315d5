"""`YoloNmsLayer`: the object form of `yolo_nms` -- three NMS limits fixed at construction, applied on call.

Interface of reference core/yolo_nms_layer.py:16-29 (a Keras `Layer` subclass there; it owns no weights, so a plain
callable carries everything the callers at inference.py:114-115 and evaluate_yolov3.py:111-112 use).
"""
from .yolo_nms import yolo_nms


class YoloNmsLayer:
    _LIMITS = ("yolo_max_boxes", "nms_iou_threshold", "nms_score_threshold")

    def __init__(self, yolo_max_boxes, nms_iou_threshold, nms_score_threshold, **kwargs):
        for key, value in zip(self._LIMITS, (yolo_max_boxes, nms_iou_threshold, nms_score_threshold)):
            setattr(self, key, value)
        self.name = kwargs.pop("name", "yolo_nms_layer")   # the only Layer keyword the callers pass

    def call(self, decoded_outputs, **kwargs):
        """(bboxes, confidences, class_probs) -> the NMS 5-tuple"""
        return yolo_nms(decoded_outputs, *(getattr(self, key) for key in self._LIMITS))

    __call__ = call
