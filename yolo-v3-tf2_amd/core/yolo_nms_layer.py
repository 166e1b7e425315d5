"""Host mirror of reference core/yolo_nms_layer.py (a Keras Layer holding the three NMS hyper-parameters)."""
from .yolo_nms import yolo_nms


class YoloNmsLayer:
    def __init__(self, yolo_max_boxes, nms_iou_threshold, nms_score_threshold, **kwargs):
        # reference: core/yolo_nms_layer.py:18-22 (parameters stored, then Layer.__init__(**kwargs))
        self.yolo_max_boxes = yolo_max_boxes
        self.nms_iou_threshold = nms_iou_threshold
        self.nms_score_threshold = nms_score_threshold
        self.name = kwargs.get("name", "yolo_nms_layer")

    def call(self, decoded_outputs, **kwargs):
        # reference: core/yolo_nms_layer.py:26-29
        return yolo_nms(decoded_outputs, self.yolo_max_boxes, self.nms_iou_threshold, self.nms_score_threshold)

    __call__ = call
