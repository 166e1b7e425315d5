"""Evaluation consumer of the detect 5-tuple (SURVEY.md 8f "n4") -- host side, NumPy.

Counterpart of reference evaluate_detections.py:16-165 (`EvaluateDetections`): per-class pred / gt / TP / FP / FN
counters from IoU matching of one image's predictions against its ground truth.  Semantics kept, including the
vectorised quirk of the reference: every prediction is matched against the *initial* (empty) assignment vector
(reference: evaluate_detections.py:107-112), so two predictions that pick the same ground-truth box both count as
true positives; the assignment vector only feeds the false-negative count.
"""
from __future__ import annotations

import numpy as np


class EvaluateDetections:
    def __init__(self, nclasses, iou_thresh):
        self.nclasses = nclasses
        self.iou_thresh = iou_thresh
        z = lambda: np.zeros(nclasses, np.int64)
        self.counters = {"preds": z(), "gts": z(), "tp": z(), "fp": z(), "fn": z(), "errors": 0, "examples": 0}

    @staticmethod
    def iou_alg(box_1, box_2):
        """box_1 [4], box_2 [G,4] (xmin,ymin,xmax,ymax) -> [G]; no epsilon in the union (evaluate_detections.py:39-49)"""
        box_1 = np.asarray(box_1, np.float32)[None]
        box_2 = np.asarray(box_2, np.float32)
        ow = np.maximum(np.minimum(box_1[..., 2], box_2[..., 2]) - np.maximum(box_1[..., 0], box_2[..., 0]), 0)
        oh = np.maximum(np.minimum(box_1[..., 3], box_2[..., 3]) - np.maximum(box_1[..., 1], box_2[..., 1]), 0)
        inter = ow * oh
        a1 = (box_1[..., 2] - box_1[..., 0]) * (box_1[..., 3] - box_1[..., 1])
        a2 = (box_2[..., 2] - box_2[..., 0]) * (box_2[..., 3] - box_2[..., 1])
        return inter / (a1 + a2 - inter)

    def evaluate(self, pred_bboxes, pred_classes, gt_bboxes, gt_classes):
        pred_bboxes = np.asarray(pred_bboxes, np.float32).reshape(-1, 4)
        pred_classes = np.asarray(pred_classes).astype(np.int64).reshape(-1)
        gt_bboxes = np.asarray(gt_bboxes, np.float32).reshape(-1, 4)
        gt_classes = np.asarray(gt_classes).astype(np.int64).reshape(-1)
        c = self.counters
        if (gt_classes < 0).any() or (gt_classes >= self.nclasses).any():
            c["errors"] += 1            # reference: skips samples with a bad class id (evaluate_detections.py:66-70)
            return c
        assigned = np.zeros(len(gt_classes), bool)
        if len(pred_classes) and len(gt_classes):
            iou = np.stack([self.iou_alg(p, gt_bboxes) for p in pred_bboxes])      # [P,G]
            best = iou.argmax(axis=1)                                              # first maximum
            max_iou = iou[np.arange(len(best)), best]
            decisions = (max_iou > self.iou_thresh) & (gt_classes[best] == pred_classes) & ~assigned[best]
            np.logical_or.at(assigned, best, decisions)
        else:
            decisions = np.zeros(len(pred_classes), bool)
        np.add.at(c["tp"], pred_classes, decisions.astype(np.int64))
        np.add.at(c["fp"], pred_classes, (~decisions).astype(np.int64))
        np.add.at(c["fn"], gt_classes, (~assigned).astype(np.int64))
        np.add.at(c["gts"], gt_classes, 1)
        np.add.at(c["preds"], pred_classes, 1)
        c["examples"] += 1
        return c

    @staticmethod
    def gather_nms_output(bboxes_padded, class_indices_padded, scores_padded, selected_indices_padded,
                          num_valid_detections):
        """reference: evaluate_detections.py:168-176"""
        sel = np.asarray(selected_indices_padded)[:int(num_valid_detections)]
        return np.asarray(bboxes_padded)[sel], np.asarray(class_indices_padded)[sel], np.asarray(scores_padded)[sel]

    def recall_precision(self):
        tp, fp, fn = (self.counters[k].sum() for k in ("tp", "fp", "fn"))
        return tp / max(tp + fn, 1), tp / max(tp + fp, 1)
