"""Counterpart of the reference's evaluation driver (reference: evaluate_yolov3.py:31-237; SURVEY.md 8f "n4").

TFRecord dataset -> batches through the GPU detect path -> per-image gather -> `EvaluateDetections` counters ->
recall / precision per NMS score threshold.  Same function names and argument meaning as the reference; NumPy arrays
where it has tf tensors, Python lists where it has ragged tensors.  The reference script as shipped cannot run (it
imports `decoded_output` but calls `yolo_decode`, evaluate_yolov3.py:23,109); the intended behaviour is built here.
"""
from __future__ import annotations

import numpy as np
import yaml

from .core.load_tfrecords import parse_tfrecords
from .core.parse_model import Input, ParseModel
from .core.utils import get_anchors, resize_image
from .evaluate_detections import EvaluateDetections
from .inference import DetectModel


def arrange_predict_output(batch_bboxes_padded, batch_class_indices_padded, batch_scores_padded,
                           batch_selected_indices_padded, batch_num_valid_detections, batch_gt_y):
    """-> (bboxes_batch, classes_batch, gt_bboxes_batch, gt_classes_batch); the first two are per-image lists
    (ragged), the last two stacked arrays (reference: evaluate_yolov3.py:31-82)."""
    bboxes_batch, classes_batch, gt_bboxes_batch, gt_classes_batch = [], [], [], []
    for bb, cc, ss, sel, nv, gt_y in zip(batch_bboxes_padded, batch_class_indices_padded, batch_scores_padded,
                                         batch_selected_indices_padded, batch_num_valid_detections, batch_gt_y):
        b, c, _ = EvaluateDetections.gather_nms_output(bb, cc, ss, sel, nv)
        bboxes_batch.append(b)
        classes_batch.append(c)
        gt_y = np.asarray(gt_y, np.float32)
        gt_bboxes_batch.append(gt_y[..., 0:4])                     # tf.split(gt_y, [4, 1, 1], axis=-1)
        gt_classes_batch.append(gt_y[..., 5].astype(np.int32))
    return bboxes_batch, classes_batch, np.stack(gt_bboxes_batch), np.stack(gt_classes_batch)


def prepare_dataset(tfrecords_dir, batch_size, image_size, yolo_max_boxes, classes_name_file):
    """reference: evaluate_yolov3.py:85-94.  Rows whose objectness column is 1 are kept (drops the zero padding), so
    -- as in the reference -- a batch only stacks when its images hold equally many boxes."""
    dataset = parse_tfrecords(tfrecords_dir, image_size=image_size, max_bboxes=yolo_max_boxes,
                              class_file=classes_name_file)
    dataset = dataset.map(lambda x, y: (x, y[y[..., 4] == 1]))
    dataset = dataset.batch(batch_size)
    dataset = dataset.map(lambda img, y: (resize_image(img, image_size, image_size), y))
    return dataset


def create_model(model_config_file, nclasses, anchors_table, nms_score_threshold, nms_iou_threshold, yolo_max_boxes,
                 input_weights_path, weights=None):
    """reference: evaluate_yolov3.py:97-116 -> callable with .predict(batch) returning the NMS 5-tuple"""
    with open(model_config_file, "r") as _stream:
        model_config = yaml.safe_load(_stream)
    inputs = Input(shape=(None, None, 3))
    model = ParseModel().build_model(inputs, nclasses=nclasses, **model_config)
    if weights is not None:
        model.set_weights_dict(weights)
    else:
        model.load_weights(input_weights_path).expect_partial()
    print("weights loaded")
    return DetectModel(model, anchors_table, nclasses, yolo_max_boxes, nms_iou_threshold, nms_score_threshold)


def calc_recal_precision(counters):
    """reference: evaluate_yolov3.py:119-125 -- per-class vectors, 1e-20 in the denominators"""
    tp, fp, fn = (np.asarray(counters[k], np.float32) for k in ("tp", "fp", "fn"))
    recall = tp / (tp + fn + np.float32(1e-20))
    precision = tp / (tp + fp + np.float32(1e-20))
    print(f"recall: {recall}, precision: {precision}")
    return recall, precision


def evaluate(detect_config, evaluate_nms_score_thresholds, evaluate_iou_threshold=0.5, max_batches=20, weights=None,
             one_class=True):
    """The loop of reference evaluate_yolov3.py:153-232: for every score threshold build the detect model, run the
    first `max_batches` dataset batches (`dataset.take(20)` there), count, and report (recall, precision).
    Returns [(threshold, recall, precision, counters, counters_oneclass)]."""
    anchors_table = np.asarray(get_anchors(detect_config["anchors_file"]), np.float32)
    class_names = [c.strip() for c in open(detect_config["classes_name_file"]).readlines()]
    nclasses = len(class_names)
    dataset = prepare_dataset(detect_config["tfrecords_dir"], detect_config["batch_size"], detect_config["image_size"],
                              detect_config["yolo_max_boxes"], detect_config["classes_name_file"])
    results = []
    for nms_score_threshold in evaluate_nms_score_thresholds:
        model = create_model(detect_config["model_config_file"], nclasses, anchors_table, nms_score_threshold,
                             detect_config["nms_iou_threshold"], detect_config["yolo_max_boxes"],
                             detect_config.get("input_weights_path"), weights)
        eval_detections = EvaluateDetections(nclasses, evaluate_iou_threshold)
        eval_detections_oneclass = EvaluateDetections(nclasses, evaluate_iou_threshold)
        for n, (batch_images, batch_gt_y) in enumerate(dataset):
            if max_batches is not None and n >= max_batches:
                break
            out = model.predict(batch_images)
            bboxes_batch, classes_batch, gt_bboxes_batch, gt_classes_batch = arrange_predict_output(*out, batch_gt_y)
            for pred_bboxes, pred_classes, gt_bboxes, gt_classes in zip(bboxes_batch, classes_batch, gt_bboxes_batch,
                                                                        gt_classes_batch):
                eval_detections.evaluate(pred_bboxes, pred_classes, gt_bboxes, gt_classes)
                if one_class:   # boxes only: every class id forced to 0
                    eval_detections_oneclass.evaluate(pred_bboxes, np.zeros_like(pred_classes), gt_bboxes,
                                                      np.zeros_like(gt_classes))
        recall, precision = calc_recal_precision(eval_detections.counters)
        results.append((nms_score_threshold, recall, precision, eval_detections.counters,
                        eval_detections_oneclass.counters))
    return results


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config/detect_config.yaml")
    ap.add_argument("--evaluate-config", default="config/evaluate_config.yaml")
    a = ap.parse_args(argv)
    with open(a.evaluate_config) as s:
        thresholds = yaml.safe_load(s)["evaluate_nms_score_thresholds"]
    with open(a.config) as s:
        detect_config = yaml.safe_load(s)
    print([(t, float(r.mean()), float(p.mean())) for t, r, p, _, _ in evaluate(detect_config, thresholds)])


if __name__ == "__main__":
    main()
