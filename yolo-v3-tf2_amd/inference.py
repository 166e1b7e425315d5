"""Counterpart of the reference's only working caller (reference: inference.py:19-201).

Same YAML keys (`config/detect_config_coco.yaml` of the reference works unchanged apart from the weight
file format), same 5-tuple per batch, same per-image gather and the same `detect.txt` line payload
(`str([(label_str, xmin, ymin, xmax, ymax), ...])`, reference: core/render_utils.py:91, inference.py:39-41).
The composite model image -> (bboxes, class_indices, scores, selected_indices_padded, num_valid) runs as
conv program -> fused decode/score -> NMS on the GPU.  Rendering stays on the CPU with Pillow.
"""
from __future__ import annotations

import os

import numpy as np
import yaml

from .core.parse_model import Input, ParseModel
from .core.utils import dir_filelist, get_anchors, load_image_u8


class DetectModel:
    """Model(inputs, nms_output, name="yolo_nms") of reference inference.py:109-117: the composition
    `model(inputs)` -> `yolo_decode(grids, anchors_table, nclasses)` -> `YoloNmsLayer(max, iou, score)(decoded)`
    through the plugin surface (core/yolo_decode_layer.py, core/yolo_nms_layer.py), exactly as the reference wires it.
    fused=True runs the same arithmetic as one decode+score kernel that never materialises the [B,N,nclasses]
    probabilities (what bench.py and y3_net_detect use); both give identical tuples."""

    def __init__(self, model, anchors_table, nclasses, yolo_max_boxes, nms_iou_threshold, nms_score_threshold,
                 fused=False):
        from .core.yolo_nms_layer import YoloNmsLayer
        self.model = model
        self.anchors_table = np.asarray(anchors_table, np.float32)
        self.nclasses = nclasses
        self.yolo_max_boxes = int(yolo_max_boxes)
        self.nms_iou_threshold = float(nms_iou_threshold)
        self.nms_score_threshold = float(nms_score_threshold)
        self.fused = bool(fused)
        self.nms_layer = YoloNmsLayer(self.yolo_max_boxes, self.nms_iou_threshold, self.nms_score_threshold)

    def __call__(self, images):
        """images: [B,S,S,3] fp32 (numpy or CUDA tensor) -> the 5-tuple as CUDA tensors."""
        grids = self.model(images)
        if self.fused:
            from . import runtime
            bboxes, cls, scores = runtime.yolo_decode_scores(grids, self.anchors_table, self.nclasses)
            sel, nv = runtime.nms_padded(bboxes, scores, self.yolo_max_boxes, self.nms_iou_threshold,
                                         self.nms_score_threshold)
            return bboxes, cls, scores, sel, nv
        from .core.yolo_decode_layer import yolo_decode
        decoded_output = yolo_decode(grids, self.anchors_table, self.nclasses)      # reference: inference.py:111
        return self.nms_layer(decoded_output)                                        # reference: inference.py:114-115

    def predict(self, images, **_):
        return tuple(t.cpu().numpy() for t in self(images))


class Inference:

    @staticmethod
    def gather_valid_detections_results(bboxes_padded, class_indices_padded, scores_padded, selected_indices_padded,
                                        num_valid_detections):
        # reference: inference.py:21-28
        sel = selected_indices_padded[:num_valid_detections]
        return bboxes_padded[sel], class_indices_padded[sel], scores_padded[sel]

    @staticmethod
    def _dump_detections_text(image_detections_result, detections_list_outfile):
        detections_list_outfile.write(f"{image_detections_result}\n")
        detections_list_outfile.flush()

    @staticmethod
    def annotate(image, bboxes, classes_names, scores, font_size):
        """Boxes + '<class>: <score>%' labels; returns (PIL image, detections list) with the reference's
        per-detection tuple (label, xmin, ymin, xmax, ymax) in pixels of `image` (core/render_utils.py:71-91)."""
        from PIL import Image, ImageDraw
        pil = Image.fromarray(np.uint8(np.clip(image, 0, 1) * 255)).convert("RGB")
        draw = ImageDraw.Draw(pil)
        w, h = pil.size
        detections = []
        for bbox, name, score in zip(bboxes, classes_names, scores):
            xmin, ymin, xmax, ymax = (float(v) for v in bbox * np.array([w, h, w, h], np.float32))
            label = "{}: {}%".format(name, int(100 * score))
            draw.rectangle([(xmin, ymin), (max(xmax, xmin), max(ymax, ymin))], outline=(255, 255, 255))
            draw.text((max(xmin, 0), max(ymin, 0)), label, fill=(255, 255, 0))
            detections.append((label, xmin, ymin, xmax, ymax))
        return pil, detections

    def build(self, model_config_file, classes_name_file, anchors_file, input_weights_path, yolo_max_boxes,
              nms_iou_threshold, nms_score_threshold, weights=None):
        anchors_table = get_anchors(anchors_file).astype(np.float32)      # reference: inference.py:83
        class_names = [c.strip() for c in open(classes_name_file).readlines()]
        nclasses = len(class_names)
        with open(model_config_file, "r") as f:
            model_config = yaml.safe_load(f)
        from .graph import find_config_root
        model = ParseModel().build_model(Input(shape=(None, None, 3)), model_config["sub_models_configs"],
                                         model_config["output_stage"], nclasses=nclasses,
                                         config_root=find_config_root(model_config_file,
                                                                      model_config["sub_models_configs"]))
        with open("model_inference_summary.txt", "w") as f:
            model.summary(print_fn=lambda x: f.write(x + "\n"))
        if weights is not None:
            model.set_weights_dict(weights)
        else:
            model.load_weights(input_weights_path).expect_partial()
        print("weights loaded")
        return DetectModel(model, anchors_table, nclasses, yolo_max_boxes, nms_iou_threshold,
                           nms_score_threshold), class_names

    def __call__(self, model_config_file, classes_name_file, anchors_file, input_weights_path, image_size,
                 input_data_source, images_dir, tfrecords_dir, batch_size, image_file_path, output_dir, yolo_max_boxes,
                 nms_iou_threshold, nms_score_threshold, bbox_color, font_size, display_result_images=None,
                 save_model_path=None, weights=None):
        os.makedirs(output_dir, exist_ok=True)
        detections_text_list_outfile = f"{output_dir}/detect.txt"
        try:
            os.remove(detections_text_list_outfile)
        except OSError:
            pass
        out = open(detections_text_list_outfile, "a")
        model, class_names = self.build(model_config_file, classes_name_file, anchors_file, input_weights_path,
                                        yolo_max_boxes, nms_iou_threshold, nms_score_threshold, weights)
        results = []
        import torch
        from . import runtime
        if input_data_source == "tfrecords":
            # reference: inference.py:119-144.  Records are parsed on the host (core/load_tfrecords.py); the
            # resize(0..255 values) / 255 of parse_tfrecord_fn runs on the GPU straight into the batch tensor.  The
            # reference's following resize_image(img, S, S) sees an S x S image: scale 1, no padding -- the identity.
            from .core.load_tfrecords import decode_jpeg_u8, iter_examples
            image_index = 0   # running index (the reference numbers files within a batch, overwriting earlier ones)
            pend = []

            def run_batch(images_u8):
                nonlocal image_index
                batch_dev = torch.empty((len(images_u8), image_size, image_size, 3), dtype=torch.float32,
                                        device="cuda")
                for slot, u8 in enumerate(images_u8):
                    runtime.preprocess_image(torch.from_numpy(u8).cuda(), batch_dev, slot, divide_after=True)
                b_boxes, b_cls, b_scores, b_sel, b_nv = (t.cpu().numpy() for t in model(batch_dev))
                for bb, cc, ss, sel, nv, img in zip(b_boxes, b_cls, b_scores, b_sel, b_nv, batch_dev.cpu().numpy()):
                    bboxes, classes, scores = self.gather_valid_detections_results(bb, cc, ss, sel, int(nv))
                    classes_names = [class_names[idx] for idx in classes]
                    annotated, detections = self.annotate(img, bboxes, classes_names, scores, font_size)
                    self._dump_detections_text(detections, out)
                    annotated.save(f"{output_dir}/detect_{image_index}.jpg")
                    results.append((bboxes, classes, scores, classes_names))
                    image_index += 1

            for example in iter_examples(tfrecords_dir):
                pend.append(decode_jpeg_u8(example["image/encoded"][0]))
                if len(pend) == int(batch_size):
                    run_batch(pend)
                    pend = []
            if pend:
                run_batch(pend)
            filenames = []
        elif input_data_source == "image_file":
            filenames = [image_file_path]
        elif input_data_source == "images_dir":
            filenames = dir_filelist(images_dir, (".jpeg", ".jpg", ".png", ".bmp"))
        else:
            filenames = []
        for image_index, file in enumerate(filenames):
            # decode on the host, then uint8 -> [0,1] -> bilinear resize on the GPU straight into the batch tensor
            # (reference: inference.py:157-158 decode_image + tf.image.resize)
            orig_image = load_image_u8(file)
            batch_dev = torch.empty((1, image_size, image_size, 3), dtype=torch.float32, device="cuda")
            runtime.preprocess_image(torch.from_numpy(orig_image).cuda(), batch_dev, 0)
            b_boxes, b_cls, b_scores, b_sel, b_nv = (t.cpu().numpy() for t in model(batch_dev))
            batch = batch_dev.cpu().numpy()
            for bb, cc, ss, sel, nv, img in zip(b_boxes, b_cls, b_scores, b_sel, b_nv, batch):
                bboxes, classes, scores = self.gather_valid_detections_results(bb, cc, ss, sel, int(nv))
                classes_names = [class_names[idx] for idx in classes]
                annotated, detections = self.annotate(img, bboxes, classes_names, scores, font_size)
                annotated = annotated.resize((orig_image.shape[1], orig_image.shape[0]))
                self._dump_detections_text(detections, out)
                annotated.save(f"{output_dir}/detect_{image_index}.jpg")
                results.append((bboxes, classes, scores, classes_names))
        if results:
            bboxes, classes, scores, classes_names = results[-1]
            for class_name, bbox, score in zip(classes_names, bboxes, scores):
                print(f"{class_name} bbox: {bbox} score: {score}")
        out.close()
        return results


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="config/detect_config.yaml", help="yaml config file")
    args = parser.parse_args(argv)
    with open(args.config, "r") as stream:
        detect_config = yaml.safe_load(stream)
    Inference()(**detect_config)


if __name__ == "__main__":
    main()
