"""Host-side mirror of the reference's Python surface (no GPU)."""
import inspect
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_anchors_layout():
    from yolo_v3_tf2_amd.core.utils import get_anchors
    a = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt"))
    assert a.shape == (3, 3, 2) and a.dtype == np.float64
    # row 0 = largest anchors (116x90, 156x198, 373x326)/416 -> coarsest grid (reference: core/utils.py:31-37)
    assert np.allclose(a[0] * 416, [[116, 90], [156, 198], [373, 326]], atol=1e-4)
    assert np.allclose(a[2] * 416, [[10, 13], [16, 30], [33, 23]], atol=1e-4)


def test_signatures_match_reference_surface():
    from yolo_v3_tf2_amd.core.parse_model import ParseModel
    from yolo_v3_tf2_amd.core.yolo_decode_layer import yolo_decode
    from yolo_v3_tf2_amd.core.yolo_nms import yolo_nms
    from yolo_v3_tf2_amd.core.yolo_nms_layer import YoloNmsLayer
    from yolo_v3_tf2_amd.inference import Inference
    assert list(inspect.signature(ParseModel.build_model).parameters)[:6] == [
        "self", "model_inputs", "sub_models_configs", "output_stage", "decay_factor", "nclasses"]
    assert list(inspect.signature(yolo_decode).parameters) == ["model_output_grids", "anchors_table", "nclasses"]
    assert list(inspect.signature(yolo_nms).parameters) == ["outputs", "yolo_max_boxes", "nms_iou_threshold",
                                                            "nms_score_threshold"]
    layer = YoloNmsLayer(100, 0.5, 0.1, name="nms")
    assert (layer.yolo_max_boxes, layer.nms_iou_threshold, layer.nms_score_threshold) == (100, 0.5, 0.1)
    keys = ["model_config_file", "classes_name_file", "anchors_file", "input_weights_path", "image_size",
            "input_data_source", "images_dir", "tfrecords_dir", "batch_size", "image_file_path", "output_dir",
            "yolo_max_boxes", "nms_iou_threshold", "nms_score_threshold", "bbox_color", "font_size",
            "display_result_images", "save_model_path"]
    assert list(inspect.signature(Inference.__call__).parameters)[1:19] == keys


def test_detect_config_has_the_reference_keys():
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config/detect_config_coco.yaml")))
    from yolo_v3_tf2_amd.inference import Inference
    params = set(inspect.signature(Inference.__call__).parameters) - {"self", "weights"}
    assert set(cfg) == params


def test_model_builds_and_summarises_without_gpu():
    from yolo_v3_tf2_amd.core.parse_model import ParseModel
    m = ParseModel().create_model(80, os.path.join(ROOT, "config/models/yolov3/model.yaml"))
    lines = []
    m.summary(print_fn=lines.append)
    assert any("62,001,757" in l for l in lines) and len(lines) == 78
    with pytest.raises(RuntimeError, match="no weights"):
        m(np.zeros((1, 416, 416, 3), np.float32))
    with pytest.raises(FileNotFoundError):
        m.load_weights("does/not/exist.safetensors")


def test_resize_bilinear_half_pixel_centres():
    """tf.image.resize default (bilinear, antialias=False): src = (dst+0.5)*in/out-0.5, edge clamped."""
    from yolo_v3_tf2_amd.core.utils import resize_bilinear
    img = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
    assert np.array_equal(resize_bilinear(img, 4, 4), img)                    # identity
    up = resize_bilinear(img, 8, 8)
    # closed form for a linear ramp: interior values follow the ramp, borders clamp
    xs = np.clip((np.arange(8) + 0.5) * 0.5 - 0.5, 0, 3)
    ref = (xs[:, None] * 4 + xs[None, :]).astype(np.float32)[..., None]
    assert np.allclose(up, ref, atol=1e-6)
    down = resize_bilinear(img, 2, 2)                                          # taps at 0.5 and 2.5
    assert np.allclose(down[..., 0], [[2.5, 4.5], [10.5, 12.5]])


def test_gather_valid_and_detect_line_format():
    from yolo_v3_tf2_amd.inference import Inference
    bb = np.arange(40, dtype=np.float32).reshape(10, 4) / 40
    cc = np.arange(10, dtype=np.int64)
    ss = np.linspace(0.9, 0.1, 10).astype(np.float32)
    sel = np.array([7, 2, 5, 0, 0], np.int32)
    b, c, s = Inference.gather_valid_detections_results(bb, cc, ss, sel, 3)
    assert np.array_equal(b, bb[[7, 2, 5]]) and c.tolist() == [7, 2, 5] and np.array_equal(s, ss[[7, 2, 5]])
    img = np.zeros((20, 40, 3), np.float32)
    pil, det = Inference.annotate(img, b, ["a", "b", "c"], s, 15)
    assert pil.size == (40, 20) and len(det) == 3
    label, xmin, ymin, xmax, ymax = det[0]
    assert label == "a: {}%".format(int(100 * s[0])) and np.isclose(xmin, b[0, 0] * 40) and np.isclose(ymax, b[0, 3] * 20)
    assert str(det).startswith("[('a: ")          # the detect.txt payload is str(list of tuples)


def test_shard_range_covers_batch():
    from yolo_v3_tf2_amd.parallel import shard_range
    for n in (0, 1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - s for s, e in r) - min(e - s for s, e in r) <= 1


def test_evaluate_detections_counters():
    from yolo_v3_tf2_amd.evaluate_detections import EvaluateDetections
    ev = EvaluateDetections(3, 0.5)
    gt_b = np.array([[0.1, 0.1, 0.4, 0.4], [0.5, 0.5, 0.9, 0.9]], np.float32)
    gt_c = np.array([0, 2])
    pr_b = np.array([[0.1, 0.1, 0.4, 0.41],      # TP for gt 0
                     [0.11, 0.1, 0.4, 0.4],      # same gt again: the reference's vectorised check also counts it as TP
                     [0.5, 0.5, 0.9, 0.9],       # right box, wrong class -> FP
                     [0.0, 0.6, 0.1, 0.7]], np.float32)   # no overlap -> FP
    pr_c = np.array([0, 0, 1, 2])
    c = ev.evaluate(pr_b, pr_c, gt_b, gt_c)
    assert c["tp"].tolist() == [2, 0, 0] and c["fp"].tolist() == [0, 1, 1]
    assert c["fn"].tolist() == [0, 0, 1] and c["gts"].tolist() == [1, 0, 1] and c["preds"].tolist() == [2, 1, 1]
    assert c["examples"] == 1
    ev.evaluate(pr_b, pr_c, gt_b, np.array([0, -1]))
    assert ev.counters["errors"] == 1 and ev.counters["examples"] == 1
    r, p_ = ev.recall_precision()
    assert abs(r - 2 / 3) < 1e-9 and abs(p_ - 0.5) < 1e-9
    b, cl, s_ = EvaluateDetections.gather_nms_output(pr_b, pr_c, np.arange(4.0), np.array([3, 1, 0, 0]), 2)
    assert cl.tolist() == [2, 0] and s_.tolist() == [3.0, 1.0]


def test_bench_host_helpers(monkeypatch):
    """bench.py's host-side pieces that need no GPU: the seeded batch every leg shares, and the CPU share used for the
    oracle / PyTorch-CPU legs (affinity capped by the cgroup quota, overridable)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("y3_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a, b = bench.host_images(2, 32, 0), bench.host_images(2, 32, 0)
    assert a.shape == (2, 32, 32, 3) and a.dtype.name == "float32" and (a == b).all() and 0.0 <= a.min() and a.max() < 1.0
    assert not (bench.host_images(2, 32, 1) == a).all()          # every rank draws its own images
    n = bench.host_cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.setenv("Y3_CPU_THREADS", "1")
    assert bench.host_cpu_share() == 1



def test_bench_bf16_parity_gate_passes_the_floor_and_rejects_garbage(program, weights):
    """bench.py's bf16 gate on the CPU: fed the oracle's own bf16 forward with the other summation order (which IS the floor the
    gate measures) it passes with rel == floor; fed zeros, or logits with 5 % noise, it stops the run."""
    import importlib.util
    import os
    import torch
    from oracle import oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("y3_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    x = bench.host_images(1, 64, 0)
    alt = [torch.from_numpy(np.ascontiguousarray(g)) for g in O.forward(program, weights, x, bf16=True, acc64=True)]
    out = bench.parity_gate_bf16(program, weights, x, alt, 1)
    assert out["images"] == 1 and abs(out["rel_l2_vs_bf16_oracle"] - out["floor_rel_l2_oracle_vs_oracle"]) < 1e-12
    assert 1e-3 < out["floor_rel_l2_oracle_vs_oracle"] < 2e-2 and out["rel_l2_vs_fp32_oracle"] < 3e-2
    with pytest.raises(SystemExit, match="PARITY GATE FAILED"):
        bench.parity_gate_bf16(program, weights, x, [torch.zeros_like(g) for g in alt], 1)
    rng = np.random.default_rng(0)
    noisy = [g * torch.from_numpy((1.0 + 0.05 * rng.standard_normal(tuple(g.shape))).astype(np.float32)) for g in alt]
    with pytest.raises(SystemExit, match="PARITY GATE FAILED"):
        bench.parity_gate_bf16(program, weights, x, noisy, 1)


def test_tuning_override_applies_only_to_the_plan_it_names(tmp_path, monkeypatch):
    """Y3_TUNING_FILE used to replace the table of EVERY plan in the process (bench.py's alt measurements re-plan
    the net in f32x3 / f32x2 and would have received fp32 tile ids).  The override now names its mode and geometry."""
    import json
    from yolo_v3_tf2_amd import PACKAGE_DIR
    from yolo_v3_tf2_amd.runtime import tuning_table_path
    packaged = os.path.join(PACKAGE_DIR, "tuning", "f32_b64_s416.json")
    monkeypatch.delenv("Y3_TUNING_FILE", raising=False)
    assert tuning_table_path("f32", 64, 416) == packaged
    f = tmp_path / "t.json"
    f.write_text(json.dumps({"dtype": "f32", "batch": 64, "image_size": 416, "lanes": 1, "tiles": {}}))
    monkeypatch.setenv("Y3_TUNING_FILE", str(f))
    assert tuning_table_path("f32", 64, 416) == str(f)
    assert tuning_table_path("f32x2", 64, 416).endswith("tuning/f32x2_b64_s416.json")      # another mode: packaged table
    assert tuning_table_path("f32", 32, 416).endswith("tuning/f32_b32_s416.json")          # another batch
    assert tuning_table_path("f32", 64, 608).endswith("tuning/f32_b64_s608.json")          # another image size
    f.write_text(json.dumps({"batch": 64, "image_size": 416, "tiles": {}}))                 # no mode named: never applied
    assert tuning_table_path("f32", 64, 416) == packaged
