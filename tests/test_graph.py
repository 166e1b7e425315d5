"""Host logic: model description -> conv program (no GPU)."""
import os

import pytest
import yaml

from yolo_v3_tf2_amd.graph import ConvOp, build_program, eval_int_expr, load_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_totals_match_survey(program):
    # 75 convs, 23 shortcuts, 65.864 / 140.692 GFLOP, 49.032 backbone, 62.0 M params
    assert len(program.conv_nodes) == 75
    assert sum(1 for n in program.nodes if n.kind == "add") == 23
    assert abs(program.flops_per_image(416) / 1e9 - 65.864075) < 1e-5
    assert abs(program.flops_per_image(608) / 1e9 - 140.691900) < 1e-5
    assert abs(program.flops_per_image(416, backbone_only=True) / 1e9 - 49.031610) < 1e-5
    assert program.n_params() == 62001757


def test_lowering_folds_everything(program):
    ops = program.ops
    assert all(isinstance(o, ConvOp) for o in ops) and len(ops) == 75
    assert sum(1 for o in ops if o.residual >= 0) == 23
    cat = [o for o in ops if o.src1 >= 0]
    assert [(o.conv_index, o.cin, o.c0, o.src0_upsample) for o in cat] == [(60, 768, 256, True), (68, 384, 128, True)]
    # heads: 13, 26, 52 grids in model order; bias + linear
    heads = [o for o in ops if o.dst in program.outputs]
    assert [o.conv_index for o in heads] == [58, 66, 74]
    assert all((not o.bn) and (not o.leaky) and o.cout == 255 for o in heads)
    assert [program.tensors[t].div for t in program.outputs] == [32, 16, 8]
    # creation order == Darknet weight order
    assert [n.conv_index for n in program.conv_nodes] == list(range(75))
    subs = [n.sub_model for n in program.conv_nodes]
    assert subs[:52] == ["backbone"] * 52 and subs[52:57] == ["neck0"] * 5 and subs[57:59] == ["head0"] * 2


def test_stride2_convs_and_shapes(program):
    s2 = [n.conv_index for n in program.conv_nodes if n.stride == 2]
    assert s2 == [1, 4, 9, 26, 43]
    t = program.tensors
    outs = {n.conv_index: (t[n.output].channels, t[n.output].div) for n in program.conv_nodes}
    assert outs[0] == (32, 1) and outs[1] == (64, 2) and outs[43] == (1024, 32) and outs[74] == (255, 8)


def _write(tmp, name, layers):
    p = tmp / name
    p.write_text(yaml.safe_dump({"layers_config": layers}))
    return str(p)


def test_error_behaviour_matches_reference(tmp_path):
    conv = {"type": "convolutional", "batch_normalize": 1, "filters": 32, "size": 3, "stride": 1, "pad": 1,
            "activation": "leaky"}
    # unknown layer type -> ValueError (reference: core/parse_model.py:277)
    f = _write(tmp_path, "a.yaml", [{"type": "route", "source": {"inputs": [0]}}, {"type": "dropout"}])
    with pytest.raises(ValueError, match="not recognized"):
        build_program([{"name": "head0", "layers_config_file": f, "outputs_layers": [-1]}], "head", 80)
    # route with three sources -> ValueError (core/parse_model.py:140)
    f = _write(tmp_path, "b.yaml", [{"type": "route", "source": {"inputs": [0]}}, dict(conv), dict(conv), dict(conv),
                                    {"type": "route", "source": {"layers": [-1, -2, -3]}}])
    with pytest.raises(ValueError, match="Invalid number of layers"):
        build_program([{"name": "head0", "layers_config_file": f, "outputs_layers": [-1]}], "head", 80)
    # bad activation -> AssertionError (core/parse_model.py:48)
    bad = dict(conv, activation="relu")
    f = _write(tmp_path, "c.yaml", [{"type": "route", "source": {"inputs": [0]}}, bad])
    with pytest.raises(AssertionError, match="Invalid activation"):
        build_program([{"name": "head0", "layers_config_file": f, "outputs_layers": [-1]}], "head", 80)
    # missing source sub-model -> Exception (core/parse_model.py:230)
    f = _write(tmp_path, "d.yaml", [{"type": "route", "source": {"inputs": [0]}}, dict(conv)])
    with pytest.raises(Exception):
        build_program([{"name": "head0", "inputs": {"source": [{"name": "nope"}]}, "layers_config_file": f,
                        "outputs_layers": [-1]}], "head", 80)


def test_filters_expression():
    assert eval_int_expr("3*(2+2+1+nclasses)", {"nclasses": 80}) == 255
    assert eval_int_expr(64, {}) == 64
    with pytest.raises(ValueError):
        eval_int_expr("__import__('os').system('true')", {})


def test_grid_sizes_follow_image_size(program):
    # F5: the reference hard-codes 13/26/52; here grids derive from the tensor divisors
    assert [608 // program.tensors[o].div for o in program.outputs] == [19, 38, 76]


def test_reference_style_model_yaml_loads():
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), nclasses=3)
    assert p.tensors[p.outputs[0]].channels == 24
