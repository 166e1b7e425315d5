"""The oracle's own model.yaml interpreter (oracle/model_reader.py) against (a) a hand-written wiring table of the
network, (b) the product's graph builder (yolo_v3_tf2_amd/graph.py) -- two independently written readers of the
reference's YAML schema must produce the same node list -- and (c) the node-walking oracle, bit for bit.  No GPU."""
import os

import numpy as np
import pytest

from oracle import model_reader as R
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join(ROOT, "config/models/yolov3/model.yaml")
REF_MODEL = "/root/reference/config/models/yolov3/model.yaml"   # present in the build container only


def test_filters_expression_parser():
    assert R._filters(255, 80) == 255
    assert R._filters("3*(2+2+1+nclasses)", 80) == 255
    assert R._filters(" 3 * (2+2+1+nclasses) ", 3) == 24
    assert R._filters("2*3+4-1", 0) == 9
    with pytest.raises(AssertionError):
        R._filters("3*(2+", 80)
    with pytest.raises(AssertionError):
        R._filters("__import__('os')", 80)


def _node_tuples_from_product(p):
    out = []
    for n in p.nodes:
        attrs = {}
        if n.kind == "conv":
            attrs = dict(size=n.size, stride=n.stride, filters=n.filters, bn=n.bn, leaky=n.leaky,
                         conv_index=n.conv_index, sub_model=n.sub_model, cin=p.tensors[n.inputs[0]].channels)
        out.append((n.kind, tuple(n.inputs), n.output, attrs))
    return out


@pytest.mark.parametrize("which", ["repo", "reference"])
def test_two_readers_agree(which):
    """Same node list (kinds, operand ids IN ORDER, conv attributes, creation order) from both readers."""
    from yolo_v3_tf2_amd.graph import load_program
    path = MODEL if which == "repo" else REF_MODEL
    if not os.path.exists(path):
        pytest.skip("the reference tree is not on this machine")
    tr, outs = R.trace(path, 80)
    p = load_program(path, 80)
    mine = [(k, i, o, {a: v for a, v in at.items() if a != "stride" or k == "conv"}) for k, i, o, at in tr.nodes]
    theirs = _node_tuples_from_product(p)
    assert len(mine) == len(theirs) == 75 + 23 + 2 + 2 + 3
    for a, b in zip(mine, theirs):
        assert a == b, (a, b)
    assert [tr.channels[t] for t in range(len(tr.channels))] == [t.channels for t in p.tensors]
    assert tr.div == [t.div for t in p.tensors]
    # network outputs: the yolo views of the three head convs, 13 / 26 / 52 order
    assert [dict((o, i[0]) for k, i, o, _ in tr.nodes if k == "yolo")[o] for o in outs] == p.outputs


def test_wiring_table():
    """Hand-written from config/models/yolov3/*.yaml (reference: model.yaml:9-77, neck1.yaml:4-65, neck2.yaml:4-65,
    backbone.yaml): conv index -> where its input comes from.  Pins concat operand order = [upsampled, backbone],
    the three backbone taps, the shortcut sources and the head order, independently of either reader's code."""
    tr, outs = R.trace(MODEL, 80)
    conv_out = {at["conv_index"]: o for k, i, o, at in tr.nodes if k == "conv"}
    conv_in = {at["conv_index"]: i[0] for k, i, o, at in tr.nodes if k == "conv"}
    prod = {o: (k, i) for k, i, o, at in tr.nodes}
    # first residual block: conv1 (s2) -> conv2 (1x1) -> conv3 (3x3) -> add(conv1, conv3)
    k, ins = prod[conv_in[4]]
    assert k == "add" and ins == (conv_out[1], conv_out[3])
    # all 23 shortcuts: Add([layers[-3], x]) = (block input, 3x3 conv output)
    adds = [(i, o) for k, i, o, _ in tr.nodes if k == "add"]
    assert len(adds) == 23
    for (a, b), o in adds:
        kb, ib = prod[b]
        assert kb == "conv"
        one_by_one = prod[ib[0]]
        assert one_by_one[0] == "conv" and one_by_one[1] == (a,)
    # backbone taps (outputs_layers -39, -14, -1): inputs of the stride-2 convs 26 and 43, and the last add
    tap0, tap1 = conv_in[26], conv_in[43]
    tap2 = conv_in[52]                               # neck0's first conv reads backbone[2]
    assert prod[tap0][0] == prod[tap1][0] == prod[tap2][0] == "add"
    assert (tr.channels[tap0], tr.div[tap0]) == (256, 8) and (tr.channels[tap1], tr.div[tap1]) == (512, 16)
    assert (tr.channels[tap2], tr.div[tap2]) == (1024, 32)
    # neck0 = convs 52-56, head0 = 57-58 reads conv56
    assert conv_in[57] == conv_out[56] and conv_in[58] == conv_out[57]
    # neck1: conv59 (1x1 256) reads neck0's output; upsample; concat [upsampled, backbone tap1]; conv60 reads it
    assert conv_in[59] == conv_out[56]
    k, ins = prod[conv_in[60]]
    assert k == "concat" and prod[ins[0]] == ("upsample", (conv_out[59],)) and ins[1] == tap1
    assert tr.channels[conv_in[60]] == 768
    # neck2: conv67 reads neck1's output (conv64); concat [upsampled, backbone tap0]
    assert conv_in[65] == conv_out[64] and conv_in[67] == conv_out[64]
    k, ins = prod[conv_in[68]]
    assert k == "concat" and prod[ins[0]] == ("upsample", (conv_out[67],)) and ins[1] == tap0
    assert tr.channels[conv_in[68]] == 384
    assert conv_in[73] == conv_out[72]
    # heads in model order, each the yolo view of a bias/linear 1x1 conv with 255 channels
    assert [prod[o] for o in outs] == [("yolo", (conv_out[58],)), ("yolo", (conv_out[66],)), ("yolo", (conv_out[74],))]
    for k, i, o, at in tr.nodes:
        if k == "conv" and at["conv_index"] in (58, 66, 74):
            assert (at["size"], at["bn"], at["leaky"], at["filters"]) == (1, False, False, 255)
    assert [at["conv_index"] for k, i, o, at in tr.nodes if k == "conv" and at["stride"] == 2] == [1, 4, 9, 26, 43]


def test_lowered_program_wiring_table(program):
    """The same table on the product's FUSED program (what the GPU executes): per conv src0 / src1 / c0 / residual /
    upsample flag / divisors."""
    ops = {o.conv_index: o for o in program.conv_ops()}
    assert len(ops) == 75
    # residual convs: the 3x3 of every block adds the block input, which is what the block's 1x1 conv read
    res = [i for i, o in ops.items() if o.residual >= 0]
    assert res == [3, 6, 8] + list(range(11, 26, 2)) + list(range(28, 43, 2)) + list(range(45, 52, 2))
    for i in res:
        assert ops[i].residual == ops[i - 1].src0 and ops[i].src0 == ops[i - 1].dst and ops[i].size == 3
    tap0, tap1, tap2 = ops[25].dst, ops[42].dst, ops[51].dst
    assert (ops[26].src0, ops[43].src0, ops[52].src0) == (tap0, tap1, tap2)
    o = ops[60]   # neck1 lateral: channels [0,256) = up-sampled conv59, [256,768) = backbone tap1
    assert (o.src0, o.src0_upsample, o.c0, o.src1, o.cin, o.in_div, o.out_div) == (ops[59].dst, True, 256, tap1, 768, 16, 16)
    o = ops[68]   # neck2 lateral: [0,128) = up-sampled conv67, [128,384) = backbone tap0
    assert (o.src0, o.src0_upsample, o.c0, o.src1, o.cin, o.in_div, o.out_div) == (ops[67].dst, True, 128, tap0, 384, 8, 8)
    assert ops[59].src0 == ops[56].dst and ops[67].src0 == ops[64].dst
    assert ops[57].src0 == ops[56].dst and ops[65].src0 == ops[64].dst and ops[73].src0 == ops[72].dst
    for i, o in ops.items():
        if i not in (60, 68):
            assert o.src1 < 0 and not o.src0_upsample and o.c0 == o.cin
        assert o.out_div == o.in_div * o.stride
    assert program.outputs == [ops[58].dst, ops[66].dst, ops[74].dst]


def test_interpreter_equals_node_walk_bitwise(program, weights):
    """oracle.forward re-reads the YAML itself; walking the product's node list must give identical bits (so the two
    readers also agree numerically), in fp32 and in the bf16-emulating mode."""
    x = np.random.default_rng(5).random((1, 64, 64, 3), dtype=np.float32)
    assert program.model_config_file
    for bf16 in (False, True):
        a = O.forward(program, weights, x, bf16=bf16)
        b = O._forward_nodes(program, weights, x, bf16=bf16)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    keep = {program.conv_ops()[3].dst, program.conv_ops()[60].dst}
    _, ka = O.forward(program, weights, x, keep=keep)
    _, kb = O._forward_nodes(program, weights, x, keep=keep)
    assert set(ka) == set(kb) == keep and all(np.array_equal(ka[t], kb[t]) for t in keep)


def test_bf16_storage_rule_matches_fused_program(program):
    """Where a bf16 pipeline rounds, derived from the node graph (model_reader.bf16_stored) == the destinations of the
    fused launches, for the full model and for the backbone alone (whose outputs are residual sums)."""
    tr, outs = R.trace(MODEL, 80)
    stored = R.bf16_stored(tr, outs)
    dsts = {o.dst for o in program.conv_ops()} - set(program.outputs)
    assert stored == dsts
    trb, outsb = R.trace(MODEL, 0, sub_models=["backbone"], output_stage="backbone")
    sb = R.bf16_stored(trb, outsb)
    assert set(outsb) <= sb and len(outsb) == 3          # backbone taps are sums: stored (rounded) even as outputs
