import numpy as np
import pytest

from yolo_v3_tf2_amd import weights as W


def test_synthetic_weights_are_seeded_and_complete(program):
    a, b = W.synthetic_weights(program, 4321), W.synthetic_weights(program, 4321)
    assert sorted(a) == sorted(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert sum(v.size for v in a.values()) == program.n_params()
    assert a["conv0.w"].shape == (3, 3, 3, 32) and a["conv58.bias"].shape == (255,) and "conv58.gamma" not in a
    obj = a["conv58.bias"].reshape(3, 85)[:, 4]
    assert (obj == -4.0).all()


def test_safetensors_round_trip(tmp_path, program, weights):
    p = str(tmp_path / "w.safetensors")
    W.save_weights(p, weights)
    back = W.load_weights(p)
    assert sorted(back) == sorted(weights) and all(np.array_equal(back[k], weights[k]) for k in weights)


def test_darknet_layout_round_trip(tmp_path, program, weights):
    """Layout restated from reference convert.py:36-74,93-95: header 5 x int32; per conv [beta,gamma,mean,var] or bias,
    then (Cout,Cin,kh,kw) weights."""
    p = str(tmp_path / "yolov3.weights")
    W.write_darknet_weights(p, program, weights)
    import os
    assert os.path.getsize(p) == 20 + 4 * program.n_params()
    back = W.read_darknet_weights(p, program)
    assert all(np.array_equal(back[k], weights[k]) for k in weights)
    raw = np.fromfile(p, dtype="<f4", offset=20)
    # first conv: 32 x [beta, gamma, mean, var] then weights in (Cout,Cin,kh,kw) order
    assert np.array_equal(raw[:32], weights["conv0.beta"]) and np.array_equal(raw[32:64], weights["conv0.gamma"])
    assert raw[128] == weights["conv0.w"][0, 0, 0, 0] and raw[129] == weights["conv0.w"][0, 1, 0, 0]
    with open(p, "ab") as f:
        f.write(b"\0\0\0\0")
    with pytest.raises(ValueError, match="trailing"):
        W.read_darknet_weights(p, program)


def test_darknet_reader_against_hand_built_file(tmp_path):
    """The reader against a file laid out BY HAND from the reference's description of the format (convert.py:93-95 header
    of five int32; :50-55 a batch-normalised conv stores 4 x filters floats in Darknet order beta, gamma, mean, var;
    :58 a bias conv stores its bias; :61-68 then filters x in_dim x size x size weights, which Keras wants transposed
    [2,3,1,0] to (kh, kw, Cin, Cout)) -- independent of this repository's writer."""
    import struct
    from tests.helpers import mini_program
    p = mini_program(2, [dict(filters=3, size=3)], [dict(filters=4, size=1, bn=False, act="linear"),
                                                      dict(filters=4, size=1, bn=False, act="linear"),
                                                      dict(filters=4, size=1, bn=False, act="linear")])
    val = lambda layer, n, c, u, v: 1000.0 * layer + 100.0 * n + 10.0 * c + 3.0 * u + v     # distinct per element
    blob = struct.pack("<5i", 0, 2, 0, 32013312, 0)
    # conv 0: 3 filters, 2 input channels, 3x3, batch-normalised
    blob += struct.pack("<3f", 0.1, 0.2, 0.3)          # beta
    blob += struct.pack("<3f", 1.1, 1.2, 1.3)          # gamma
    blob += struct.pack("<3f", -0.1, -0.2, -0.3)       # mean
    blob += struct.pack("<3f", 0.9, 0.8, 0.7)          # var
    for n in range(3):
        for c in range(2):
            for u in range(3):
                for v in range(3):
                    blob += struct.pack("<f", val(0, n, c, u, v))
    # convs 1..3: 4 filters, 3 input channels, 1x1, bias
    for layer in (1, 2, 3):
        blob += struct.pack("<4f", *[layer + 0.25 * n for n in range(4)])
        for n in range(4):
            for c in range(3):
                blob += struct.pack("<f", val(layer, n, c, 0, 0))
    path = tmp_path / "hand.weights"
    path.write_bytes(blob)
    got = W.read_darknet_weights(str(path), p)
    assert np.allclose(got["conv0.beta"], [0.1, 0.2, 0.3]) and np.allclose(got["conv0.gamma"], [1.1, 1.2, 1.3])
    assert np.allclose(got["conv0.mean"], [-0.1, -0.2, -0.3]) and np.allclose(got["conv0.var"], [0.9, 0.8, 0.7])
    w0 = got["conv0.w"]
    assert w0.shape == (3, 3, 2, 3)
    for u in range(3):
        for v in range(3):
            for c in range(2):
                for n in range(3):
                    assert w0[u, v, c, n] == np.float32(val(0, n, c, u, v))
    for layer in (1, 2, 3):
        assert np.allclose(got[f"conv{layer}.bias"], [layer + 0.25 * n for n in range(4)])
        w = got[f"conv{layer}.w"]
        assert w.shape == (1, 1, 3, 4) and "conv%d.gamma" % layer not in got
        assert all(w[0, 0, c, n] == np.float32(val(layer, n, c, 0, 0)) for c in range(3) for n in range(4))
    path.write_bytes(blob[:-4])
    with pytest.raises(ValueError, match="truncated"):
        W.read_darknet_weights(str(path), p)
