"""Weight containers for the conv program.

* `synthetic_weights`  -- seeded random-init weights (no checkpoint ships with the reference:
  SURVEY.md F6); distribution per SURVEY.md 8(d) so activations neither explode nor vanish
  through 75 layers and O(10^2..10^3) boxes per image clear the 0.1 score threshold.
* `save_weights` / `load_weights` -- safetensors container, one entry per conv in creation
  order (the order the reference's converter walks, reference: convert.py:96-137):
      conv{i}.w [k,k,Cin,Cout] (HWIO, the Keras Conv2D kernel layout),
      conv{i}.gamma/.beta/.mean/.var [Cout]   for BatchNormalization convs,
      conv{i}.bias [Cout]                      for the three linear head convs.
* `read_darknet_weights` -- Darknet `.weights` reader (layout restated from
  reference: convert.py:36-74,93-95).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

BN_EPS = 1e-3  # Keras BatchNormalization() default (reference: core/parse_model.py:46)


def synthetic_weights(program, seed: int = 4321) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    nc = program.nclasses
    out: Dict[str, np.ndarray] = {}
    # convs whose output is the residual branch of a shortcut get a small BN gamma, otherwise the
    # 23 shortcut adds double the activation variance 23 times over
    residual_branch = {nd.inputs[1] for nd in program.nodes if nd.kind == "add"}
    for n in program.conv_nodes:
        cin = program.tensors[n.inputs[0]].channels
        k, cout, i = n.size, n.filters, n.conv_index
        std = np.sqrt(2.0 / ((1.0 + 0.1 ** 2) * k * k * cin))
        if not n.bn:
            std *= 0.15     # head logits ~ N(bias, ~1): exp(tw) stays O(1), sigmoids do not saturate
        out[f"conv{i}.w"] = (rng.standard_normal((k, k, cin, cout)) * std).astype(np.float32)
        if n.bn:
            g = 0.25 if n.output in residual_branch else 1.0
            out[f"conv{i}.gamma"] = (g * rng.uniform(0.9, 1.1, cout)).astype(np.float32)
            out[f"conv{i}.beta"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
            out[f"conv{i}.mean"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
            out[f"conv{i}.var"] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
        else:
            bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            if cout == 3 * (5 + nc):
                b = bias.reshape(3, 5 + nc)
                b[:, 4] = -4.0      # objectness
                b[:, 5:] = -1.0     # classes
                bias = b.reshape(-1)
            out[f"conv{i}.bias"] = bias
    return out


def save_weights(path: str, weights: Dict[str, np.ndarray]):
    from safetensors.numpy import save_file
    save_file({k: np.ascontiguousarray(v) for k, v in weights.items()}, path)


def load_weights(path: str) -> Dict[str, np.ndarray]:
    from safetensors.numpy import load_file
    return load_file(path)


def read_darknet_weights(path: str, program) -> Dict[str, np.ndarray]:
    """Darknet binary: 5 x int32 header, then per conv (creation order) float32 LE:
    BN conv: [beta, gamma, mean, var] x Cout, then weights (Cout,Cin,kh,kw);
    bias conv: bias x Cout, then weights.  Weights are transposed to HWIO."""
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        header = np.fromfile(f, dtype="<i4", count=5)
        if header.size != 5:
            raise ValueError("truncated Darknet weights header")
        for n in program.conv_nodes:
            cin = program.tensors[n.inputs[0]].channels
            k, cout, i = n.size, n.filters, n.conv_index
            if n.bn:
                bn = np.fromfile(f, dtype="<f4", count=4 * cout)
                if bn.size != 4 * cout:
                    raise ValueError(f"truncated Darknet weights at conv {i}")
                bn = bn.reshape(4, cout)
                out[f"conv{i}.beta"], out[f"conv{i}.gamma"] = bn[0].copy(), bn[1].copy()
                out[f"conv{i}.mean"], out[f"conv{i}.var"] = bn[2].copy(), bn[3].copy()
            else:
                bias = np.fromfile(f, dtype="<f4", count=cout)
                if bias.size != cout:
                    raise ValueError(f"truncated Darknet weights at conv {i}")
                out[f"conv{i}.bias"] = bias
            w = np.fromfile(f, dtype="<f4", count=cout * cin * k * k)
            if w.size != cout * cin * k * k:
                raise ValueError(f"truncated Darknet weights at conv {i}")
            out[f"conv{i}.w"] = np.ascontiguousarray(w.reshape(cout, cin, k, k).transpose(2, 3, 1, 0))
        if f.read(1):
            raise ValueError("trailing bytes in Darknet weights file (wrong model description?)")
    return out


def write_darknet_weights(path: str, program, weights: Dict[str, np.ndarray]):
    """Inverse of `read_darknet_weights` (used by tests for the round trip)."""
    with open(path, "wb") as f:
        np.array([0, 2, 0, 0, 0], dtype="<i4").tofile(f)
        for n in program.conv_nodes:
            i = n.conv_index
            if n.bn:
                for key in ("beta", "gamma", "mean", "var"):
                    weights[f"conv{i}.{key}"].astype("<f4").tofile(f)
            else:
                weights[f"conv{i}.bias"].astype("<f4").tofile(f)
            np.ascontiguousarray(weights[f"conv{i}.w"].transpose(3, 2, 0, 1)).astype("<f4").tofile(f)
