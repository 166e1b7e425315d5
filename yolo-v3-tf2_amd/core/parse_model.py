"""Host mirror of the reference's graph builder (reference: core/parse_model.py).

`ParseModel().build_model(model_inputs, sub_models_configs, output_stage, decay_factor, nclasses)`
keeps the reference's name, argument meaning and error behaviour (ValueError for an unknown layer
type or a route with more than two sources, AssertionError for a bad activation, Exception for a
missing source sub-model) but returns a `YoloModel` whose forward pass is the fused HIP conv program
instead of a Keras functional graph.  There is no CPU fallback: calling the model without a GPU or
without liby3hip.so raises.
"""
from __future__ import annotations

import os

import numpy as np
import yaml

from ..graph import build_program
from .. import weights as _weights


class Input:
    """Placeholder standing in for tensorflow.keras.Input (reference: inference.py:87)."""

    def __init__(self, shape=(None, None, 3), name="input", **_):
        self.shape = tuple(shape)
        self.name = name


class _LoadStatus:
    def expect_partial(self):  # reference: inference.py:102
        return self


class YoloModel:
    """Callable network: image batch [B,S,S,3] (fp32, NHWC, values in [0,1]) -> [grid13, grid26, grid52],
    each [B,g,g,3,5+nclasses] (reference: the Keras Model returned by build_model, core/parse_model.py:313)."""

    def __init__(self, program, name="yolo"):
        self.program = program
        self.name = name
        self._net = None
        self._weights = None

    # lazily create the device object so that building/inspecting a model works without a GPU
    def _device_net(self):
        if self._net is None:
            from ..runtime import Net
            self._net = Net(self.program)
            if self._weights is not None:
                self._net.load_weights(self._weights)
        return self._net

    def set_weights_dict(self, weights):
        self._weights = weights
        if self._net is not None:
            self._net.load_weights(weights)

    def load_weights(self, path):
        """`.safetensors` (this package's container) or Darknet `.weights` (reference: convert.py:93-137).
        TensorFlow checkpoints cannot be read without TensorFlow."""
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        if path.endswith(".weights"):
            w = _weights.read_darknet_weights(path, self.program)
        elif path.endswith(".safetensors"):
            w = _weights.load_weights(path)
        else:
            raise ValueError(f"unsupported weight file {path!r}: use .safetensors or Darknet .weights")
        self.set_weights_dict(w)
        return _LoadStatus()

    def __call__(self, images, training=False):
        import torch
        if training:
            raise NotImplementedError("the HIP path is inference only")
        if self._weights is None:
            raise RuntimeError("model has no weights: call load_weights() or set_weights_dict() first")
        net = self._device_net()
        if isinstance(images, np.ndarray):
            images = torch.from_numpy(np.ascontiguousarray(images, np.float32)).cuda()
        return net.forward(images.contiguous())

    def predict(self, images, batch_size=None, **_):
        return [g.cpu().numpy() for g in self(images)]

    def summary(self, print_fn=print):
        p = self.program
        print_fn(f'Model: "{self.name}"')
        for n in p.conv_nodes:
            cin = p.tensors[n.inputs[0]].channels
            print_fn(f"conv{n.conv_index:<3d} {n.sub_model:<9s} {n.size}x{n.size}/{n.stride} {cin:>4d} -> {n.filters:<4d}"
                     f" {'bn' if n.bn else 'bias':<4s} {'leaky' if n.leaky else 'linear'}")
        print_fn(f"Total params: {p.n_params():,}")
        print_fn(f"GFLOP / image @416: {p.flops_per_image(416) / 1e9:.3f}")


class ParseModel:
    def build_model(self, model_inputs, sub_models_configs, output_stage="head", decay_factor=0, nclasses=0, **kwargs):
        """reference: core/parse_model.py:279-314.  `model_inputs` is accepted for signature compatibility
        (the input is always [B,S,S,3]); `decay_factor` only affects training in the reference."""
        in_ch = 3
        shape = getattr(model_inputs, "shape", None)
        if shape is not None and len(shape) >= 1 and shape[-1] not in (None, 3):
            raise ValueError("the HIP path expects 3-channel images")
        program = build_program(sub_models_configs, output_stage, nclasses, config_root=kwargs.get("config_root"),
                                in_channels=in_ch)
        return YoloModel(program)

    def create_model(self, nclasses, model_config_file):
        """reference: core/parse_model.py:316-322"""
        with open(model_config_file, "r") as f:
            cfg = yaml.safe_load(f)
        return self.build_model(Input(shape=(None, None, 3)), cfg["sub_models_configs"],
                                cfg.get("output_stage", "head"), nclasses=nclasses)
