"""The oracle's OWN reader of the reference's model description (model.yaml + one YAML per sub-model).

TEST INFRASTRUCTURE ONLY (see oracle/y3_oracle.c).  Nothing here imports the product package: the
checker must not share the product's YAML -> graph code (yolo_v3_tf2_amd/graph.py), otherwise a wiring
error (route operand order, entry_index, a shortcut's `from`) would be common to the HIP path and to
what it is compared with.

It is an interpreter: it walks the YAML the way the reference's graph builder does and calls a backend
for every layer, holding whatever the backend returns (arrays for `Numeric`, symbolic ids for `Tracer`).

reference: core/parse_model.py
  :13-56   _parse_convolutional   (ZeroPadding top/left for stride 2, Conv2D, BatchNormalization, LeakyReLU)
  :59-75   _parse_upsample        (UpSampling2D(size=stride))
  :102-140 _parse_route           (selected = layers[...] then inputs[...]; one source: identity, two: Concatenate(axis=3))
  :143-160 _parse_shortcut        (Add()([layers[from], x]))
  :163-213 _parse_yolo            (Reshape((g, g, 3, 5 + nclasses)))
  :216-246 create_sub_model_inputs
  :248-278 create_sub_model_layers
  :279-314 build_model
"""
import os

import yaml


def _filters(value, nclasses):
    """`filters` may be an int or an arithmetic string such as '3*(2+2+1+nclasses)' (the reference eval()s it,
    core/parse_model.py:258-259).  Recursive descent over + - * ( ) integers and the name nclasses."""
    if isinstance(value, int):
        return value
    s = str(value).replace(" ", "")
    pos = 0

    def atom():
        nonlocal pos
        assert pos < len(s), value
        if s[pos] == "(":
            pos += 1
            v = expr()
            assert pos < len(s) and s[pos] == ")", value
            pos += 1
            return v
        if s.startswith("nclasses", pos):
            pos += len("nclasses")
            return nclasses
        j = pos
        while j < len(s) and s[j].isdigit():
            j += 1
        assert j > pos, value
        v = int(s[pos:j])
        pos = j
        return v

    def term():
        nonlocal pos
        v = atom()
        while pos < len(s) and s[pos] == "*":
            pos += 1
            v *= atom()
        return v

    def expr():
        nonlocal pos
        v = term()
        while pos < len(s) and s[pos] in "+-":
            op = s[pos]
            pos += 1
            t = term()
            v = v + t if op == "+" else v - t
        return v

    out = expr()
    assert pos == len(s), value
    return out


def _locate(path, model_config_file):
    """layers_config_file paths are relative to the directory the reference is started from."""
    if os.path.isabs(path) or os.path.exists(path):
        return path
    d = os.path.dirname(os.path.abspath(model_config_file))
    while True:
        cand = os.path.join(d, path)
        if os.path.exists(cand):
            return cand
        up = os.path.dirname(d)
        if up == d:
            raise FileNotFoundError(path)
        d = up


def _run_layers(layers_config, inputs, nclasses, be, sub_name):
    """create_sub_model_layers (core/parse_model.py:248-278): x starts as the sub-model's inputs entry."""
    x = inputs
    layers = []
    for conf in layers_config:
        kind = conf["type"]
        if kind == "convolutional":
            stride, size = int(conf["stride"]), int(conf["size"])
            act = conf["activation"]
            assert act in ("linear", "leaky"), "Invalid activation: {}".format(act)
            x = be.conv(x, _filters(conf["filters"], nclasses), size, stride, "batch_normalize" in conf,
                        act == "leaky", int(conf.get("pad", 1)), sub_name)
        elif kind == "shortcut":
            assert conf["activation"] == "linear", "Invalid activation: {}".format(conf["activation"])
            x = be.add(layers[int(conf["from"])], x)                         # Add()([from_layer, x])
        elif kind == "yolo":
            x = be.yolo(x, nclasses)
        elif kind == "route":
            src = conf["source"]
            sel = [layers[int(i)] for i in src["layers"]] if "layers" in src else []
            if "inputs" in src:
                sel += [inputs[i] for i in src["inputs"]] if isinstance(inputs, list) else [inputs]
            if len(sel) == 1:
                x = sel[0]
            elif len(sel) == 2:
                x = be.concat(sel[0], sel[1])                                 # Concatenate(axis=3)(selected_layers)
            else:
                raise ValueError("Invalid number of layers: {}".format(len(sel)))
        elif kind == "upsample":
            x = be.upsample(x, int(conf["stride"]))
        else:
            raise ValueError("{} not recognized as layer_conf type".format(kind))
        layers.append(x)
    return layers


def run_model(model_config_file, nclasses, be, model_input, sub_models=None, output_stage=None):
    """build_model (core/parse_model.py:279-314).  `sub_models`: optional subset of sub-model names to build (in
    file order); `output_stage` defaults to the file's.  Returns the list of outputs of the sub-models whose
    name contains output_stage, flattened in order."""
    with open(model_config_file) as f:
        cfg = yaml.safe_load(f)
    stage = output_stage or cfg.get("output_stage", "head")
    built = []   # (name, output value or list of values)
    for sm in cfg["sub_models_configs"]:
        if sub_models is not None and sm["name"] not in sub_models:
            continue
        ic = sm.get("inputs")
        if ic:
            data = []
            for src in ic["source"]:
                match = [o for n, o in built if n == src["name"]]
                if not match:
                    raise Exception(f'Error: sub-model {src["name"]} not found')
                prod = match[0]
                # a Keras Model with one output returns a bare tensor: entry_index only indexes lists (:233-241)
                data.append(prod[src.get("entry_index", 0)] if isinstance(prod, list) else prod)
            inputs = data[0] if len(data) == 1 else data
        else:
            inputs = model_input
        with open(_locate(sm["layers_config_file"], model_config_file)) as f:
            layers_config = yaml.safe_load(f)["layers_config"]
        layers = _run_layers(layers_config, inputs, nclasses, be, sm["name"])
        outs = [layers[int(i)] for i in sm["outputs_layers"]]
        built.append((sm["name"], outs[0] if len(outs) == 1 else outs))
    result = []
    for name, o in built:
        if stage in name:
            result += o if isinstance(o, list) else [o]
    return result


class Tracer:
    """Symbolic backend: values are integer ids (0 = the model input, then one per created tensor, in creation
    order); records one tuple per node.  Used to compare this reader with the product's graph builder and to
    derive consumer counts."""

    def __init__(self, in_channels=3):
        self.nodes = []          # (kind, inputs tuple, output, attrs dict)
        self.channels = [in_channels]
        self.div = [1]
        self.nconv = 0

    def _new(self, channels, div):
        self.channels.append(channels)
        self.div.append(div)
        return len(self.channels) - 1

    def conv(self, x, filters, size, stride, bn, leaky, pad, sub):
        out = self._new(filters, self.div[x] * stride)
        self.nodes.append(("conv", (x,), out, dict(size=size, stride=stride, filters=filters, bn=bn, leaky=leaky,
                                                    conv_index=self.nconv, sub_model=sub, cin=self.channels[x])))
        self.nconv += 1
        return out

    def add(self, a, b):
        out = self._new(self.channels[a], self.div[a])
        self.nodes.append(("add", (a, b), out, {}))
        return out

    def concat(self, a, b):
        out = self._new(self.channels[a] + self.channels[b], self.div[a])
        self.nodes.append(("concat", (a, b), out, {}))
        return out

    def upsample(self, x, stride):
        out = self._new(self.channels[x], self.div[x] // stride)
        self.nodes.append(("upsample", (x,), out, dict(stride=stride)))
        return out

    def yolo(self, x, nclasses):
        out = self._new(self.channels[x], self.div[x])
        self.nodes.append(("yolo", (x,), out, {}))
        return out


def trace(model_config_file, nclasses, sub_models=None, output_stage=None, in_channels=3):
    t = Tracer(in_channels)
    outs = run_model(model_config_file, nclasses, t, 0, sub_models, output_stage)
    return t, outs


def bf16_stored(tracer, outputs):
    """Which tensors a bf16 pipeline holds in bf16 (= where a value is rounded), stated on the node graph:
    every conv result after BatchNorm/LeakyReLU, except that a conv whose only consumer is the shortcut Add (as
    its second operand) is added first and the SUM is what is stored; a network output that nothing else reads and
    that comes straight from an MFMA conv (not the Cin = 3 first layer, no shortcut) is written in fp32 from the
    accumulators and is not rounded.  Up-sampling / concatenation / the yolo reshape move stored values around."""
    consumers = {}
    for kind, ins, out, _ in tracer.nodes:
        for i in ins:
            consumers.setdefault(i, []).append((kind, ins, out))
    stored = set()
    outs = set(outputs)
    view_of = {out: ins[0] for kind, ins, out, _ in tracer.nodes if kind == "yolo"}
    outs |= {view_of[o] for o in outputs if o in view_of}
    for kind, ins, out, attrs in tracer.nodes:
        cons = [c for c in consumers.get(out, []) if c[0] != "yolo"]
        if kind == "conv":
            fused_add = len(cons) == 1 and cons[0][0] == "add" and cons[0][1][1] == out and cons[0][1][0] != out \
                and out not in outs
            if fused_add:
                continue
            if out in outs and not cons and attrs["cin"] != 3:
                continue
            stored.add(out)
        elif kind == "add":
            stored.add(out)
    return stored
