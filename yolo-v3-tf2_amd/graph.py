"""Model description -> flat conv program.

Reads the model YAML schema of the reference's graph builder
(reference: core/parse_model.py:216-314 `create_sub_model_inputs`,
`create_sub_model_layers`, `build_model`) and produces

  * a node list with the reference's layer semantics (conv / shortcut / route /
    upsample / yolo, reference: core/parse_model.py:13-56,59-75,102-160,209-213);
  * a *lowered* program for the HIP runtime in which BatchNorm, LeakyReLU, the
    shortcut add and the nearest-upsample + channel-concat feeding a 1x1 conv
    are folded into the conv launches.

No arithmetic happens here; this is host logic only and runs without a GPU.
"""
from __future__ import annotations

import ast
import operator
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import yaml

# ----------------------------------------------------------------------------
# tiny arithmetic evaluator (the reference calls eval() on `filters` strings,
# core/parse_model.py:258-259; we accept the same expressions without eval)
# ----------------------------------------------------------------------------
_BINOPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul,
           ast.FloorDiv: operator.floordiv}


def eval_int_expr(expr, names: Dict[str, int]) -> int:
    if isinstance(expr, int):
        return expr

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, int):
            return n.value
        if isinstance(n, ast.Name) and n.id in names:
            return names[n.id]
        if isinstance(n, ast.BinOp) and type(n.op) in _BINOPS:
            return _BINOPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.USub):
            return -ev(n.operand)
        raise ValueError(f"unsupported expression in model config: {expr!r}")

    return int(ev(ast.parse(str(expr), mode="eval")))


# ----------------------------------------------------------------------------
# node-level graph (reference semantics, nothing fused)
# ----------------------------------------------------------------------------
@dataclass
class Tensor:
    id: int
    channels: int
    div: int            # spatial size = image_size // div
    producer: str = ""  # debugging aid
    yolo: bool = False  # viewed as [B,g,g,3,5+nc]


@dataclass
class Node:
    kind: str                       # conv | add | upsample | concat | yolo
    inputs: List[int]
    output: int
    # conv attributes
    size: int = 0
    stride: int = 1
    filters: int = 0
    bn: bool = False
    leaky: bool = False
    conv_index: int = -1            # creation order == Darknet weight order
    sub_model: str = ""


@dataclass
class ConvOp:
    """One launch of the fused conv kernel."""
    conv_index: int
    size: int
    stride: int
    cin: int
    cout: int
    bn: bool
    leaky: bool
    src0: int                       # tensor id holding input channels [0, c0)
    src0_upsample: bool             # src0 is read through a nearest x2 up-sampling
    c0: int
    src1: int                       # -1 or tensor id holding channels [c0, cin)
    residual: int                   # -1 or tensor id added after the activation
    dst: int
    in_div: int                     # input spatial divisor (after up-sampling)
    out_div: int


@dataclass
class AuxOp:
    kind: str                       # add | upsample | concat
    inputs: List[int]
    dst: int


@dataclass
class Program:
    tensors: List[Tensor]
    nodes: List[Node]
    ops: list                       # ConvOp | AuxOp in execution order
    input_tensor: int
    outputs: List[int]              # head outputs in model order (13, 26, 52 grid)
    nclasses: int
    conv_nodes: List[Node] = field(default_factory=list)
    model_config_file: Optional[str] = None   # the model.yaml this program was read from (load_program)

    def conv_ops(self) -> List[ConvOp]:
        return [o for o in self.ops if isinstance(o, ConvOp)]

    def flops_per_image(self, image_size: int, backbone_only: bool = False) -> float:
        """2*MAC over the convolutions (SURVEY.md 8d algorithmic work)."""
        tot = 0.0
        for n in self.conv_nodes:
            if backbone_only and n.sub_model != "backbone":
                continue
            t_in = self.tensors[n.inputs[0]]
            t_out = self.tensors[n.output]
            ho = image_size // t_out.div
            tot += 2.0 * n.size * n.size * t_in.channels * n.filters * ho * ho
        return tot

    def n_params(self) -> int:
        tot = 0
        for n in self.conv_nodes:
            cin = self.tensors[n.inputs[0]].channels
            tot += n.size * n.size * cin * n.filters + (4 * n.filters if n.bn else n.filters)
        return tot


class _Builder:
    def __init__(self, nclasses: int, config_root: Optional[str]):
        self.tensors: List[Tensor] = []
        self.nodes: List[Node] = []
        self.nclasses = nclasses
        self.nconv = 0
        self.root = config_root

    def new_tensor(self, channels, div, producer="", yolo=False) -> int:
        t = Tensor(len(self.tensors), channels, div, producer, yolo)
        self.tensors.append(t)
        return t.id

    # reference: core/parse_model.py:13-56
    def conv(self, x, conf, sub):
        stride, filters, size = int(conf["stride"]), int(conf["filters"]), int(conf["size"])
        act = conf["activation"]
        assert act in ("linear", "leaky"), "Invalid activation: {}".format(act)
        tin = self.tensors[x]
        if stride > 1 and (stride != 2 or size != 3):
            raise ValueError("only the 3x3/2 down-sampling conv of the reference is supported")
        out = self.new_tensor(filters, tin.div * stride, f"{sub}.conv{self.nconv}")
        self.nodes.append(Node("conv", [x], out, size=size, stride=stride, filters=filters,
                               bn="batch_normalize" in conf, leaky=(act == "leaky"),
                               conv_index=self.nconv, sub_model=sub))
        self.nconv += 1
        return out

    # reference: core/parse_model.py:143-160
    def shortcut(self, x, conf, layers, sub):
        assert conf["activation"] == "linear", "Invalid activation: {}".format(conf["activation"])
        frm = layers[int(conf["from"])]
        a, b = self.tensors[frm], self.tensors[x]
        if (a.channels, a.div) != (b.channels, b.div):
            raise ValueError("shortcut operands differ in shape")
        out = self.new_tensor(a.channels, a.div, f"{sub}.add")
        self.nodes.append(Node("add", [frm, x], out, sub_model=sub))
        return out

    # reference: core/parse_model.py:102-140
    def route(self, conf, inputs_entry, layers, sub):
        sel = []
        if "layers" in conf["source"]:
            sel = [layers[int(i)] for i in conf["source"]["layers"]]
        if "inputs" in conf["source"]:
            if isinstance(inputs_entry, list):
                sel += [inputs_entry[i] for i in conf["source"]["inputs"]]
            else:
                sel += [inputs_entry]
        if len(sel) == 1:
            return sel[0]
        if len(sel) == 2:
            a, b = self.tensors[sel[0]], self.tensors[sel[1]]
            if a.div != b.div:
                raise ValueError("route operands differ in spatial size")
            out = self.new_tensor(a.channels + b.channels, a.div, f"{sub}.concat")
            self.nodes.append(Node("concat", list(sel), out, sub_model=sub))
            return out
        raise ValueError("Invalid number of layers: {}".format(len(sel)))

    # reference: core/parse_model.py:59-75
    def upsample(self, x, conf, sub):
        s = int(conf["stride"])
        if s != 2:
            raise ValueError("only x2 up-sampling is supported")
        t = self.tensors[x]
        if t.div % 2:
            raise ValueError("cannot up-sample past the input resolution")
        out = self.new_tensor(t.channels, t.div // 2, f"{sub}.up")
        self.nodes.append(Node("upsample", [x], out, sub_model=sub))
        return out

    # reference: core/parse_model.py:209-213 (Reshape to [g,g,3,5+nc]; a view)
    def yolo(self, x, sub):
        t = self.tensors[x]
        if t.channels != 3 * (5 + self.nclasses):
            raise ValueError("yolo layer input must have 3*(5+nclasses) channels")
        out = self.new_tensor(t.channels, t.div, f"{sub}.yolo", yolo=True)
        self.nodes.append(Node("yolo", [x], out, sub_model=sub))
        return out

    # reference: core/parse_model.py:248-278
    def sub_model_layers(self, layers_config, inputs, sub):
        x = inputs if not isinstance(inputs, list) else None
        layers: List[int] = []
        for conf in layers_config:
            conf = dict(conf)
            kind = conf["type"]
            if kind == "convolutional":
                conf["filters"] = eval_int_expr(conf["filters"], {"nclasses": self.nclasses})
                x = self.conv(x, conf, sub)
            elif kind == "shortcut":
                x = self.shortcut(x, conf, layers, sub)
            elif kind == "yolo":
                x = self.yolo(x, sub)
            elif kind == "route":
                x = self.route(conf, inputs, layers, sub)
            elif kind == "upsample":
                x = self.upsample(x, conf, sub)
            else:
                # 'maxpool' (tiny model only) is outside the hot path: SURVEY.md section 2
                raise ValueError("{} not recognized as layer_conf type".format(kind))
            layers.append(x)
        return layers


def _resolve(path, root):
    if os.path.isabs(path) or os.path.exists(path) or root is None:
        return path
    return os.path.join(root, path)


def build_program(sub_models_configs: Sequence[dict], output_stage: str = "head", nclasses: int = 80,
                  config_root: Optional[str] = None, in_channels: int = 3) -> Program:
    """reference: core/parse_model.py:279-314 (build_model)."""
    b = _Builder(nclasses, config_root)
    model_input = b.new_tensor(in_channels, 1, "input")
    subs = []  # [{'name':..., 'outputs': [tensor ids]}]
    for cfg in sub_models_configs:
        name = cfg["name"]
        inputs_config = cfg.get("inputs")
        if inputs_config:
            if "shape" in inputs_config:
                raise ValueError("explicit input shapes are not supported; the model input is [B,S,S,3]")
            data_inputs = []
            for src in inputs_config["source"]:
                found = [s for s in subs if s["name"] == src["name"]]
                if not found:
                    raise Exception(f'Error: sub-model {src["name"]} not found')
                outs = found[0]["outputs"]
                # Keras hands back a bare tensor for single-output sub-models, so the reference
                # honours entry_index only for multi-output producers (core/parse_model.py:233-241)
                idx = src.get("entry_index", 0) if len(outs) > 1 else 0
                data_inputs.append(outs[idx])
            sub_inputs = data_inputs[0] if len(data_inputs) == 1 else data_inputs
        else:
            sub_inputs = model_input
        with open(_resolve(cfg["layers_config_file"], config_root), "r") as f:
            layers_config = yaml.safe_load(f)["layers_config"]
        layers = b.sub_model_layers(layers_config, sub_inputs, name)
        subs.append({"name": name, "outputs": [layers[int(i)] for i in cfg["outputs_layers"]]})
    outputs = []
    for s in subs:
        if output_stage in s["name"]:
            outputs += s["outputs"]
    prog = Program(b.tensors, b.nodes, [], model_input, outputs, nclasses)
    prog.conv_nodes = [n for n in b.nodes if n.kind == "conv"]
    _lower(prog)
    return prog


def find_config_root(model_config_file: str, sub_models_configs) -> Optional[str]:
    """Paths inside model.yaml are relative to the directory the reference is run from; when the process runs
    elsewhere, walk up from the model file until they resolve."""
    probe = sub_models_configs[0]["layers_config_file"]
    if os.path.isabs(probe) or os.path.exists(probe):
        return None
    d = os.path.dirname(os.path.abspath(model_config_file))
    while d != os.path.dirname(d):
        if os.path.exists(os.path.join(d, probe)):
            return d
        d = os.path.dirname(d)
    return None


def load_program(model_config_file: str, nclasses: int = 80) -> Program:
    with open(model_config_file, "r") as f:
        cfg = yaml.safe_load(f)
    root = find_config_root(model_config_file, cfg["sub_models_configs"])
    prog = build_program(cfg["sub_models_configs"], cfg.get("output_stage", "head"), nclasses, root)
    prog.model_config_file = os.path.abspath(model_config_file)
    return prog


# ----------------------------------------------------------------------------
# lowering: fold add / upsample / concat into conv launches where legal
# ----------------------------------------------------------------------------
def _lower(p: Program):
    consumers: Dict[int, List[Node]] = {}
    for n in p.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    producer = {n.output: n for n in p.nodes}
    external = set(p.outputs)
    folded = set()          # node outputs that never materialise
    alias: Dict[int, int] = {}

    def real(t):
        while t in alias:
            t = alias[t]
        return t

    ops = []
    for n in p.nodes:
        if n.kind == "yolo":
            alias[n.output] = n.inputs[0]       # pure view
            continue
        if n.kind == "conv":
            src = n.inputs[0]
            tin = p.tensors[src]
            op = ConvOp(n.conv_index, n.size, n.stride, tin.channels, n.filters, n.bn, n.leaky,
                        src0=real(src), src0_upsample=False, c0=tin.channels, src1=-1, residual=-1,
                        dst=n.output, in_div=tin.div, out_div=p.tensors[n.output].div)
            pn = producer.get(src)
            # K6: [upsample(a), b] concat -> 1x1 conv; read both sources in the conv's A-gather
            if (pn is not None and pn.kind == "concat" and n.size == 1 and src not in external
                    and len(consumers.get(src, [])) == 1):
                a, bsrc = pn.inputs
                pa = producer.get(a)
                ca = p.tensors[a].channels
                if ca % 32 == 0:
                    op.c0, op.src1 = ca, real(bsrc)
                    if (pa is not None and pa.kind == "upsample" and a not in external
                            and len(consumers.get(a, [])) == 1):
                        op.src0, op.src0_upsample = real(pa.inputs[0]), True
                        folded.add(a)
                    else:
                        op.src0 = real(a)
                    folded.add(src)
            # K5: conv -> add(from, conv) with the conv feeding nothing else
            cons = consumers.get(n.output, [])
            if (len(cons) == 1 and cons[0].kind == "add" and n.output not in external
                    and cons[0].inputs[1] == n.output and cons[0].inputs[0] != n.output):
                op.residual = real(cons[0].inputs[0])
                op.dst = cons[0].output
                folded.add(cons[0].output)
                alias[n.output] = cons[0].output
            ops.append(op)
        elif n.kind in ("add", "upsample", "concat"):
            if n.output in folded:
                continue
            ops.append(AuxOp(n.kind, [real(i) for i in n.inputs], n.output))
    # drop aux producers that were folded away before being visited (upsample/concat precede the conv)
    ops = [o for o in ops if not (isinstance(o, AuxOp) and o.dst in folded)]
    p.ops = ops
    p.outputs = [real(o) for o in p.outputs]
