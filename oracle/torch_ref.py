"""Second, independent CPU implementation of the network: PyTorch-CPU operators (oneDNN convolutions) driven by the
oracle's own model.yaml interpreter.

TEST INFRASTRUCTURE ONLY (see oracle/y3_oracle.c).  Two uses:
  * tests/test_oracle.py cross-checks the C restatement against it (different code path, same maths);
  * bench.py's `cpu_baseline` leg times it: the reference's CPU path is TensorFlow 2.8 (Eigen/oneDNN convolutions),
    which is not installed; PyTorch-CPU runs the same layer sequence on the same oneDNN family of kernels, so it is the
    closest stand-in for "the reference on the host cores" that can run here (case 2).

Layer semantics follow reference core/parse_model.py:13-56 (ZeroPadding2D(((1,0),(1,0))) + 'valid' for stride 2, 'same'
for stride 1, BatchNormalization eps 1e-3 on moving statistics, LeakyReLU(0.1)), :59-75, :102-160, :209-213.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import model_reader as R

BN_EPS = 1e-3


class _TorchBackend:
    def __init__(self, weights):
        self.w = weights
        self.nconv = 0
        self._cache = {}

    def _t(self, key):
        t = self._cache.get(key)
        if t is None:
            a = self.w[key]
            t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
            if key.endswith(".w"):
                t = t.permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW
            self._cache[key] = t
        return t

    def conv(self, x, filters, size, stride, bn, leaky, pad, sub):
        i = self.nconv
        self.nconv += 1
        w = self._t(f"conv{i}.w")
        if stride == 2:
            y = F.conv2d(F.pad(x, (1, 0, 1, 0)), w, stride=2)   # ZeroPadding2D(((1,0),(1,0))) + 'valid'
        else:
            y = F.conv2d(x, w, padding=size // 2)               # 'same'
        if bn:
            y = F.batch_norm(y, self._t(f"conv{i}.mean"), self._t(f"conv{i}.var"), self._t(f"conv{i}.gamma"),
                             self._t(f"conv{i}.beta"), False, 0.0, BN_EPS)
        else:
            y = y + self._t(f"conv{i}.bias").view(1, -1, 1, 1)
        return F.leaky_relu(y, 0.1) if leaky else y

    def add(self, a, b):
        return a + b

    def concat(self, a, b):
        return torch.cat([a, b], 1)

    def upsample(self, x, stride):
        return F.interpolate(x, scale_factor=stride, mode="nearest")

    def yolo(self, x, nclasses):
        return x                                                 # reshaped on the way out


class TorchNet:
    """Callable network; weight tensors are converted once (so timing loops measure the operators only)."""

    def __init__(self, model_config_file, weights, nclasses, sub_models=None, output_stage=None):
        self.args = (model_config_file, nclasses)
        self.kw = dict(sub_models=sub_models, output_stage=output_stage)
        self.be = _TorchBackend(weights)
        self.nclasses = nclasses

    def __call__(self, images_nhwc):
        """[B,S,S,3] fp32 NumPy -> list of NHWC NumPy outputs (head grids as [B,g,g,3,5+nc])."""
        self.be.nconv = 0
        x = torch.from_numpy(np.ascontiguousarray(images_nhwc, np.float32)).permute(0, 3, 1, 2)
        with torch.no_grad():
            outs = R.run_model(self.args[0], self.args[1], self.be, x, **self.kw)
        res = []
        for o in outs:
            a = o.permute(0, 2, 3, 1).contiguous().numpy()
            if self.nclasses > 0 and a.shape[-1] == 3 * (5 + self.nclasses):
                a = a.reshape(a.shape[0], a.shape[1], a.shape[2], 3, 5 + self.nclasses)
            res.append(a)
        return res


def forward(model_config_file, weights, images, nclasses, **kw):
    return TorchNet(model_config_file, weights, nclasses, **kw)(images)


def detect(program, weights, images, anchors_table, yolo_max_boxes=100, nms_iou_threshold=0.5,
           nms_score_threshold=0.1):
    """The reference's 5-tuple with the network on PyTorch-CPU operators and the oracle's C decode / score / NMS: the
    second, independent fp32 CPU implementation next to oracle.detect (same inputs, different summation order inside
    the convolutions).  The deviation between the two is the floor any third implementation is measured against."""
    from . import oracle as O
    grids = TorchNet(program.model_config_file, weights, program.nclasses)(images)
    return O.yolo_nms(O.yolo_decode(grids, anchors_table, program.nclasses), yolo_max_boxes, nms_iou_threshold,
                      nms_score_threshold)


def box_deviation(got_boxes, ref_boxes, got_scores, ref_scores):
    """The numbers every parity report quotes, for any pair of runs: raw, inside the unit range, scaled."""
    err = np.abs(got_boxes - ref_boxes)
    unit = np.abs(ref_boxes) <= 1.0
    return {"max_abs_dbox_raw": float(err.max()),
            "max_abs_dbox_coords_within_unit_range": float(err[unit].max()) if unit.any() else 0.0,
            "max_dbox_over_max1_abs_coord": float((err / np.maximum(1.0, np.abs(ref_boxes))).max()),
            "largest_abs_box_coord": float(np.abs(ref_boxes).max()),
            "max_abs_dscore": float(np.abs(got_scores - ref_scores).max())}
