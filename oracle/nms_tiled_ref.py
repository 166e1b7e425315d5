"""Literal NumPy restatement of TF 2.8's tiled `non_max_suppression_padded` ("v2").

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: restated from the published algorithm in
tensorflow/python/ops/image_ops_impl.py of tensorflow==2.8.1 (requirements.txt:41 of the
reference; TF is not installed and its source is not vendored).
The reference reaches it through core/yolo_nms.py:26-33 with a 3-D `boxes`, which forces
the tiled path.  Structure (tiles of 512, cross-suppression against earlier tiles, then
self-suppression iterated while the IoU mass keeps shrinking, early exit once every image
has `max_output_size` survivors) follows TF op by op so that its quirks are visible:

  * the self-suppression loop stops when `iou_sum - iou_sum_new > iou_threshold` is false,
    i.e. it can stop one round early if a round removes total IoU mass <= threshold
    (only possible when a single pair with IoU == threshold, up to fp32 summation error,
    is removed in the final round).  oracle/y3_oracle.c and the HIP kernel compute the
    converged fixed point (= plain greedy NMS); tests/test_oracle.py (the `tiled` tests) checks that both
    agree on the stress sets and documents this corner as the one known divergence.

Slow (seconds per image at N=10647); used for cross-validation at small/medium sizes.
"""
import numpy as np

F = np.float32


def _bbox_overlap(a, b):
    """a [B,Na,4], b [B,Nb,4] as (y_min,x_min,y_max,x_max) -> iou [B,Na,Nb]"""
    a_y_min, a_x_min, a_y_max, a_x_max = [a[..., i:i + 1] for i in range(4)]
    b_y_min, b_x_min, b_y_max, b_x_max = [b[..., i:i + 1] for i in range(4)]
    T = lambda t: np.transpose(t, (0, 2, 1))
    i_xmin = np.maximum(a_x_min, T(b_x_min))
    i_xmax = np.minimum(a_x_max, T(b_x_max))
    i_ymin = np.maximum(a_y_min, T(b_y_min))
    i_ymax = np.minimum(a_y_max, T(b_y_max))
    i_area = np.maximum(i_xmax - i_xmin, F(0)) * np.maximum(i_ymax - i_ymin, F(0))
    a_area = (a_y_max - a_y_min) * (a_x_max - a_x_min)
    b_area = (b_y_max - b_y_min) * (b_x_max - b_x_min)
    u_area = a_area + T(b_area) - i_area + F(1e-8)
    return (i_area / u_area).astype(F)


def _self_suppression(iou, iou_sum, thr):
    B = iou.shape[0]
    can_suppress_others = (iou.max(axis=1) < thr).astype(F).reshape(B, -1, 1)
    iou_after = ((can_suppress_others * iou).max(axis=1) < thr).astype(F).reshape(B, -1, 1) * iou
    iou_sum_new = iou_after.sum(axis=(1, 2), dtype=F)
    return iou_after, bool(np.any(iou_sum - iou_sum_new > thr)), iou_sum_new


def non_max_suppression_padded(boxes, scores, max_output_size, iou_threshold=0.5, score_threshold=-np.inf,
                               tile_size=512, converge=False):
    """boxes [B,N,4], scores [B,N] -> (idx [B,M] int32, num_valid [B] int32).
    converge=True iterates self-suppression to the true fixed point instead of TF's stop test."""
    boxes = np.array(boxes, dtype=F, copy=True)
    scores = np.array(scores, dtype=F, copy=True)
    B, N = scores.shape
    M = int(max_output_size)
    thr = F(iou_threshold)
    if score_threshold != -np.inf:
        mask = (scores > F(score_threshold)).astype(F)
        scores = scores * mask
        boxes = boxes * mask[..., None]
    # canonicalize_coordinates: decision from box [0,0] only
    y1, x1, y2, x2 = [boxes[..., i:i + 1] for i in range(4)]
    if not (y1[0, 0, 0] <= y2[0, 0, 0]):
        y1, y2 = y2, y1
    if not (x1[0, 0, 0] <= x2[0, 0, 0]):
        x1, x2 = x2, x1
    boxes = np.concatenate([y1, x1, y2, x2], axis=2)
    # argsort DESCENDING == top_k(k=N): stable w.r.t. index
    order = np.argsort(-scores, axis=1, kind="stable").astype(np.int32)
    scores = np.take_along_axis(scores, order, axis=1)
    boxes = np.take_along_axis(boxes, order[..., None], axis=1)
    pad = int(np.ceil(max(N, M) / tile_size)) * tile_size - N
    boxes = np.pad(boxes, [(0, 0), (0, pad), (0, 0)])
    Np = N + pad
    num_iterations = Np // tile_size
    output_size = np.zeros(B, np.int32)
    idx = 0
    tri = (np.arange(tile_size)[None, :] > np.arange(tile_size)[:, None])[None]
    while output_size.min() < M and idx < num_iterations:
        sl = slice(idx * tile_size, (idx + 1) * tile_size)
        box_slice = boxes[:, sl].copy()
        for inner in range(idx):  # _cross_suppression
            new_slice = boxes[:, inner * tile_size:(inner + 1) * tile_size]
            iou = _bbox_overlap(new_slice, box_slice)
            box_slice = (np.all(iou < thr, axis=1)).astype(F)[..., None] * box_slice
        iou = _bbox_overlap(box_slice, box_slice)
        iou = iou * (tri & (iou >= thr)).astype(F)
        iou_sum = iou.sum(axis=(1, 2), dtype=F)
        cond = True
        while cond:
            iou_new, cond, iou_sum_new = _self_suppression(iou, iou_sum, thr)
            if converge:
                cond = bool(np.any(iou_new != iou))
            iou, iou_sum = iou_new, iou_sum_new
        suppressed = iou.sum(axis=1, dtype=F) > 0
        box_slice = box_slice * (F(1) - suppressed.astype(F))[..., None]
        boxes[:, sl] = box_slice
        output_size = output_size + np.any(box_slice > 0, axis=2).sum(axis=1).astype(np.int32)
        idx += 1
    num_valid = np.minimum(output_size, M).astype(np.int32)
    sel_flag = np.any(boxes > 0, axis=2).astype(np.int32) * np.arange(Np, 0, -1, dtype=np.int32)[None]
    # top_k(values, M): largest first, lower index first among equals
    top = -np.sort(-sel_flag, axis=1, kind="stable")[:, :M]
    if top.shape[1] < M:
        top = np.pad(top, [(0, 0), (0, M - top.shape[1])])
    pos = np.minimum(Np - top, N - 1)
    out = np.take_along_axis(order, pos.astype(np.int64), axis=1).astype(np.int32)
    out = np.where(np.arange(M)[None] < num_valid[:, None], out, 0).astype(np.int32)
    return out, num_valid
