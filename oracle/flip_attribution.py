"""Attribution of end-to-end NMS selection differences ("an explicit near-tie detector that
reports, not hides, any flip").

TEST INFRASTRUCTURE ONLY (imported by tests/ and by bench.py's parity gate).

Two runs of the detect path (reference side r, device side g) feed *different* fp32 boxes / scores (summation order)
into the *same* NMS (reference core/yolo_nms.py:26-33; the NMS itself is bit-exact on identical inputs, checked
separately).  When their selections differ, every difference must trace back to ONE decision of the algorithm whose
inputs sit on opposite sides of a threshold in the two runs:

  score_threshold : a box is a candidate in one run only          margin = max |score - S| over the two runs
  sort_order      : two kept boxes swap places                    margin = |score_a - score_b| in either run
  iou_threshold   : a box is suppressed by an earlier kept box    margin = max |IoU(k, x) - T| over the two runs
                    in one run only

`margin` is how far the deciding quantity is from its threshold; it is bounded by the deviation between the runs, so a
flip with a margin far above max|dscore| / the IoU change caused by max|dbox| is NOT explained and is returned with
cause "unexplained".
"""
from __future__ import annotations

import numpy as np

EPS_UNION = np.float32(1e-8)


def _iou(a, b):
    """IoU of two boxes in TF's association order, fp32."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    iw = max(np.float32(min(a[2], b[2]) - max(a[0], b[0])), np.float32(0))
    ih = max(np.float32(min(a[3], b[3]) - max(a[1], b[1])), np.float32(0))
    inter = np.float32(iw * ih)
    area_a = np.float32((a[2] - a[0]) * (a[3] - a[1]))
    area_b = np.float32((b[2] - b[0]) * (b[3] - b[1]))
    return float(inter / np.float32(area_a + area_b - inter + EPS_UNION))


def _first_difference(sel_r, n_r, sel_g, n_g):
    n = min(n_r, n_g)
    for p in range(n):
        if sel_r[p] != sel_g[p]:
            return p
    return n if n_r != n_g else -1


def _explain_missing(x, p, kept_other, boxes_own, boxes_other, scores_own, scores_other, iou_thr, score_thr):
    """Box x is kept at position p in its OWN run and is not at position p in the OTHER run.  Why?"""
    s_own, s_other = float(scores_own[x]), float(scores_other[x])
    if not (s_other > score_thr):
        return {"cause": "score_threshold", "margin": max(abs(s_own - score_thr), abs(s_other - score_thr)),
                "box": int(x)}
    # suppressed in the other run by one of the boxes it kept ahead of x?
    for k in kept_other:
        if k == x:
            break
        i_other = _iou(boxes_other[k], boxes_other[x])
        if i_other >= iou_thr:
            i_own = _iou(boxes_own[k], boxes_own[x])
            side = min(float(min(bx[2] - bx[0], bx[3] - bx[1])) for bx in (boxes_own[k], boxes_own[x]))
            return {"cause": "iou_threshold", "margin": max(abs(i_own - iou_thr), abs(i_other - iou_thr)),
                    "box": int(x), "suppressor": int(k), "iou_own": i_own, "iou_other": i_other,
                    "min_box_side": side}
    # kept by both, at different positions: an order swap with whoever sits at p in the other run
    if x in kept_other and p < len(kept_other):
        y = kept_other[p]
        return {"cause": "sort_order", "margin": max(abs(float(scores_own[x]) - float(scores_own[y])),
                                                       abs(float(scores_other[x]) - float(scores_other[y]))),
                "box": int(x), "swapped_with": int(y)}
    return None


def attribute(boxes_r, scores_r, sel_r, nv_r, boxes_g, scores_g, sel_g, nv_g, iou_thr=0.5, score_thr=0.1):
    """Per image with a differing selection: the FIRST differing position and the decision that explains it (later
    differences are consequences of the first).  Returns a list of dicts
    {image, position, ref_index, dev_index, cause, margin, ...}; empty when the selections are equal."""
    out = []
    for b in range(len(nv_r)):
        n_r, n_g = int(nv_r[b]), int(nv_g[b])
        p = _first_difference(sel_r[b], n_r, sel_g[b], n_g)
        if p < 0:
            continue
        kr, kg = [int(v) for v in sel_r[b][:n_r]], [int(v) for v in sel_g[b][:n_g]]
        rec = {"image": b, "position": p, "ref_index": kr[p] if p < n_r else None, "dev_index": kg[p] if p < n_g else None}
        why = None
        if p < n_r:       # the reference keeps kr[p] here, the device does not
            why = _explain_missing(kr[p], p, kg, boxes_r[b], boxes_g[b], scores_r[b], scores_g[b], iou_thr, score_thr)
            if why:
                why["kept_by"] = "reference"
        if why is None and p < n_g:
            why = _explain_missing(kg[p], p, kr, boxes_g[b], boxes_r[b], scores_g[b], scores_r[b], iou_thr, score_thr)
            if why:
                why["kept_by"] = "device"
        rec.update(why or {"cause": "unexplained", "margin": float("inf")})
        out.append(rec)
    return out


def explained(flips, dscore_max, dbox_max):
    """A flip is explained when its margin is within what the measured deviation between the runs can move the
    deciding quantity: 2 * max|dscore| for the two score decisions; for the IoU decision, moving every coordinate of
    two boxes by at most d changes intersection and union by <= ~4 d * side each, i.e. the IoU by <= 8 d / min(side)
    (first order) -- the bound used here, with the smaller side of the two boxes involved."""
    for f in flips:
        if f["cause"] == "unexplained":
            return False
        if f["cause"] in ("score_threshold", "sort_order") and f["margin"] > 2.0 * max(dscore_max, 1e-7):
            return False
        if f["cause"] == "iou_threshold" and f["margin"] > 8.0 * dbox_max / max(f.get("min_box_side", 0.0), 1e-6):
            return False
    return True
