"""The near-tie detector of the end-to-end selection check (oracle/flip_attribution.py) on constructed flips."""
import numpy as np

from oracle import flip_attribution as FA
from oracle import oracle as O


def _scene():
    # five well separated boxes + one that overlaps box 0 with IoU just under 0.5
    b = np.array([[0.10, 0.10, 0.30, 0.30], [0.50, 0.50, 0.70, 0.70], [0.10, 0.60, 0.30, 0.80],
                  [0.60, 0.10, 0.80, 0.30], [0.40, 0.05, 0.45, 0.10], [0.10, 0.10, 0.30, 0.1998]], np.float32)[None]
    s = np.array([[0.9, 0.8, 0.7, 0.6, 0.100001, 0.5]], np.float32)
    return b, s


def _nms(b, s):
    return O.nms_padded(b, s, 100, 0.5, 0.1)


def test_equal_selections_give_no_flip():
    b, s = _scene()
    sel, nv = _nms(b, s)
    assert FA.attribute(b, s, sel, nv, b, s, sel, nv) == []


def test_score_threshold_flip_is_attributed():
    b, s = _scene()
    s2 = s.copy()
    s2[0, 4] = 0.0999995                      # the device run drops box 4 (strict > 0.1)
    sel_r, nv_r = _nms(b, s)
    sel_g, nv_g = _nms(b, s2)
    assert nv_r[0] == nv_g[0] + 1
    flips = FA.attribute(b, s, sel_r, nv_r, b, s2, sel_g, nv_g)
    assert len(flips) == 1 and flips[0]["cause"] == "score_threshold" and flips[0]["box"] == 4
    assert flips[0]["margin"] < 2e-6
    assert FA.explained(flips, dscore_max=float(np.abs(s - s2).max()), dbox_max=0.0)
    assert not FA.explained(flips, dscore_max=1e-8, dbox_max=0.0)      # a deviation that small cannot explain it


def test_sort_order_flip_is_attributed():
    b, s = _scene()
    s1, s2 = s.copy(), s.copy()
    s1[0, 2], s1[0, 3] = 0.65, 0.6500001
    s2[0, 2], s2[0, 3] = 0.6500001, 0.65
    sel_r, nv_r = _nms(b, s1)
    sel_g, nv_g = _nms(b, s2)
    flips = FA.attribute(b, s1, sel_r, nv_r, b, s2, sel_g, nv_g)
    assert len(flips) == 1 and flips[0]["cause"] == "sort_order" and {flips[0]["box"], flips[0]["swapped_with"]} == {2, 3}
    assert FA.explained(flips, dscore_max=2e-7, dbox_max=0.0)


def test_iou_threshold_flip_is_attributed():
    b, s = _scene()
    iou = FA._iou(b[0, 0], b[0, 5])
    assert 0.49 < iou < 0.5                    # kept in the reference run
    b2 = b.copy()
    b2[0, 5, 3] += 5e-4                        # a taller box 5: IoU with box 0 crosses 0.5, the device run suppresses it
    assert FA._iou(b2[0, 0], b2[0, 5]) >= 0.5
    sel_r, nv_r = _nms(b, s)
    sel_g, nv_g = _nms(b2, s)
    assert nv_r[0] == nv_g[0] + 1
    flips = FA.attribute(b, s, sel_r, nv_r, b2, s, sel_g, nv_g)
    assert len(flips) == 1 and flips[0]["cause"] == "iou_threshold" and flips[0]["box"] == 5 and flips[0]["suppressor"] == 0
    assert flips[0]["iou_own"] < 0.5 <= flips[0]["iou_other"]
    assert FA.explained(flips, dscore_max=0.0, dbox_max=5e-4)
    assert not FA.explained(flips, dscore_max=0.0, dbox_max=1e-6)


def test_unexplained_difference_is_flagged():
    b, s = _scene()
    sel_r, nv_r = _nms(b, s)
    sel_g = sel_r.copy()
    sel_g[0, 1] = 3                             # a selection no decision of the algorithm produces
    flips = FA.attribute(b, s, sel_r, nv_r, b, s, sel_g, nv_r)
    assert len(flips) == 1 and not FA.explained(flips, 1e-6, 1e-6)
