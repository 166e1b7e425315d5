#!/usr/bin/env python3
"""Where does a bf16 conv tile spend its time?  Needs the diagnostic build (per-workgroup s_memrealtime stamps):
   python yolo-v3-tf2_amd/csrc/build.py --variant yolo-v3-tf2_amd/lib/liby3hip_stamps.so conv_bf16.hip -DY3_PHASE_STAMPS
   Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_stamps.so python tools/phase_stamps.py --cin 128 --cout 256 --s 52 --batch 64
Runs one 3x3 conv (+ shortcut) many times, reads the stamps of the last launch and prints, over all workgroups, the median duration
of each phase; then rebuilds every CU's timeline (HW_ID / XCC_ID) and prints the gap between one workgroup's last stamp and the next
workgroup's first stamp on the same CU (wave teardown + dispatch + kernel-argument load)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402
from tests.helpers import mini_program  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--s", type=int, default=52)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tile", type=int, default=24)
    ap.add_argument("--shortcut", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--size", type=int, default=3, help="3: 3x3 conv cin -> cout (the block's second conv); 1: time the block's 1x1 conv cout -> cin instead")
    a = ap.parse_args()
    lib = C.CDLL(_lib.LIB_PATH)
    f32 = a.dtype == "f32"
    sel, cpy, cap = ("y3_dbg32_select_k", "y3_dbg32_copy_stamps", 32768) if f32 else ("y3_dbg_select_k", "y3_dbg_copy_stamps", 8192)
    if not hasattr(lib, sel):
        sys.exit("this library has no phase stamps: build the -DY3_PHASE_STAMPS variant and set Y3_LIB_PATH")
    # residual block: 1x1 cout -> cin, then the 3x3 cin -> cout under test with the block input as shortcut
    chain = [dict(filters=a.cin, size=1), dict(filters=a.cout, size=3, shortcut=-3 if a.shortcut else 0)]
    if not a.shortcut:
        chain[1].pop("shortcut")
    p = mini_program(a.cout, chain, [dict(filters=64, size=1), dict(filters=64, size=1), dict(filters=64, size=1)])
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p, seed=1))
    slot = 1 if a.size == 3 else 0
    if a.tile >= 0:
        (net.set_tile if f32 else net.set_tile_bf16)(slot, a.tile)
    net.plan(a.batch, a.s, _lib.Y3_DTYPE_F32 if f32 else _lib.Y3_DTYPE_BF16)
    net.set_lanes(1)
    x = torch.randn((a.batch, a.s, a.s, a.cout), device="cuda")
    if not f32:
        x = x.to(torch.bfloat16)
    K = 9 * a.cin if a.size == 3 else a.cout
    assert getattr(lib, sel)(C.c_int(K)) == 0
    for _ in range(a.reps):
        net.forward(x)
    torch.cuda.synchronize()
    W = 8                                             # words per workgroup record
    buf = (C.c_ulonglong * (W * cap))()
    assert getattr(lib, cpy)(buf, C.c_int(W * cap)) == 0
    st_all = np.frombuffer(buf, dtype=np.uint64).reshape(cap, W).astype(np.int64)
    n = int((st_all[:, 0] > 0).sum())                 # stamped workgroups (blockIdx order: a prefix)
    nwg = n
    st = st_all[:n]
    bm = bn = 0
    if a.tile >= 0:
        bm, bn = (_lib.TILES if f32 else _lib.TILES_BF16)[a.tile][0], (_lib.TILES if f32 else _lib.TILES_BF16)[a.tile][1]
    t = st[:, :5] * 10.0 / 1000.0          # us (100 MHz ticks)
    t0 = t[:, 0].min()
    names = ["entry -> addresses ready", "first fetch + barrier", "K loop", "epilogue (to this wave's last store issue)"]
    print(f"{a.dtype}: {n} stamped workgroups" + (" (buffer full)" if n == cap else "") + f", tile {bm}x{bn} (0x0: the plan's choice), K = {K}, "
          f"launch span {t[:, 4].max() - t0:.1f} us")
    for k in range(4):
        d = t[:, k + 1] - t[:, k]
        print(f"  {names[k]:46s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    tot = t[:, 4] - t[:, 0]
    print(f"  {'whole workgroup (entry -> last stamp)':46s} median {np.median(tot):7.2f} us")
    # per-CU timelines: key = (xcc, se, sh, cu) from the registers
    hw, xcc = st[:, 5], st[:, 6] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = xcc * 10000 + se * 100 + sh * 16 + cu
    gaps = []
    per_cu = {}
    for i in np.argsort(t[:, 0]):
        per_cu.setdefault(int(key[i]), []).append(i)
    for k_, idx in per_cu.items():
        for a_, b_ in zip(idx, idx[1:]):
            gaps.append(t[b_, 0] - t[a_, 4])
    gaps = np.array(gaps)
    print(f"  distinct CUs seen: {len(per_cu)}; workgroups per CU: {np.mean([len(v) for v in per_cu.values()]):.2f}")
    # how many workgroups does a CU hold at once?  sweep every CU's [entry, last stamp] intervals
    peak, avg = [], []
    for idx in per_cu.values():
        ev = sorted([(t[i, 0], 1) for i in idx] + [(t[i, 4], -1) for i in idx])
        cur = mx = 0
        area = 0.0
        last_t = ev[0][0]
        for tt, d in ev:
            area += cur * (tt - last_t)
            last_t = tt
            cur += d
            mx = max(mx, cur)
        peak.append(mx)
        avg.append(area / max(1e-9, ev[-1][0] - ev[0][0]))
    print(f"  workgroups resident per CU at once: peak median {int(np.median(peak))} (min {min(peak)}, max {max(peak)}), time-average {np.mean(avg):.2f}")
    if len(gaps) and not f32:      # fp32 tiles: several workgroups are resident per CU, consecutive ones overlap
        print(f"  {'gap on a CU: last stamp -> next entry':46s} median {np.median(gaps):7.2f} us   p10 {np.percentile(gaps, 10):7.2f}   p90 {np.percentile(gaps, 90):7.2f}")
    starts = np.sort(t[:, 0] - t0)
    print(f"  first-round entries spread over {starts[min(255, n - 1)]:.2f} us; last workgroup enters at {starts[-1]:.1f} us")


if __name__ == "__main__":
    main()
