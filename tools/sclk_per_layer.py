#!/usr/bin/env python3
"""Shader clock held inside every conv launch of the fp32 plan (GPU), next to the launch's duration.

The per-layer table of bench.py --per-layer shows identical convs (3x3 128->256 @52) getting faster through the
network (0.93 -> 0.84 ms in the backbone, 0.74 ms in the head).  This tool tells clock (power management: the chip sits
at its power cap and the clock follows what each layer draws) from kernel efficiency: for each conv it runs --forwards
forwards back to back and, in the last one, lets a steady-state workgroup of that conv's launch stamp s_memtime /
s_memrealtime (y3_net_measure_sclk_conv).  TF/s at the measured clock / (peak at that clock) is the efficiency that is
the kernel's own."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--forwards", type=int, default=100)
    ap.add_argument("--weights-scale", type=float, default=1.0, help="multiply conv weights (data-dependent power probe)")
    a = ap.parse_args()
    B, S = a.batch, a.image_size
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p))
    net.plan(B, S, _lib.Y3_DTYPE_F32)
    x = torch.rand((B, S, S, 3), device="cuda")
    out = net.forward(x)
    torch.cuda.synchronize()
    ms = None
    for _ in range(3):
        m = net.profile_convs(x)
        ms = m if ms is None else np.minimum(ms, m)
    peak_per_mhz = 157.3 / 2400.0      # TF/s per MHz (256 CUs x 4 SIMDs x 64 FLOP/clk: 157.3 TF/s at 2.4 GHz)
    # every conv of ONE forward in the steady state: the last of --forwards back-to-back forwards
    mhz, t0, t1 = net.measure_sclk_all(x, out, forwards=a.forwards)
    stamped = [i for i in range(len(mhz)) if mhz[i] > 0]
    print(f"# steady state: forward {a.forwards} of {a.forwards} back to back; 'ms' = isolated launch (events, min of 3); "
          f"'next' = start of the next stamped launch - start of this one (the real forward's timeline)")
    print(f"{'conv':<6s}{'shape':<28s}{'ms':>8s}{'next ms':>9s}{'TF/s':>8s}{'MHz':>8s}{'of clock peak':>15s}")
    wsum = wt = 0.0
    for k, slot in enumerate(stamped):
        o = net.conv_ops[slot]
        ho = S // o.out_div
        fl = 2.0 * o.size ** 2 * o.cin * o.cout * ho * ho * B
        sig = f"k{o.size}s{o.stride}_c{o.cin}_n{o.cout}_h{ho}_r{int(o.residual >= 0)}"
        nxt = (t0[stamped[k + 1]] - t0[slot]) / 1e3 if k + 1 < len(stamped) else float("nan")
        dur = nxt if nxt == nxt else ms[slot]
        tf = fl / dur / 1e9
        print(f"{o.conv_index:<6d}{sig:<28s}{ms[slot]:8.3f}{nxt:9.3f}{tf:8.1f}{mhz[slot]:8.0f}{tf / (peak_per_mhz * mhz[slot]):15.3f}")
        wsum += mhz[slot] * dur
        wt += dur
    print(f"time-weighted clock {wsum / wt:.0f} MHz over {wt:.2f} ms of stamped launches "
          f"(first to last stamp {(t1[stamped[-1]] - t0[stamped[0]]) / 1e3:.2f} ms)")


if __name__ == "__main__":
    main()
