#!/usr/bin/env python3
"""Upper bound for "the residual block's 1x1 and 3x3 in ONE kernel" (VERDICT r04 #1), measured before any kernel is written.

The fused kernel would remove the 1x1 LAUNCH of every residual block (its prologue / epilogue / partially filled rounds, the write
of the middle tensor and its 9-tap read) and pay the 1x1's FLOPs (x the halo recompute) inside the 3x3 launch.  This tool measures
the first half exactly and for free: programs in which the blocks of a stage have NO 1x1 launch at all -- every block of the stage
reads the middle tensor the stage's FIRST block produced (same shape; wrong values, timing only) -- against the shipped program, same
process, alternating.  The saving per dropped launch x the stage's block count is what a fusion with a ZERO-cost phase 1 would return;
the real kernel returns that minus its phase 1 (printed beside it from the stage's FLOPs at the rate its 3x3 launches run at).

    python tools/gate_block_fusion.py [--dtype bf16] [--batch 128] [--rounds 3]
"""
import argparse
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import ConvOp, load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def residual_blocks(p):
    """[(index of the 1x1 op, index of the 3x3 op, output spatial divisor)] for every 1x1 -> 3x3 + shortcut pair."""
    out = []
    ops = p.ops
    for i in range(len(ops) - 1):
        a, b = ops[i], ops[i + 1]
        if (isinstance(a, ConvOp) and isinstance(b, ConvOp) and a.size == 1 and b.size == 3 and b.stride == 1 and b.src0 == a.dst
                and b.residual == a.src0 and a.src1 < 0 and a.residual < 0):
            out.append((i, i + 1, b.out_div))
    return out


def drop_stage(p, divs, free_shortcut=False):
    """A copy of the program in which the blocks at the given spatial divisors keep only their stage's first 1x1.  free_shortcut: the
    3x3 convs of those blocks (the first one too) also lose their shortcut operand -- the fused kernel takes it from the input patch
    it has in LDS anyway, so its separate read disappears as well (wrong values, timing only)."""
    q = copy.deepcopy(p)
    blocks = residual_blocks(q)
    first = {}
    dropped = []
    for i1, i3, div in blocks:
        if div not in divs:
            continue
        if free_shortcut:
            q.ops[i3].residual = -1
        if div not in first:
            first[div] = q.ops[i1].dst
            continue
        q.ops[i3].src0 = first[div]
        dropped.append(i1)
    q.ops = [o for i, o in enumerate(q.ops) if i not in set(dropped)]
    return q, len(dropped)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--free-shortcut", action="store_true", help="bound for the weight-resident fused block at 104^2: also drop the 3x3 convs' shortcut reads")
    a = ap.parse_args()
    dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16}[a.dtype]
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    w = synthetic_weights(p)
    S, B = a.image_size, a.batch
    blocks = residual_blocks(p)
    per_div = {}
    for _, _, d in blocks:
        per_div[d] = per_div.get(d, 0) + 1
    print(f"# {a.dtype}, {B} x {S}^2, tuning table of the plan; residual blocks per stage (grid size: count): "
          + ", ".join(f"{S // d}: {n}" for d, n in sorted(per_div.items())))
    variants = [("shipped", None)] + [(f"no 1x1 launches @{S // d}", {d}) for d in sorted(per_div) if per_div[d] > 1] + [("no 1x1 launches, all stages", set(per_div))]
    if a.free_shortcut:   # the HBM-bound early stages only: 1x1 launches gone AND the 3x3's shortcut read gone
        variants = [("shipped", None)] + [(f"no 1x1 launch, no shortcut read @{S // d}", {d}) for d in sorted(per_div) if per_div[d] > 1 and S // d >= 104]
    nets = []
    x = torch.rand((B, S, S, 3), device="cuda")
    for name, divs in variants:
        q, nd = (p, 0) if divs is None else drop_stage(p, divs, a.free_shortcut)
        net = runtime.Net(q)
        net.load_weights(w)
        net.plan(B, S, dt)
        g = [torch.empty((B, s, s, 3, 85), device="cuda") for s in net.grid_sizes()]
        for _ in range(3):
            net.forward(x, out=g)
        nets.append((name, net, g, nd, []))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds):
        for name, net, g, nd, res in nets:
            for _ in range(3):
                net.forward(x, out=g)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                net.forward(x, out=g)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / a.iters)
            print(f"round {r}  {name:34s} {res[-1]:8.4f} ms", flush=True)
    base = min(nets[0][4])
    print(f"# conv stack, min over rounds; saving = what the dropped launches cost the step; 'all blocks' scales it to every block of the stage")
    for name, net, g, nd, res in nets:
        m = min(res)
        if nd == 0:
            print(f"{name:34s} {m:8.4f} ms")
            continue
        n_all = sum(per_div[d] for d in per_div if name.endswith("all stages") or name.endswith(f"@{S // d}"))
        sav = base - m
        print(f"{name:34s} {m:8.4f} ms   {nd:2d} launches dropped: -{sav:6.4f} ms = {100 * sav / base:5.2f} % of the step; all {n_all} blocks: {100 * sav / base * n_all / nd:5.2f} %")


if __name__ == "__main__":
    main()
