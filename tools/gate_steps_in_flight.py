#!/usr/bin/env python3
"""Gate: what does keeping D batches in flight return?  D nets (own arenas, own detect scratch), each on its own stream, every
step = y3_net_detect of one whole batch (conv program -> fused decode -> NMS -> pack); step i goes to net i mod D, so consecutive
steps overlap: the memory-bound early layers, the NMS tail and the partially filled last rounds of one batch run under the
MFMA-bound layers of the other.  K steps between two device-wide fences, same K for every D; results of every net compared
with the serial run's, bit for bit.
   python tools/gate_steps_in_flight.py --dtype bf16 --batch 128 --graph --depths 1,2,3 --offsets 0,0.5
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--depths", default="1,2")
    ap.add_argument("--offsets", default="0,0.5", help="start offset of net d as a fraction d*offset of the serial step time")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=0, help="sub-batch lanes inside every net (0: the tuning table's)")
    a = ap.parse_args()
    import numpy as np
    import torch
    import yolo_v3_tf2_amd  # noqa: F401
    from yolo_v3_tf2_amd import runtime, _lib
    from yolo_v3_tf2_amd.core.utils import get_anchors
    from yolo_v3_tf2_amd.graph import load_program
    from yolo_v3_tf2_amd.weights import synthetic_weights

    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    w = synthetic_weights(p, seed=4321)
    anchors = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype(np.float32)
    B, S = a.batch, a.image_size
    dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16}[a.dtype]
    depths = [int(v) for v in a.depths.split(",")]
    D = max(depths)
    x = torch.from_numpy(np.random.default_rng(1234).random((B, S, S, 3), dtype=np.float32)).cuda()
    nets, streams, graphs, outs = [], [], [], []
    for d in range(D):
        n = runtime.Net(p)
        n.load_weights(w)
        n.plan(B, S, dt)
        if a.lanes > 0:
            n.set_lanes(a.lanes)
        nets.append(n)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    for d in range(D):
        with torch.cuda.stream(streams[d]):
            for _ in range(3):
                o = nets[d].detect(x, anchors, 100, 0.5, 0.1)
            g = None
            if a.graph:
                streams[d].synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[d]):
                    o = nets[d].detect(x, anchors, 100, 0.5, 0.1)
            graphs.append(g)
            outs.append(o)
        torch.cuda.synchronize()

    def enqueue(d):
        with torch.cuda.stream(streams[d]):
            if graphs[d] is not None:
                graphs[d].replay()
            else:
                outs[d] = nets[d].detect(x, anchors, 100, 0.5, 0.1)

    # reference result: net 0 alone
    enqueue(0)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs[0]]
    # cycles of torch.cuda._sleep per ms
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)

    def run(depth, offset, serial_ms):
        torch.cuda.synchronize()
        for d in range(depth):
            enqueue(d)      # clocks / caches
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if offset > 0 and serial_ms:
            for d in range(1, depth):
                with torch.cuda.stream(streams[d]):
                    torch.cuda._sleep(int(cyc_per_ms * serial_ms * offset * d))
        for i in range(a.steps):
            enqueue(i % depth)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        ok = all(torch.equal(outs[d][0], ref[0]) and torch.equal(outs[d][1], ref[1]) for d in range(depth))
        return dt_ / a.steps * 1e3, ok

    print(f"# {a.dtype} {B} x {S}^2, lanes per net {nets[0].lanes}, graph replay {a.graph}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default (4)')}")
    serial = None
    for r in range(a.rounds):
        for depth in depths:
            for off in ([0.0] if depth == 1 else [float(v) for v in a.offsets.split(",")]):
                ms, ok = run(depth, off, serial)
                if depth == 1:
                    serial = ms if serial is None else min(serial, ms)
                print(f"round {r} depth {depth} offset {off:.2f}: {ms:.3f} ms per step  {B / ms * 1e3:9.1f} images/s  "
                      f"outputs {'bit-identical to the serial run' if ok else 'DIFFER'}", flush=True)


if __name__ == "__main__":
    main()
