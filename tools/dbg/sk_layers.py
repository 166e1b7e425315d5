import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt
from yolo_v3_tf2_amd.graph import load_program
from yolo_v3_tf2_amd.weights import synthetic_weights
p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
w = synthetic_weights(p, seed=4321)
S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = torch.from_numpy(np.random.default_rng(1234).random((B, S, S, 3), dtype=np.float32)).cuda()
def run(sk):
    net = rt.Net(p); net.load_weights(w); net.keep_activations(True); net.plan(B, S)
    for slot, o in enumerate(net.conv_ops):
        if o.cin != 3:
            cp = (o.cout + 31) // 32 * 32
            net.set_tile(slot, (33 if cp % 128 == 0 else 34 if cp % 64 == 0 else -1) if sk else (10 if cp % 128 == 0 else 11 if cp % 64 == 0 else -1))
    g = net.forward(x)
    torch.cuda.synchronize()
    outs = {}
    for o in net.conv_ops:
        if o.dst in p.outputs:
            outs[o.conv_index] = g[p.outputs.index(o.dst)].clone().reshape(B, -1)
        else:
            outs[o.conv_index] = net.read_tensor(o.dst, B).reshape(B, -1)
    return outs, net
a, na = run(False)
b, nb = run(True)
for o in na.conv_ops:
    i = o.conv_index
    d = (a[i] - b[i]).abs().max().item(); m = a[i].abs().max().item()
    ho = S // o.out_div
    flag = "  <-- BAD" if d > 1e-4 * max(1, m) else ""
    print(f"conv{i:<3d} k{o.size}s{o.stride} {o.cin:>4d}->{o.cout:<4d} @{ho:<3d} res={int(o.residual>=0)} cat={int(o.src1>=0)} maxdiff {d:.3e} (max {m:.2e}){flag}")
