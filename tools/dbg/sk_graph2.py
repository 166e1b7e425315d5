import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt
from yolo_v3_tf2_amd.graph import load_program
from yolo_v3_tf2_amd.weights import synthetic_weights
from yolo_v3_tf2_amd.core.utils import get_anchors
p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
w = synthetic_weights(p, seed=4321)
anchors = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype(np.float32)
S, B = 96, 3
x = torch.from_numpy(np.random.default_rng(1234).random((B, S, S, 3), dtype=np.float32)).cuda()
variant = sys.argv[1] if len(sys.argv) > 1 else "a"
net = rt.Net(p); net.load_weights(w); net.plan(B, S)
for slot, o in enumerate(net.conv_ops):
    if o.cin != 3:
        cp = (o.cout + 31) // 32 * 32
        net.set_tile(slot, 33 if cp % 128 == 0 else 34 if cp % 64 == 0 else -1)
got = [g.clone() for g in net.forward(x)]
again = net.forward(x)
print("again equal", all(torch.equal(a, b) for a, b in zip(got, again)))
if variant in ("b", "c"):
    bb, cls, sc = rt.yolo_decode_scores(got, anchors, 80)
    if variant == "c":
        sel, nv = rt.nms_padded(bb, sc, 100, 0.5, 0.1)
    torch.cuda.synchronize()
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph(); outs = [torch.empty_like(g) for g in got]
with torch.cuda.stream(st):
    net.forward(x, out=outs)
    with torch.cuda.graph(graph, stream=st):
        net.forward(x, out=outs)
torch.cuda.current_stream().wait_stream(st)
for r in range(3):
    for o in outs: o.zero_()
    graph.replay(); torch.cuda.synchronize()
    print("graph rep", r, "maxdiff", [float((a - b).abs().max()) for a, b in zip(got, outs)], "outs absmax", [float(o.abs().max()) for o in outs])
e = net.forward(x)
print("eager after", [float((a - b).abs().max()) for a, b in zip(got, e)])
