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
S, B = 96, 3
x = torch.from_numpy(np.random.default_rng(1234).random((B, S, S, 3), dtype=np.float32)).cuda()
net = rt.Net(p); net.load_weights(w); net.plan(B, S)
for slot, o in enumerate(net.conv_ops):
    if o.cin != 3:
        cp = (o.cout + 31) // 32 * 32
        net.set_tile(slot, 33 if cp % 128 == 0 else 34 if cp % 64 == 0 else -1)
got = [g.clone() for g in net.forward(x)]
nd = 0
for r in range(30):
    g = net.forward(x)
    torch.cuda.synchronize()
    d = max(float((a - b).abs().max()) for a, b in zip(got, g))
    if d != 0: nd += 1; print("eager rep", r, "maxdiff", d)
print("eager nondeterministic reps:", nd)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph(); outs = [torch.empty_like(g) for g in got]
with torch.cuda.stream(st):
    net.forward(x, out=outs)
    with torch.cuda.graph(graph, stream=st):
        net.forward(x, out=outs)
torch.cuda.current_stream().wait_stream(st)
for r in range(10):
    for o in outs: o.zero_()
    graph.replay(); torch.cuda.synchronize()
    d = [float((a - b).abs().max()) for a, b in zip(got, outs)]
    print("graph rep", r, "maxdiff per head", d)
# back-to-back eager without sync
for r in range(5):
    gs = [net.forward(x) for _ in range(5)]
    torch.cuda.synchronize()
    print("b2b", r, [max(float((a - b).abs().max()) for a, b in zip(got, g)) for g in gs])
