#!/usr/bin/env python3
"""Headline benchmark: images/s of the full YOLOv3 detect path (Darknet-53 + 3 heads -> decode -> NMS ->
packed detections [-> RCCL all-gather]) at 416x416, batch 64 per GPU, fp32, synthetic data.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher: starts the line above as a child process and relays its JSON line)

--gpus N means N ranks on N GPUs or a non-zero exit: a WORLD_SIZE that differs from N, or fewer than N visible GPUs,
stops the run without a number.

One rank per GPU; every rank owns 64 images (weak scaling: the path is per-image independent, the only exchange
is one all-gather of the packed [64,100,7] detections + [64] num_valid per step).  Rank 0 prints ONE JSON line.
A step = one pass of the hot path over one batch already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (guides/MI355X_MICROARCH.md)
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense bf16 matrix peak (256 CU x 4096 FLOP/clk x 2.4 GHz; the ~5 PF figure is 2:1 sparse)


def host_images(B, S, rank):
    """The batch every leg of this run sees (same tensors feed the GPU path and the CPU
    baseline): uniform [0,1) NHWC fp32 from a NumPy generator seeded per rank."""
    import numpy as np
    return np.random.default_rng(1234 + rank).random((B, S, S, 3), dtype=np.float32)


def parity_gate(program, weights, anchors, images_host, device_out, n, M, iou, score):
    """Outside the timed region, before any number is reported: the first n images of the SAME batch through the
    oracle.  Boxes / scores within 1e-4 of the oracle's (north_star's bar); NMS index selection bit-exact on identical
    NMS inputs (the device's own boxes and scores through the oracle's NMS).  Raises on failure."""
    import numpy as np
    from oracle import oracle as O
    from oracle import flip_attribution as FA
    from oracle import torch_ref
    # the gate's images come from BOTH ends of the batch: with two sub-batch lanes (the shipped fp32 plan) image 0 runs in
    # lane 0 and image B-1 in lane 1
    B = len(images_host)
    idx = sorted(set(list(range((n + 1) // 2)) + list(range(B - n // 2, B))))
    gb, gc, gs, gsel, gnv = (t.cpu().numpy()[idx] for t in device_out)
    rb, rc, rs, rsel, rnv = O.detect(program, weights, images_host[idx], anchors, M, iou, score)
    # The bar: 1e-4 absolute wherever |coord| <= 1 (where detections live), 1e-4 * |coord| for the unclipped boxes tens
    # of image widths wide that random-init heads emit (w = exp(tw) * anchor, reference core/yolo_decode_layer.py:23: a
    # 1e-5 summation-order difference in tw is 1e-5 * w in the corner).  The evidence under the second part is measured
    # right here: the same images through a SECOND CPU fp32 implementation (PyTorch-CPU/oneDNN convolutions, same
    # decode/NMS) -- cpu_vs_cpu below.  Two CPU fp32 runs miss an absolute 1e-4 on the wide boxes too.
    dev = torch_ref.box_deviation(gb, rb, gs, rs)
    if not (dev["max_abs_dbox_coords_within_unit_range"] <= 1e-4 and dev["max_dbox_over_max1_abs_coord"] <= 1e-4
            and dev["max_abs_dscore"] <= 1e-4):
        raise SystemExit(f"PARITY GATE FAILED: {dev} (bar 1e-4) -- no number reported")
    s2, n2 = O.nms_padded(gb, gs, M, iou, score)
    if not (np.array_equal(s2, gsel) and np.array_equal(n2, gnv)):
        raise SystemExit("PARITY GATE FAILED: NMS selection differs from the oracle on identical inputs -- no number reported")
    # end-to-end selection: equal, or every difference attributed to ONE near-tie decision (which image / position /
    # box, which threshold, how close) that the measured deviation covers; anything else fails the gate
    flips = FA.attribute(rb, rs, rsel, rnv, gb, gs, gsel, gnv, iou, score)
    if not FA.explained(flips, dev["max_abs_dscore"], dev["max_abs_dbox_coords_within_unit_range"]):
        raise SystemExit(f"PARITY GATE FAILED: end-to-end selection differs without a near-tie: {flips} -- no number reported")
    tb, tc, ts, tsel, tnv = torch_ref.detect(program, weights, images_host[idx], anchors, M, iou, score)
    cpu2 = torch_ref.box_deviation(tb, rb, ts, rs)
    cpu2["selection_equal"] = bool(np.array_equal(tsel, rsel) and np.array_equal(tnv, rnv))
    # how far the GPU sits from the oracle RELATIVE to a second CPU fp32 implementation on the same images is bounded too -- a
    # different summation order over K <= 4608 on the matrix cores may cost a small multiple of what another CPU order costs, not an order of magnitude
    ratio_box = dev["max_abs_dbox_coords_within_unit_range"] / max(cpu2["max_abs_dbox_coords_within_unit_range"], 1e-6)
    ratio_score = dev["max_abs_dscore"] / max(cpu2["max_abs_dscore"], 1e-7)
    if not (ratio_box <= 4.0 and ratio_score <= 4.0):
        raise SystemExit(f"PARITY GATE FAILED: the device is {ratio_box:.1f} x (boxes) / {ratio_score:.1f} x (scores) further from the oracle than a second "
                         f"CPU fp32 implementation (bar 4 x) -- no number reported")
    cpu2["what"] = "oracle/torch_ref.py (PyTorch-CPU convolutions) vs oracle/y3_oracle.c on the same images: the floor of fp32 on this input"
    out = {"images": len(idx), "image_indices": idx}
    out.update(dev)
    out.update({"bar": "1e-4 absolute for |coord| <= 1; 1e-4 * |coord| beyond (the raw deviation of two CPU fp32 runs "
                       f"is {cpu2['max_abs_dbox_raw']:.2e} on the same images)",
                "cpu_vs_cpu": cpu2,
                "gpu_over_cpu_vs_cpu": {"boxes_unit_range": round(ratio_box, 2), "scores": round(ratio_score, 2), "bar": 4.0},
                "nms_index_selection": "bit-exact on identical inputs",
                "end_to_end_selection_equal": len(flips) == 0, "end_to_end_selection_flips": flips,
                "class_argmax_flips": int((rc != gc).sum())})   # arg-max flips between near-equal class probabilities
    return out


def parity_gate_bf16(program, weights, images_host, grids, n):
    """bf16 plans (BASELINE config 5): the head logits of the first n images of the batch against the bf16-emulating oracle
    (rounds to bf16 exactly where the pipeline stores).  Two bf16 pipelines that differ only in the order of their fp32 partial
    sums decorrelate to ~1 ulp rms within ~10 layers (tests/test_oracle.py::test_bf16_free_running_floor), so the bar is that
    floor, measured here on these very images: the oracle against itself with fp64 partial sums.  The kernels must be no further
    from the oracle than twice (relative L2) / three times (max) the oracle is from itself; the distance to the fp32 oracle (what
    bf16 costs) is reported and bounded.  The per-layer one-ulp bar on identical inputs is a test
    (test_bf16_every_layer_teacher_forced_within_one_ulp), not a bench step.  Raises on failure."""
    import numpy as np
    from oracle import oracle as O
    x = images_host[:n]
    got = [g[:n].cpu().numpy().reshape(n, g.shape[1], g.shape[2], -1) for g in grids]
    ref16 = O.forward(program, weights, x, bf16=True)
    ref16b = O.forward(program, weights, x, bf16=True, acc64=True)
    ref32 = O.forward(program, weights, x)
    rel = lambda a, b: float(np.linalg.norm(a.reshape(-1) - b.reshape(-1)) / np.linalg.norm(b.reshape(-1)))   # noqa: E731
    mx = lambda a, b: float(np.abs(a.reshape(-1) - b.reshape(-1)).max())                                      # noqa: E731
    floor_rel = max(rel(a, b) for a, b in zip(ref16b, ref16))
    floor_max = max(mx(a, b) for a, b in zip(ref16b, ref16))
    rel16 = max(rel(g, r) for g, r in zip(got, ref16))
    d16 = max(mx(g, r) for g, r in zip(got, ref16))
    rel32 = max(rel(g, r) for g, r in zip(got, ref32))
    d32 = max(mx(g, r) for g, r in zip(got, ref32))
    out = {"images": n, "what": "head logits (three grids) vs oracle.forward(bf16=True)",
           "rel_l2_vs_bf16_oracle": rel16, "max_abs_vs_bf16_oracle": d16,
           "floor_rel_l2_oracle_vs_oracle": floor_rel, "floor_max_abs_oracle_vs_oracle": floor_max,
           "rel_l2_vs_fp32_oracle": rel32, "max_abs_vs_fp32_oracle": d32,
           "bar": "rel <= 2 x floor and max <= 3 x floor (two bf16 pipelines with different summation orders decorrelate to ~1 ulp rms); "
                  "vs fp32: rel < 3e-2"}
    if not (rel16 <= 2.0 * floor_rel and d16 <= 3.0 * floor_max and rel32 < 3e-2):
        raise SystemExit(f"PARITY GATE FAILED (bf16): {out} -- no number reported")
    return out


def host_cpu_share():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's
    share of a 256-thread host to the job; running 256 threads on it thrashes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    cap = os.environ.get("Y3_CPU_THREADS")
    return max(1, min(n, int(cap))) if cap else n


def cpu_baseline(program, weights, anchors, images_host, budget_s=20.0):
    """The reference's path on the host cores, on a bounded sample of the same batch (kind 'port': TensorFlow is not
    installed, so the reference itself cannot be timed).  Headline: the network on PyTorch-CPU operators (oneDNN
    convolutions -- the kernel family TF 2.8 uses on CPU; oracle/torch_ref.py) + the C decode/score/NMS of the oracle.
    Second field: the whole path through the oracle's naive C loops (what the parity tests compare against)."""
    import numpy as np
    import torch
    cores = int(os.environ.get("OMP_NUM_THREADS", host_cpu_share()))   # fixed at start-up (main)
    torch.set_num_threads(cores)
    from oracle import oracle as O
    from oracle import torch_ref
    S = images_host.shape[1]
    net = torch_ref.TorchNet(program.model_config_file, weights, program.nclasses)

    def run(x):
        grids = net(x)
        return O.yolo_nms(O.yolo_decode(grids, anchors, program.nclasses), 100, 0.5, 0.1)

    run(images_host[:1])                                   # untimed: thread pools, oneDNN primitive creation
    t0 = time.time()
    run(images_host[:2])
    t2 = time.time() - t0
    n = int(max(2, min(len(images_host), 32, 2 * (budget_s * 0.6) // max(t2, 1e-3))))
    n -= n % 2
    t0 = time.time()
    for i in range(0, n, 2):                               # Keras predict() re-batches too; 2 images keep memory flat
        run(images_host[i:i + 2])
    dt = time.time() - t0
    # naive C port end to end, one image
    O.detect(program, weights, images_host[:1], anchors)
    t0 = time.time()
    O.detect(program, weights, images_host[:1], anchors)
    t_naive = time.time() - t0
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"first {n} images of the same {S}x{S} batch, 2 at a time: network on PyTorch-CPU operators "
                      f"(oneDNN conv, {cores} threads; oracle/torch_ref.py) + C decode/score/NMS; TensorFlow is not "
                      f"installed so the reference itself cannot be timed",
            "naive_c_port_value": round(1.0 / t_naive, 4),
            "naive_c_port_sample": f"1 image end to end through oracle/y3_oracle.c (scalar loops, OpenMP, {cores} threads)"}


def launch_ranks(n, rehearse):
    """`python bench.py --gpus N` without a launcher: N ranks through torch.distributed.run as a child process, one per GPU.
    Fails (non-zero, no JSON line) when the box has fewer than N GPUs, unless the gloo rehearsal mode shares them."""
    import socket
    import subprocess
    import torch     # device_count() below does not initialise the GPU
    have = torch.cuda.device_count()
    if have < n and not rehearse:
        print(f"[bench] --gpus {n} needs {n} GPUs, one rank each; this box shows {have} -- no number reported "
              f"(Y3_BENCH_REHEARSE_GLOO=1 rehearses the control flow with ranks sharing GPUs)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching " + " ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if r.returncode != 0:
        print(f"[bench] the {n}-rank run exited with status {r.returncode} -- no number reported", file=sys.stderr)
        return r.returncode
    if len(lines) != 1:
        print(f"[bench] expected ONE JSON line from rank 0, got {len(lines)} -- no number reported", file=sys.stderr)
        return 3
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed TOTAL number of images split over the ranks (parallel.shard_range): the strong-scaling run "
                         "SURVEY.md 8(d) names for config 4 (e.g. 512 over 1/2/4/8 GPUs); 0 = --batch images per GPU (weak)")
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-images", type=int, default=2,
                    help="images of the batch run through the oracle before anything is reported (0: skip the gate)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="concurrent sub-batches per forward (y3_net_set_lanes); 0 = what the tuning table of the mode says")
    ap.add_argument("--dtype", choices=["f32", "f32x3", "f32x2", "bf16"], default="f32",
                    help="conv arithmetic: f32 (headline, fp32 MFMA), f32x3 / f32x2 (fp32-accurate plane-split modes on the "
                         "bf16 / fp16 matrix cores) or bf16 (BASELINE config 5: bf16 MFMA, fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra f32x3 / f32x2 measurements appended to the f32 line")
    ap.add_argument("--graph", action="store_true", help="capture the per-batch pipeline in a HIP graph and replay it")
    ap.add_argument("--collective", choices=["y3", "torch"], default="y3",
                    help="all-gather of the packed detections: y3 = y3_allgather_results (RCCL behind the C ABI, one group on "
                         "the compute stream, graph-capturable); torch = torch.distributed.all_gather_into_tensor (RCCL)")
    ap.add_argument("--per-layer", action="store_true", help="also print the per-conv timing table to stderr")
    ap.add_argument("--no-sclk", action="store_true",
                    help="skip the in-kernel clock measurement (~1 s of extra forwards after the timed region): profiler runs "
                         "(tools/profile.sh) use it so that every conv launch of the run belongs to a counted step")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # --gpus N means N ranks, or no number.  Without a launcher (no WORLD_SIZE in the environment) and N > 1, this process
    # becomes the launcher: it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD
    # (never an exec; nothing here has touched the GPU yet), relays rank 0's one JSON line and exits with the child's status.
    rehearse = os.environ.get("Y3_BENCH_REHEARSE_GLOO") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, rehearse))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks: one rank per GPU, "
                         f"--gpus must equal --nproc-per-node -- no number reported")

    # host threads for the CPU legs (oracle parity gate, cpu_baseline): this process's real CPU share, fixed before any
    # OpenMP runtime starts (torch's and the oracle's both read the environment once)
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cpu_share()))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner) write to fd 1 directly, so
    # fd 1 points at stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import yolo_v3_tf2_amd  # noqa: F401
    from yolo_v3_tf2_amd import runtime
    from yolo_v3_tf2_amd._lib import require_gpu, TILE_NAMES
    from yolo_v3_tf2_amd.core.utils import get_anchors
    from yolo_v3_tf2_amd.graph import load_program
    from yolo_v3_tf2_amd.weights import synthetic_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    require_gpu()
    # Y3_BENCH_REHEARSE_GLOO=1: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (tools/visits:
    # two ranks on the build box's one GPU) -- ranks share devices, the process group is gloo and the gather goes through host
    # copies.  It checks sharding, the rank-order gate, the barriers and the max-over-ranks timing; it is NOT a measurement (the
    # line says so in config.collective) and RCCL itself needs one GPU per rank.
    if not rehearse and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"[bench] rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible: one rank "
                         f"per GPU -- no number reported")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if rehearse else local_rank)
    use_dist = world > 1 or os.environ.get("Y3_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    B, S, nc, M = args.batch, args.image_size, 80, 100
    scaling = "weak"
    if args.global_batch > 0:
        from yolo_v3_tf2_amd.parallel import shard_range
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} must be a multiple of the {world} ranks (equal shards: the "
                             f"gather of the packed rows is one fixed-size RCCL group)")
        b0, b1 = shard_range(args.global_batch, rank, world)
        B, scaling = b1 - b0, "strong"

    program = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), nc)
    weights = synthetic_weights(program, seed=4321)
    anchors = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype(np.float32)
    net = runtime.Net(program)
    net.load_weights(weights)
    from yolo_v3_tf2_amd import _lib as y3lib
    dt_ids = {"f32": y3lib.Y3_DTYPE_F32, "f32x3": y3lib.Y3_DTYPE_F32X3, "f32x2": y3lib.Y3_DTYPE_F32X2,
              "bf16": y3lib.Y3_DTYPE_BF16}
    net.plan(B, S, dt_ids[args.dtype])
    if args.lanes > 0:
        net.set_lanes(args.lanes)
    headline_lanes = int(getattr(net, "lanes", 1))     # the alt measurements below re-plan the net with their own tables
    images_host = host_images(B, S, rank)
    images = torch.from_numpy(images_host).cuda()      # resident in HBM before the timed region starts
    grids = [torch.empty((B, g, g, 3, 5 + nc), device="cuda") for g in net.grid_sizes()]
    from yolo_v3_tf2_amd.parallel import Y3Comm, allgather_detections
    gathered, comm, collective = None, None, "none (1 rank)"
    if use_dist:
        gathered = (torch.empty((world * B, M, 7), dtype=torch.int32, device="cuda"),
                    torch.empty((world * B,), dtype=torch.int32, device="cuda"))
        collective = "torch.distributed.all_gather_into_tensor (RCCL)"
        if rehearse:
            collective = "REHEARSAL: gloo all-gather of host copies, ranks sharing GPUs -- control flow only, not a measurement"
        if args.collective == "y3" and not rehearse:
            try:
                comm = Y3Comm.from_torch_distributed()
                collective = "y3_allgather_results (RCCL group behind the C ABI, on the compute stream)"
            except Exception as e:   # reported in the JSON line; the exchange itself still runs over RCCL
                print(f"[bench] y3_comm unavailable ({e}); using torch.distributed for the all-gather", file=sys.stderr)
                collective += f" -- y3_comm init failed: {e}"
            # every rank must take the same route
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
                collective = "torch.distributed.all_gather_into_tensor (RCCL) -- y3_comm init failed on another rank"

    def gather_torch(packed, nv):
        if not rehearse:
            return allgather_detections(packed, nv, out=gathered)   # torch.distributed (RCCL) when world > 1
        hp, hn = allgather_detections(packed.cpu(), nv.cpu())        # gloo on host copies
        gathered[0].copy_(hp)
        gathered[1].copy_(hn)
        return gathered

    if use_dist:
        # fail loud and early: one untimed gather of rank-stamped rows through the very route the timed loop uses; every
        # rank checks that block r of the result carries r (a communicator that mixes ranks up, or a route that silently
        # returns the local rows, stops the run here instead of producing a number)
        stamp = (torch.full((B, M, 7), rank, dtype=torch.int32, device="cuda"),
                 torch.full((B,), rank, dtype=torch.int32, device="cuda"))
        got = comm.allgather(*stamp, out=gathered) if comm is not None else gather_torch(*stamp)
        torch.cuda.synchronize()
        want = torch.arange(world, dtype=torch.int32, device="cuda").repeat_interleave(B)
        if not (torch.equal(got[1], want) and torch.equal(got[0][:, 0, 6], want) and torch.equal(got[0][:, M - 1, 0], want)):
            raise SystemExit(f"[bench] rank {rank}: the all-gather did not return rank-ordered rows ({collective})")

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    last = {}

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        # conv program with the head convs decoding their own tiles (y3_net_forward_decode: the grids are neither written nor read
        # back); the events bracket it, so roofline.achieved counts the conv FLOPs against conv stack + fused decode
        bboxes, cls, scores = net.forward_decode(images, anchors)
        if i is not None:
            ev[i][1].record()
        sel, nv = runtime.nms_padded(bboxes, scores, M, 0.5, 0.1)
        packed = runtime.pack_detections(bboxes, cls, scores, sel, nv)
        last["tuple"] = (bboxes, cls, scores, sel, nv)
        if comm is not None:
            return comm.allgather(packed, nv, out=gathered)     # one RCCL group enqueued by liby3hip.so
        if use_dist:
            return gather_torch(packed, nv)
        return allgather_detections(packed, nv, out=gathered)   # single process: returns its inputs

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        out = step()
    # parity gate: nothing is timed or reported unless the path's results on this very batch
    # match the oracle; fp32-accurate modes only (bf16 is measured against its own bar in tests/)
    parity = None
    fused_equal = None
    if rank == 0 and args.parity_images > 0:
        # the route the loop times (y3_net_forward_decode: the head convs decode their own tiles) against the composed route
        # (forward into grids + y3_yolo_decode_scores) on this very plan and batch: bit-identical by contract, checked here at the
        # bench's geometry for both dtypes (the bf16 gate below compares head logits, which the timed route never writes)
        fb, fc, fs = net.forward_decode(images, anchors)
        cb, cc, cs = runtime.yolo_decode_scores(net.forward(images, out=grids), anchors, nc)
        torch.cuda.synchronize()
        fused_equal = bool(torch.equal(fb, cb) and torch.equal(fc, cc) and torch.equal(fs, cs))
        if not fused_equal:
            raise SystemExit("PARITY GATE FAILED: y3_net_forward_decode differs from y3_net_forward + y3_yolo_decode_scores on the bench's plan -- no number reported")
        del fb, fc, fs, cb, cc, cs
    if rank == 0 and args.parity_images > 0 and args.dtype != "bf16":
        torch.cuda.synchronize()
        parity = parity_gate(program, weights, anchors, images_host, last["tuple"], min(args.parity_images, B), M, 0.5, 0.1)
    elif rank == 0 and args.parity_images > 0:
        net.forward(images, out=grids)       # the bf16 gate compares head logits: one forward that writes the grids
        torch.cuda.synchronize()
        parity = parity_gate_bf16(program, weights, images_host, grids, min(args.parity_images, B, 1))   # one image: three oracle passes
    if use_dist:
        dist.barrier()
    for _ in range(2):      # the gate left the GPU idle for seconds: bring clocks and caches back before the timed region
        out = step()
    graph = None
    if args.graph and (world == 1 or comm is not None):
        # one replay = one step; the conv-stack events are recorded inside the captured stream once
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        graph.replay()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            out = step(i)
    fence()
    dt = time.perf_counter() - t0
    # shader clock under this load, measured in-kernel after the timed region (sysfs / rocm-smi report the DPM level, not
    # the clock the chip holds at its power cap): ~1 s of forwards back to back, the last one's stem kernel is stamped
    sclk_mhz, sclk_src = None, None
    if rank == 0 and args.no_sclk:
        sclk_src = "not measured (--no-sclk)"
    elif rank == 0:
        try:
            fw = max(10, int(1000.0 / max(1.0, dt / args.steps * 1e3)))
            mhz, t_start, t_end = net.measure_sclk_all(images, grids, forwards=fw)
            idx = [i for i in range(len(mhz)) if mhz[i] > 0]
            if len(idx) > 1:
                # every conv launch of the last forward carries stamps: weight each launch's clock with its duration on the
                # chip's own timeline (start of the next stamped launch - its start; the last one: its stamped workgroup)
                dur = [(t_start[idx[k + 1]] if k + 1 < len(idx) else t_end[i]) - t_start[i] for k, i in enumerate(idx)]
                sclk_mhz = float(sum(mhz[i] * d for i, d in zip(idx, dur)) / sum(dur))
                sclk_src = (f"time-weighted mean over the {len(idx)} conv launches of one forward ({min(mhz[i] for i in idx):.0f}-"
                            f"{max(mhz[i] for i in idx):.0f} MHz), each from s_memtime / s_memrealtime stamps of a steady-state "
                            "workgroup, after ~1 s of back-to-back forwards (y3_net_measure_sclk_all; sysfs and rocm-smi "
                            "report the DPM level, not the clock held at the power cap)")
            else:
                sclk_mhz = float(mhz[idx[0]])
                sclk_src = ("s_memtime / s_memrealtime stamps inside the fused stem launch (the only kernel of this plan that "
                            "carries stamps) after ~1 s of back-to-back forwards (y3_net_measure_sclk_all)")
        except runtime.Y3Error as e:
            sclk_src = f"not measured: {e}"
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nv_mean = float(out[1].float().mean().item())

    if graph is not None:   # events cannot be read out of a replayed graph: time the conv stack eagerly afterwards
        for i in range(args.steps):
            ev[i][0].record()
            net.forward_decode(images, anchors)
            ev[i][1].record()
        torch.cuda.synchronize()
    conv_ms = sorted(a.elapsed_time(b) for a, b in ev)
    conv_ms_mean = sum(conv_ms) / len(conv_ms)
    flops_step = net.flops_per_image() * B
    achieved = flops_step / (conv_ms_mean * 1e-3) / 1e12
    peak = F32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    # partial products issued per fp32 product: six bf16 (three planes) or three fp16 (two planes)
    mfma_flops_factor = {"f32x3": 6.0, "f32x2": 3.0}.get(args.dtype, 1.0)

    # roofline.traffic: PMC-derived HBM bytes per conv-stack pass, measured by tools/profile.sh on this workload -- taken from
    # the newest profiles/rNN_traffic_<dtype>_b<B>_s<S>.json.  The line names the round of that file (traffic_round) and calls
    # it stale when a later round's kernel profile exists (the kernels changed after the bytes were counted).
    import glob
    import re
    traffic, traffic_source, traffic_round = None, None, None
    rounds = sorted((int(m.group(1)), q) for q in glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{args.dtype}_b{B}_s{S}.json"))
                    for m in [re.match(r"r(\d+)_", os.path.basename(q))] if m)
    if rounds:
        traffic_round, tf_path = rounds[-1]
        with open(tf_path) as f:
            traffic = json.load(f).get("conv_stack_hbm_bytes_per_step")
        newest = max([int(m.group(1)) for q in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.txt"))
                      for m in [re.match(r"r(\d+)_", os.path.basename(q))] if m] or [traffic_round])
        traffic_source = (f"{os.path.relpath(tf_path, ROOT)}: rocprofv3 --pmc passes of tools/profile.sh on this workload in an "
                          f"earlier run of round {traffic_round} (FETCH_SIZE doubled per the guide + WRITE_SIZE), NOT measured in this run")
        if newest > traffic_round:
            traffic_source = (f"STALE: counted in round {traffic_round}, the newest kernel profile is round {newest}'s (other kernels "
                              f"since) -- ") + traffic_source
    # Extra information on the default (f32) line: the same workload in the fp32-accurate three-plane mode (bf16 matrix
    # cores, same parity tests as f32).  Not the headline value.
    alts = {}
    if args.dtype == "f32" and not args.no_alt and graph is None:
        for tag, dt_id, factor, what in (
                ("f32x3", y3lib.Y3_DTYPE_F32X3, 6.0, "3 bf16 planes per value, 6 bf16 MFMAs per fp32 product"),
                ("f32x2", y3lib.Y3_DTYPE_F32X2, 3.0, "2 fp16 planes per value (2^-22 representation), 3 fp16 MFMAs per fp32 product")):
            try:
                net.plan(B, S, dt_id)
            except runtime.Y3Error as e:     # e.g. a three-plane tensor of 64 x 608^2 images exceeds 32-bit buffer offsets
                alts[tag] = {"dtype": tag, "error": str(e)}
                continue
            for _ in range(2):
                step()
            fence()
            ta = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            dta = time.perf_counter() - ta
            if use_dist:
                tt = torch.tensor([dta], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dta = float(tt.item())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                net.forward_decode(images, anchors)
            e1.record()
            torch.cuda.synchronize()
            cms = e0.elapsed_time(e1) / args.steps
            alts[tag] = {"dtype": f"{tag} (fp32-accurate, same parity tests as f32: {what})",
                         "value": round(world * B * args.steps / dta, 2), "unit": "images/s",
                         "ms_per_step": round(dta / args.steps * 1e3, 3),
                         "conv_tflops_algorithmic": round(flops_step / (cms * 1e-3) / 1e12, 2),
                         "mfma_issued_frac_of_16bit_peak": round(factor * flops_step / (cms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        line = {
            "metric": f"images/sec at {S}x{S} batch={B}" + (f" (global batch {world * B})" if world > 1 else "")
                      + f", {args.dtype} (full YOLOv3 detect: conv stack + decode + NMS)",
            "value": round(world * B * args.steps / dt, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (uniform [0,1) images from numpy default_rng(1234+rank), seeded random-init weights; no checkpoint ships with the reference)",
            "config": {"workload": f"full YOLOv3 detect, {B} images/GPU, {S}x{S}, 80 classes, { {'f32': 'fp32', 'f32x2': 'fp16 (two-plane fp32)', 'f32x3': 'bf16 (three-plane fp32)'}.get(args.dtype, 'bf16') } MFMA conv, "
                                   f"decode + class-agnostic NMS (max 100, iou 0.5, score 0.1), packed detections"
                                   + (", RCCL all-gather" if use_dist else ""),
                       # rccl_ranks: ranks that really exchanged over RCCL in this run (0 for one rank without a collective AND for
                       # the gloo rehearsal, which never touches RCCL; "rehearsal" says which of the two it was)
                       "collective": collective, "rccl_ranks": world if (use_dist and not rehearse) else 0,
                       "rehearsal": bool(use_dist and rehearse),
                       "rank_order_checked": bool(use_dist), "hip_graph": graph is not None,
                       "lanes": headline_lanes,   # concurrent sub-batches per forward (tuning table / --lanes)
                       "global_batch": world * B, "image_size": S, "parallelism": f"dp{world}",
                       "mean_num_valid": round(nv_mean, 2)},
            "roofline": {
                "bound": "mfma", "kernel": ("conv stack (fused stem kernel = conv0..2 in one launch + 69 x conv_f32_mfma + 3 x conv_head_decode_f32 (head 1x1 + decode) launches per step and lane)" if args.dtype == "f32"
                                            else "conv stack (fused stem kernel = conv0..2 in one launch + 72 x conv_bf16_mfma launches per step and lane, the three heads decoding in their epilogue)" if args.dtype == "bf16"
                                            else f"conv stack (74 x conv_f32x3_mfma{'<2 planes>' if args.dtype == 'f32x2' else ''} launches + 1 first-layer conv per step)"),
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved * mfma_flops_factor / peak, 4), "traffic": traffic, "traffic_round": traffic_round, "traffic_source": traffic_source,
                "sclk_mhz": round(sclk_mhz, 1) if sclk_mhz else None, "sclk_source": sclk_src,
                "frac_of_clock_limited_peak": (round(achieved * mfma_flops_factor / (peak * sclk_mhz / 2400.0), 4)
                                               if sclk_mhz else None),
                "issued_over_algorithmic_flops": mfma_flops_factor,
                "flops_per_launch": flops_step, "ms_per_launch": round(conv_ms_mean, 3),
                "ms_median": round(conv_ms[len(conv_ms) // 2], 3),
            },
        }
        line["parity_checked"] = parity["images"] if parity else 0
        if parity:
            parity["timed_route_equals_composed_route"] = fused_equal
            line["parity"] = parity
        for tag, alt in alts.items():
            line["alt_" + tag] = alt
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(program, weights, anchors, images_host)
        if args.per_layer:
            net.plan(B, S, dt_ids[args.dtype])       # the alt measurements re-planned the net: back to the headline mode
            ms = net.profile_convs(images)
            fused_fl = sum(2.0 * o.size * o.size * o.cin * o.cout * (S // o.out_div) ** 2 * B
                           for o, t in zip(net.conv_ops, ms) if t <= 0)
            carrier = next((o.conv_index for o, t in zip(net.conv_ops, ms) if t > 0), -1) if fused_fl else -1
            for o, t in zip(net.conv_ops, ms):
                if t <= 0:      # conv0 / the 1x1 third layer of the fused stem: their work is inside conv1's launch
                    print(f"conv{o.conv_index:<3d} {o.size}x{o.size}/{o.stride} {o.cin:>4d}->{o.cout:<4d}  (inside the fused stem launch, timed with conv{carrier})", file=sys.stderr)
                    continue
                ho = S // o.out_div
                fl = 2.0 * o.size * o.size * o.cin * o.cout * ho * ho * B + (fused_fl if o.conv_index == carrier else 0.0)
                print(f"conv{o.conv_index:<3d} {o.size}x{o.size}/{o.stride} {o.cin:>4d}->{o.cout:<4d} @{ho:<3d} "
                      f"{t:8.3f} ms {fl / t / 1e9:8.1f} TF/s" + ("  (fused stem: incl. the layers timed with it)" if o.conv_index == carrier else ""), file=sys.stderr)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
