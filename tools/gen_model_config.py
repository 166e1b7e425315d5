#!/usr/bin/env python3
"""Generate the YOLOv3 model description + COCO data files consumed by the host mirror.

The files are written in the schema the reference's graph builder reads
(reference: core/parse_model.py:248-314 -- `layers_config` lists of
convolutional / shortcut / route / upsample / yolo entries, and a `model.yaml`
with `sub_models_configs` that wires seven sub-models together), so a detect
config written for the reference keeps working.  Nothing is copied from the
reference tree: the topology below is the published Darknet-53 / YOLOv3
structure, emitted from a compact description, and `grid_size` is derived
from the image size (the reference hard-codes 13/26/52).

Usage: python tools/gen_model_config.py [--out config/models/yolov3] [--image-size 416]
"""
import argparse
import os

import numpy as np


def conv(filters, size, stride=1, bn=True, act="leaky"):
    d = {"type": "convolutional"}
    if bn:
        d["batch_normalize"] = 1
    d.update({"filters": filters, "size": size, "stride": stride, "pad": 1, "activation": act})
    return d


def shortcut():
    return {"type": "shortcut", "from": -3, "activation": "linear"}


def route(layers=None, inputs=None):
    src = {}
    if layers is not None:
        src["layers"] = list(layers)
    if inputs is not None:
        src["inputs"] = list(inputs)
    return {"type": "route", "source": src}


def res_stage(ch, n):
    """Stride-2 3x3 down-sampling conv to `ch` channels then n residual units."""
    out = [conv(ch, 3, 2)]
    for _ in range(n):
        out += [conv(ch // 2, 1), conv(ch, 3), shortcut()]
    return out


def backbone():
    layers = [route(inputs=[0]), conv(32, 3)]
    for ch, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        layers += res_stage(ch, n)
    return layers


def neck(ch, first):
    """Five alternating 1x1(ch) / 3x3(2ch) convs; non-first necks are fed by
    [lateral 1x1 -> upsample] (+) backbone feature map."""
    if first:
        layers = [route(inputs=[0])]
    else:
        return None
    for i in range(5):
        layers.append(conv(ch, 1) if i % 2 == 0 else conv(2 * ch, 3))
    return layers


def neck_lateral(ch, lateral_input, skip_input):
    layers = [route(inputs=[lateral_input]), conv(ch, 1), {"type": "upsample", "stride": 2},
              route(layers=[-1], inputs=[skip_input])]
    for i in range(5):
        layers.append(conv(ch, 1) if i % 2 == 0 else conv(2 * ch, 3))
    return layers


def head(ch, grid):
    return [route(inputs=[0]), conv(2 * ch, 3),
            conv("3*(2+2+1+nclasses)", 1, bn=False, act="linear"),
            {"type": "yolo", "grid_size": grid}]


def dump_layers(path, layers):
    # hand-rolled emitter: stable key order, readable diffs
    with open(path, "w") as f:
        f.write("---\nlayers_config:\n")
        for l in layers:
            first = True
            for k, v in l.items():
                lead = "- " if first else "  "
                first = False
                if isinstance(v, dict):
                    f.write(f"{lead}{k}:\n")
                    for kk, vv in v.items():
                        f.write(f"    {kk}:\n")
                        for x in vv:
                            f.write(f"    - {x}\n")
                else:
                    f.write(f"{lead}{k}: {v}\n")
            f.write("\n")


def dump_model(path, outdir, grids):
    subs = [
        ("backbone", None, [-39, -14, -1]),
        ("neck0", [("backbone", 2)], [-1]),
        ("head0", [("neck0", 0)], [-1]),
        ("neck1", [("backbone", 1), ("neck0", 0)], [-1]),
        ("head1", [("neck1", 0)], [-1]),
        ("neck2", [("neck1", 0), ("backbone", 0)], [-1]),
        ("head2", [("neck2", 0)], [-1]),
    ]
    with open(path, "w") as f:
        f.write("---\ndecay_factor: 0.0005\noutput_stage: head\ngrid_sizes:\n")
        for g in grids:
            f.write(f"- {g}\n")
        f.write("\nsub_models_configs:\n")
        for name, srcs, outs in subs:
            f.write(f"- name: {name}\n")
            if srcs:
                f.write("  inputs:\n    source:\n")
                for s, e in srcs:
                    f.write(f"    - name: {s}\n      entry_index: {e}\n")
            f.write(f"  layers_config_file: {outdir}/{name}.yaml\n  outputs_layers:\n")
            for o in outs:
                f.write(f"  - {o}\n")
            f.write("\n")


COCO_NAMES = """person bicycle car motorbike aeroplane bus train truck boat traffic_light fire_hydrant
stop_sign parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella
handbag tie suitcase frisbee skis snowboard sports_ball kite baseball_bat baseball_glove skateboard
surfboard tennis_racket bottle wine_glass cup fork knife spoon bowl banana apple sandwich orange broccoli
carrot hot_dog pizza donut cake chair sofa pottedplant bed diningtable toilet tvmonitor laptop mouse remote
keyboard cell_phone microwave oven toaster sink refrigerator book clock vase scissors teddy_bear hair_drier
toothbrush""".split()

# published YOLOv3 COCO anchors in pixels of a 416x416 input, small -> large
ANCHORS_PX = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="config/models/yolov3")
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--data", default="datasets/coco2012")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    os.makedirs(a.data, exist_ok=True)
    grids = [a.image_size // 32, a.image_size // 16, a.image_size // 8]
    dump_layers(f"{a.out}/backbone.yaml", backbone())
    dump_layers(f"{a.out}/neck0.yaml", neck(512, True))
    dump_layers(f"{a.out}/neck1.yaml", neck_lateral(256, 1, 0))
    dump_layers(f"{a.out}/neck2.yaml", neck_lateral(128, 0, 1))
    for i, (ch, g) in enumerate(zip((512, 256, 128), grids)):
        dump_layers(f"{a.out}/head{i}.yaml", head(ch, g))
    dump_model(f"{a.out}/model.yaml", a.out, grids)
    # anchors: three rows per scale, LARGEST scale first (reference: core/utils.py:31-37 reshapes
    # row-major to [scale,3,2]; datasets/coco2012/anchors.txt lists 116x90.. first), normalised by 416
    rows = ANCHORS_PX[6:9] + ANCHORS_PX[3:6] + ANCHORS_PX[0:3]
    with open(f"{a.data}/anchors.txt", "w") as f:
        for w, h in rows:
            f.write(f"{np.float32(w / 416.0):.8f}, {np.float32(h / 416.0):.8f}\n")
    with open(f"{a.data}/coco.names", "w") as f:
        for n in COCO_NAMES:
            f.write(n.replace("_", " ") + "\n")
    # a synthetic RGBA sample image (the reference's girl.png is its own data file and is not copied): same kind of
    # input -- non-square, with alpha -- for the single-image plumbing config
    from PIL import Image, ImageDraw
    os.makedirs(f"{a.data}/images", exist_ok=True)
    rng = np.random.default_rng(2022)
    base = (rng.random((333, 406, 4)) * 60 + 90).astype(np.uint8)
    base[..., 3] = 255
    im = Image.fromarray(base, "RGBA")
    d = ImageDraw.Draw(im)
    d.ellipse([60, 40, 220, 300], fill=(200, 120, 90, 255))
    d.rectangle([250, 150, 380, 310], fill=(40, 90, 200, 255))
    d.polygon([(20, 320), (120, 200), (200, 330)], fill=(60, 180, 80, 255))
    im.save(f"{a.data}/images/sample.png")


if __name__ == "__main__":
    main()
