#!/usr/bin/env python3
"""Cut the first records of one of the reference's TFRecord DATA files into a small test fixture.

The reference ships TensorFlow-written datasets (datasets/shapes/three_mixed_shapes/*/input/tfrecords/*/*.tfrec).
The bytes of a record -- framing, checksums, Example proto, JPEG -- are data written by TensorFlow itself, so the
first N records verbatim pin the reader in yolo-v3-tf2_amd/core/load_tfrecords.py (framing, CRC-32C masking, proto
wire format) against a genuine TF-written stream.  Output: tests/golden/shapes_red_test_head.tfrec.

Usage (in the build container, where /root/reference exists): python tools/gen_tfrecord_fixture.py [--records 3]
"""
import argparse
import struct

SRC = "/root/reference/datasets/shapes/three_mixed_shapes/red/input/tfrecords/test/file_00_100.tfrec"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=3)
    ap.add_argument("--out", default="tests/golden/shapes_red_test_head.tfrec")
    a = ap.parse_args()
    with open(SRC, "rb") as f, open(a.out, "wb") as o:
        for _ in range(a.records):
            head = f.read(12)
            (n,) = struct.unpack("<Q", head[:8])
            o.write(head + f.read(n + 4))


if __name__ == "__main__":
    main()
