"""TFRecord input source: the reference's `parse_tfrecords` without TensorFlow.

Mirrors reference core/load_tfrecords.py:17-108 (`parse_tfrecord_fn`, `pad_to_max_boxes`, `parse_tfrecords`): every
`*.tfrec` file of a directory is a stream of framed `tf.train.Example` protos holding a JPEG (`image/encoded`),
box corner lists (`image/object/bbox/{xmin,ymin,xmax,ymax}`, normalised) and class names (`image/object/class/text`).
The reference leans on three TensorFlow pieces that are restated here from their published formats:

* the TFRecord framing -- `u64 length | u32 masked_crc32c(length) | data | u32 masked_crc32c(data)`, little endian,
  mask(c) = rotr(c, 15) + 0xa282ead8 -- with both checksums verified the way `tf.data.TFRecordDataset` does
  (CRC-32C itself is the native `y3_crc32c`, include/y3.h);
* the protobuf wire format of `Example { Features features = 1 }`, `Features { map<string, Feature> feature = 1 }`,
  `Feature { oneof { BytesList = 1; FloatList = 2; Int64List = 3 } }`, lists `repeated value = 1` (packed or not);
* a tiny `Dataset` with the `map` / `batch` / iteration calls the callers make (reference inference.py:121-125,
  evaluate_yolov3.py:86-94).

JPEG decode is Pillow's (libjpeg, slow-integer DCT + fancy upsampling, the same defaults `tf.image.decode_jpeg` names);
pixel-exact agreement with TensorFlow's decoder is unpinned -- TensorFlow is not available to compare against.
"""
import glob
import io
import struct

import numpy as np

from .utils import resize_bilinear

_MASK_DELTA = 0xA282EAD8


def crc32c(data) -> int:
    """CRC-32C (Castagnoli), computed by the native library."""
    import ctypes as C
    from .. import _lib
    buf = bytes(data)
    return int(_lib.load().y3_crc32c(C.cast(C.c_char_p(buf), C.c_void_p), len(buf)))


def masked_crc32c(data) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def read_records(path, check_crc=True):
    """Yield the payload of every record of one TFRecord file; a bad checksum or truncated record raises
    ValueError (TensorFlow raises DataLossError there)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            (length,), (len_crc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and masked_crc32c(head[:8]) != len_crc:
                raise ValueError(f"{path}: corrupted record length")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError(f"{path}: truncated record")
            if check_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise ValueError(f"{path}: corrupted record data")
            yield data


def write_records(path, payloads):
    """Frame `payloads` (bytes) into one TFRecord file."""
    with open(path, "wb") as f:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


# ------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; value is an int (varint), or a memoryview slice."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        if pos > end:
            raise ValueError("truncated protobuf message")
        yield num, wt, val


def _parse_feature(buf):
    for num, wt, val in _fields(buf):
        if wt != 2:
            continue
        if num == 1:      # BytesList
            return [bytes(v) for n, w, v in _fields(val) if n == 1 and w == 2]
        if num == 2:      # FloatList: packed (one length-delimited blob) or one fixed32 per value
            out = []
            for n, w, v in _fields(val):
                if n == 1:
                    out.append(np.frombuffer(bytes(v), dtype="<f4"))
            return np.concatenate(out).astype(np.float32) if out else np.zeros((0,), np.float32)
        if num == 3:      # Int64List
            out = []
            for n, w, v in _fields(val):
                if n != 1:
                    continue
                if w == 0:
                    out.append(v)
                else:
                    p = 0
                    while p < len(v):
                        x, p = _varint(v, p)
                        out.append(x)
            return np.array([x - (1 << 64) if x >= (1 << 63) else x for x in out], dtype=np.int64)
    return []


def parse_example(record):
    """serialized tf.train.Example -> {feature name: list of bytes | float32 array | int64 array}"""
    buf = memoryview(record)
    out = {}
    for num, wt, features in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for n, w, entry in _fields(features):
            if n != 1 or w != 2:
                continue
            key, feat = None, None
            for kn, kw, kv in _fields(entry):
                if kn == 1:
                    key = bytes(kv).decode("utf-8")
                elif kn == 2:
                    feat = _parse_feature(kv)
            if key is not None:
                out[key] = feat if feat is not None else []
    return out


def _ld(num, payload):
    n, head = len(payload), bytearray()
    key = (num << 3) | 2
    for v in (key, n):
        while True:
            b = v & 0x7F
            v >>= 7
            head.append(b | (0x80 if v else 0))
            if not v:
                break
    return bytes(head) + payload


def make_example(features):
    """{name: bytes | list of bytes | float sequence} -> serialized tf.train.Example (the writer side of the schema
    at reference core/load_tfrecords.py:34-41; used by tools and tests)."""
    entries = b""
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if len(v) and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in v))
        else:
            feat = _ld(2, _ld(1, np.asarray(v, dtype="<f4").tobytes()))
        entries += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feat))
    return _ld(1, entries)


# ------------------------------------------------------------------------------------------ dataset
class Dataset:
    """The slice of tf.data.Dataset the reference's callers use: map, batch, iteration (re-iterable)."""

    def __init__(self, make_iter):
        self._make_iter = make_iter

    def __iter__(self):
        return self._make_iter()

    def map(self, fn):
        return Dataset(lambda: (fn(*e) if isinstance(e, tuple) else fn(e) for e in self))

    def batch(self, batch_size, drop_remainder=False):
        def it():
            pend = []
            for e in self:
                pend.append(e)
                if len(pend) == batch_size:
                    yield _stack(pend)
                    pend = []
            if pend and not drop_remainder:
                yield _stack(pend)
        return Dataset(it)


def _stack(elems):
    if isinstance(elems[0], tuple):
        return tuple(np.stack([e[i] for e in elems]) for i in range(len(elems[0])))
    return np.stack(elems)


def decode_jpeg_u8(encoded):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(encoded)).convert("RGB"), dtype=np.uint8)   # a writable copy


def example_boxes(example, class_table=None):
    """[n, 5 (+1)] rows (xmin, ymin, xmax, ymax, 1.0[, class]) -- reference core/load_tfrecords.py:50-70"""
    cols = [np.asarray(example.get(f"image/object/bbox/{k}", []), dtype=np.float32)
            for k in ("xmin", "ymin", "xmax", "ymax")]
    cols.append(np.ones_like(cols[3]))
    if class_table is not None:
        labels = example.get("image/object/class/text", [])
        cols.append(np.array([class_table.get(l.decode("utf-8"), -1) for l in labels], dtype=np.float32))
    return np.stack(cols, axis=1) if len(cols[0]) else np.zeros((0, len(cols)), np.float32)


def parse_tfrecord_fn(record, image_size, max_bboxes, class_table=None):
    """record -> (image [S,S,3] fp32 in [0,1], boxes).  Host restatement: bilinear resize of the 0..255 values, then
    a divide by 255 (reference :46-48).  The GPU input stage does the same arithmetic (runtime.preprocess_image)."""
    example = parse_example(record)
    img = decode_jpeg_u8(example["image/encoded"][0]).astype(np.float32)
    x = resize_bilinear(img, image_size, image_size) / np.float32(255)
    return x.astype(np.float32), example_boxes(example, class_table)


def pad_to_max_boxes(y, max_bboxes):
    if y.shape[0] > max_bboxes:   # tf.pad with a negative amount fails the same way
        raise ValueError(f"{y.shape[0]} boxes exceed max_bboxes={max_bboxes}")
    return np.concatenate([y, np.zeros((max_bboxes - y.shape[0], y.shape[1]), y.dtype)], axis=0)


def load_class_table(class_file):
    """name -> line number (reference :92-94 StaticHashTable over TextFileIndex.LINE_NUMBER, default -1)"""
    if not class_file:
        return None
    with open(class_file) as f:
        return {line.rstrip("\n"): i for i, line in enumerate(f)}


def list_tfrecord_files(tfrecords_dir):
    return sorted(glob.glob(f"{tfrecords_dir}/*.tfrec"))


def iter_examples(tfrecords_dir):
    """parsed Example dicts of every record of every *.tfrec file (sorted file order; tf's list_files shuffles)"""
    for path in list_tfrecord_files(tfrecords_dir):
        for rec in read_records(path):
            yield parse_example(rec)


def parse_tfrecords(tfrecords_dir, image_size, max_bboxes, class_file=None):
    class_table = load_class_table(class_file)

    def it():
        for path in list_tfrecord_files(tfrecords_dir):
            for rec in read_records(path):
                x, y = parse_tfrecord_fn(rec, image_size, max_bboxes, class_table)
                yield x, pad_to_max_boxes(y, max_bboxes)
    return Dataset(it)
