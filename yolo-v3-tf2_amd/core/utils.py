"""Host mirror of the helpers of reference core/utils.py that sit on the inference path."""
import os

import numpy as np


def get_anchors(anchors_file):
    """reference: core/utils.py:31-37 -- `loadtxt(delimiter=',')` then reshape(-1, 3, 2): row-major, so table[0] holds
    the first three lines of the file (the largest anchors in the COCO file, used by the coarsest grid)."""
    nanchors_per_scale = 3
    anchor_entry_size = 2
    anchors_table = np.loadtxt(anchors_file, dtype=float, delimiter=",")
    return anchors_table.reshape(-1, nanchors_per_scale, anchor_entry_size)


def dir_filelist(images_dir, ext_list=".*"):
    """reference: core/utils.py:46-53"""
    filenames = []
    for f in os.listdir(images_dir):
        ext = os.path.splitext(f)[1]
        if ext.lower() not in ext_list:
            continue
        filenames.append(f"{images_dir}/{f}")
    return filenames


def resize_bilinear(img, out_h, out_w):
    """tf.image.resize(img, (out_h, out_w)) default method (bilinear, antialias=False, half-pixel centres) for one
    HWC float32 image -- reference call site: inference.py:158.  fp32; host counterpart of y3_preprocess_image."""
    img = np.asarray(img, np.float32)
    in_h, in_w = img.shape[0], img.shape[1]

    def taps(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        src = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(src)
        lo = np.maximum(fl, 0).astype(np.int64)
        hi = np.minimum(np.ceil(src), n_in - 1).astype(np.int64)
        return lo, hi, (src - fl).astype(np.float32)

    ylo, yhi, yfr = taps(in_h, out_h)
    xlo, xhi, xfr = taps(in_w, out_w)
    # TF's ResizeBilinear interpolates along x first (top and bottom rows), then along y
    top = img[ylo][:, xlo] + (img[ylo][:, xhi] - img[ylo][:, xlo]) * xfr[None, :, None]
    bot = img[yhi][:, xlo] + (img[yhi][:, xhi] - img[yhi][:, xlo]) * xfr[None, :, None]
    return (top + (bot - top) * yfr[:, None, None]).astype(np.float32)


def resize_image(img, target_height, target_width):
    """Aspect-preserving resize then centred zero padding to (target_height, target_width) -- reference
    core/utils.py:17-28 (tf.image.resize(preserve_aspect_ratio=True) + tf.image.pad_to_bounding_box).  `img` is one
    HWC image or an NHWC batch, float32.  Scaled size = round-half-even(min(th/h, tw/w) * (h, w)), at least 1."""
    img = np.asarray(img, np.float32)
    if img.ndim == 4:
        return np.stack([resize_image(i, target_height, target_width) for i in img])
    h, w = img.shape[0], img.shape[1]
    scale = min(np.float32(target_height) / np.float32(h), np.float32(target_width) / np.float32(w))
    sh = max(1, int(np.rint(scale * np.float32(h))))
    sw = max(1, int(np.rint(scale * np.float32(w))))
    scaled = img if (sh, sw) == (h, w) else resize_bilinear(img, sh, sw)   # same-size bilinear is the identity
    top, left = (target_height - sh) // 2, (target_width - sw) // 2
    if top < 0 or left < 0 or top + sh > target_height or left + sw > target_width:
        raise ValueError("resize_image: scaled image does not fit the target")   # pad_to_bounding_box raises too
    out = np.zeros((target_height, target_width, img.shape[2]), np.float32)
    out[top:top + sh, left:left + sw] = scaled
    return out


def load_image_rgb01(path):
    """tf.image.decode_image(bytes, channels=3, dtype=float32): RGB, alpha dropped, uint8/255 -> [0,1]
    (reference: inference.py:157)."""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGBA") if im.mode in ("RGBA", "LA", "P") else im.convert("RGB")
        a = np.asarray(im, np.uint8)[..., :3]
    return a.astype(np.float32) * np.float32(1.0 / 255.0)   # convert_image_dtype: cast * (1 / 255)


def load_image_u8(path):
    """RGB uint8 [H,W,3] (alpha dropped) for the GPU input stage (runtime.preprocess_image)."""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGBA") if im.mode in ("RGBA", "LA", "P") else im.convert("RGB")
        return np.ascontiguousarray(np.asarray(im, np.uint8)[..., :3])
