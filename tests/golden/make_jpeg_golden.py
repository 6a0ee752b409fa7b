"""Writes tests/golden/jpeg/*.jpg (encoded with Pillow) and jpeg_texels.npz = the texels the
REFERENCE's stb_image build (oracle/_ref, LoadSTB) decodes from them.  Run in the build
container (needs Pillow and /root/reference):  python tests/golden/make_jpeg_golden.py"""
import os, sys
import numpy as np
from PIL import Image as PI

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import _ref  # noqa: E402


def photo(w, h, rng):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 13.0), 127 + 120 * np.cos(x / 5.0) * np.sin(y / 9.0), (x * 3 + y * 5) % 256], -1)
    img += rng.normal(0, 12, img.shape)
    img[h // 3:h // 2, w // 4:w // 2] = rng.integers(0, 256, (h // 2 - h // 3, w // 2 - w // 4, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {  # name: (w, h, mode, save kwargs)
    "baseline_444": (40, 27, "RGB", dict(quality=85, subsampling=0)),
    "baseline_420_restart": (57, 35, "RGB", dict(quality=60, subsampling=2, restart_marker_blocks=3)),
    "progressive_422": (33, 50, "RGB", dict(quality=75, subsampling=1, progressive=True)),
    "progressive_420_optimized": (64, 64, "RGB", dict(quality=92, subsampling=2, progressive=True, optimize=True)),
    "grey_baseline": (31, 17, "L", dict(quality=70)),
    "grey_progressive": (16, 9, "L", dict(quality=40, progressive=True)),
    "one_pixel": (1, 1, "RGB", dict(quality=90, subsampling=2)),
}

if __name__ == "__main__":
    rng = np.random.default_rng(3)
    out = {}
    os.makedirs(os.path.join(HERE, "jpeg"), exist_ok=True)
    for name, (w, h, mode, kw) in CASES.items():
        img = photo(w, h, rng)
        path = os.path.join(HERE, "jpeg", name + ".jpg")
        PI.fromarray(img if mode == "RGB" else img[..., 0], mode).save(path, "JPEG", **kw)
        out[name] = _ref.load_stb(path)
    np.savez_compressed(os.path.join(HERE, "jpeg", "jpeg_texels.npz"), **out)
    print({k: v.shape for k, v in out.items()})
