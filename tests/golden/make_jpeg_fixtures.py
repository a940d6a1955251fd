#!/usr/bin/env python3
"""Makes the JPEG texture fixtures (tests/golden/assets/jpeg/*.jpg) and their reference decodes (tests/golden/ref_stb_decode.json).

The files are written by Pillow (libjpeg) from a deterministic synthetic image; the reference decodes come from the REFERENCE'S OWN
stb_image build: oracle/_ref/ref_host `decode <file>` calls stbi_load(path, &w, &h, &n, 3) exactly as src/gpu_scene_builder.cpp:215 does.
Run in the build container only (needs /root/reference to have built oracle/_ref/ref_host); the outputs are committed, the reference is not."""
import json
import os
import subprocess

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "assets", "jpeg")
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_host")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.float64)
    img[..., 0] = 128 + 100 * np.sin(x / 5.0) * np.cos(y / 7.0)
    img[..., 1] = (x * 255.0 / max(1, w - 1) + y * 64.0 / max(1, h - 1)) % 256
    img[..., 2] = 255 * ((x // 4 + y // 4) % 2)
    img += rng.normal(0, 12, img.shape)
    img[h // 3:h // 3 + 3, :, :] = (250, 10, 10)                      # hard edges: chroma upsampling shows
    img[:, w // 2:w // 2 + 2, :] = (10, 10, 250)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8))


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = [
        ("base_444_q90", (37, 29), dict(quality=90, subsampling=0)),
        ("base_420_q75", (37, 29), dict(quality=75, subsampling=2)),
        ("base_422_q80", (40, 24), dict(quality=80, subsampling=1)),
        ("base_420_optimized", (64, 48), dict(quality=60, subsampling=2, optimize=True)),
        ("prog_420_q85", (37, 29), dict(quality=85, subsampling=2, progressive=True)),
        ("prog_444_q50", (48, 33), dict(quality=50, subsampling=0, progressive=True, optimize=True)),
        ("prog_422_q95", (31, 17), dict(quality=95, subsampling=1, progressive=True)),
        ("restart_420", (64, 40), dict(quality=70, subsampling=2, restart_marker_blocks=2)),
        ("restart_prog_444", (40, 40), dict(quality=70, subsampling=0, progressive=True, restart_marker_rows=1)),
        ("gray_q80", (33, 21), dict(quality=80)),
        ("gray_prog", (19, 35), dict(quality=65, progressive=True)),
        ("tiny_1x1", (1, 1), dict(quality=90, subsampling=2)),
        ("row_17x1", (17, 1), dict(quality=90, subsampling=2)),
        ("col_1x19", (1, 19), dict(quality=90, subsampling=2)),
        ("q100_444", (16, 16), dict(quality=100, subsampling=0)),
        ("q10_420", (56, 56), dict(quality=10, subsampling=2)),
    ]
    ref = {}
    for i, (name, (w, h), kw) in enumerate(cases):
        im = picture(w, h, 100 + i)
        if name.startswith("gray"):
            im = im.convert("L")
        path = os.path.join(OUT, name + ".jpg")
        im.save(path, "JPEG", **kw)
        r = json.loads(subprocess.run([REF, "decode", path], check=True, capture_output=True, text=True).stdout)
        ref[name] = r
    json.dump(ref, open(os.path.join(HERE, "ref_stb_decode.json"), "w"))
    print(len(ref), "fixtures,", sum(os.path.getsize(os.path.join(OUT, n + ".jpg")) for n in ref), "bytes of JPEG")


if __name__ == "__main__":
    main()
