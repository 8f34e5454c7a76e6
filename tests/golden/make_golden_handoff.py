#!/usr/bin/env python3
"""Golden vectors of the SD-inpaint hand-off primitives, produced by the REAL Pillow in the build container.

    python tests/golden/make_golden_handoff.py        # writes tests/golden/sd_handoff_pil.npz

Runs the reference's own sequence (guided_ldm_inpainting.py:292-307) with PIL on seeded inputs:
GaussianBlur(mask_blur) on the 'L' mask, Image.composite(reference, image, mask), and the default-resample
resize of the mask to the latent grid.  The oracle (oracle/handoff_oracle.py) must reproduce every array
bit for bit (checked here at generation time and again by tests/test_oracle_golden.py).
"""
import os
import sys

import numpy as np
import PIL
from PIL import Image, ImageFilter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import handoff_oracle as HO  # noqa: E402


def case(seed, H, W, blur):
    rng = np.random.default_rng(seed)
    mask = np.zeros((H, W), np.uint8)
    for _ in range(6):                                   # blobs + speckle, like a dilated low-confidence mask
        y, x = rng.integers(0, H), rng.integers(0, W)
        mask[max(0, y - 9):y + 9, max(0, x - 14):x + 14] = 255
    mask[rng.random((H, W)) < 0.02] = 255
    mask[0, :] = 255                                     # borders exercise the replicated-edge rule
    mask[:, W - 1] = 255
    image = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)       # RGB as PIL sees them
    reference = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    m = Image.fromarray(mask).convert('L').filter(ImageFilter.GaussianBlur(blur))
    comp = Image.composite(Image.fromarray(reference), Image.fromarray(image), m)
    lat = m.convert('RGB').resize((W // 8, H // 8))
    return {"mask": mask, "image_rgb": image, "reference_rgb": reference, "blur": np.float32(blur),
            "pil_blur": np.array(m), "pil_composite": np.array(comp), "pil_latent": np.array(lat)[..., 0]}


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    for name, (seed, H, W, blur) in {"a": (1, 96, 128, 4), "b": (2, 72, 104, 4), "c": (3, 50, 70, 2), "d": (4, 64, 64, 7.5)}.items():
        c = case(seed, H, W, blur)
        assert np.array_equal(HO.gaussian_blur_u8(c["mask"], float(c["blur"])), c["pil_blur"]), name
        assert np.array_equal(HO.composite(c["reference_rgb"], c["image_rgb"], c["pil_blur"]), c["pil_composite"]), name
        assert np.array_equal(HO.resize_bicubic_u8(c["pil_blur"], H // 8, W // 8), c["pil_latent"]), name
        for k, v in c.items():
            out[f"{name}_{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", "sd_handoff_pil.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; oracle == Pillow", PIL.__version__, "on all cases")


if __name__ == "__main__":
    main()
