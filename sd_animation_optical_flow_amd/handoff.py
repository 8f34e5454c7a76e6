"""Device-resident hand-off from the flow path to the SD inpainting model ("next" row f3).

The reference takes the warped AI frame and the inpaint mask back to the host, round-trips them through PIL
(`run_inpainting`, ofgen_keyframe_inpaint.py:255-290 -> `img2img_inpaint`, guided_ldm_inpainting.py:290-316) and
uploads the results again for the VAE encoder.  `prepare_inpaint_inputs` computes the same tensors on the GPU,
bit-exact to Pillow for the integer steps (GaussianBlur, Image.composite, the default-resample resize):

    image_mask          uint8 [B,H,W]    mask.convert('L').filter(GaussianBlur(mask_blur))          (:294-295)
    image               f32 [B,3,H,W]    Image.composite(reference, image, image_mask)/127.5 - 1    (:300-303)
    latmask (`nmask`)   f32 [B,4,h,w]    around(resize(image_mask, latent size)/255), tiled x4      (:306-311)
    conditioning_mask   f32 [B,1,H,W]    round(image_mask/255)                                      (:140-143)
    conditioning_image  f32 [B,3,H,W]    image * (1 - conditioning_mask), the VAE encoder's input   (:145-149)
    conditioning_mask_latent f32 [B,1,h,w]  nearest-neighbour resize of conditioning_mask           (:151)

The VAE / UNet / sampler that consume these are out of scope.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops


def _dev_u8(a, device) -> torch.Tensor:
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != torch.uint8:
        raise RuntimeError("expected uint8 data")
    return t.to(device).contiguous()


def prepare_inpaint_inputs(image_bgr, reference_bgr, mask, mask_blur: float = 4.0, latent_size: Optional[Tuple[int, int]] = None,
                           device="cuda") -> Dict[str, torch.Tensor]:
    """image_bgr: the frame to repaint (e.g. the warped AI frame), reference_bgr: the frame pasted through the mask,
    both uint8 [H,W,3] or [B,H,W,3] in cv2 channel order; mask: uint8 [H,W] or [B,H,W] (255 = repaint).
    numpy arrays or tensors on any device; everything returned lives on `device`.
    latent_size = (h, w) of the VAE latent (default H//8, W//8)."""
    img, ref, m = _dev_u8(image_bgr, device), _dev_u8(reference_bgr, device), _dev_u8(mask, device)
    if img.dim() == 3:
        img, ref, m = img[None], ref[None], m[None]
    if img.dim() != 4 or img.shape[3] != 3 or tuple(ref.shape) != tuple(img.shape) or tuple(m.shape) != tuple(img.shape[:3]):
        raise RuntimeError(f"shape mismatch: image {tuple(img.shape)}, reference {tuple(ref.shape)}, mask {tuple(m.shape)}")
    B, H, W, _ = img.shape
    h, w = latent_size if latent_size is not None else (H // 8, W // 8)
    image_mask = ops.gaussian_blur_u8(m.contiguous(), mask_blur)
    mask_latent = ops.resize_bicubic_u8(image_mask, h, w)
    image, cond_image, cond_mask, latmask, cml = ops.sd_handoff(img.contiguous(), ref.contiguous(), image_mask, mask_latent)
    return {"image_mask": image_mask, "image": image, "latmask": latmask, "conditioning_mask": cond_mask[:, None],
            "conditioning_image": cond_image, "conditioning_mask_latent": cml[:, None]}
