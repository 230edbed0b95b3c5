"""Patch embedding on the device: drop-in for PatchEmbed.forward (classification/vggsound/models/vit_model.py:193-199)
with the single-channel 16 x 16 / stride-16 projection AudioTransformer installs (vit_model.py:435):

    x = self.proj(x).flatten(2).transpose(1, 2)            (B, 1, H, W) -> (B, (H // 16) * (W // 16), embed_dim)

The reference runs it under @autocast() (vit_model.py:388), i.e. as a float16 convolution with float32 accumulation; here it
is one matrix product on the matrix pipe of an MI355X (csrc/patch_embed.hip), fed by the spectrogram the front-end kernels
have just written, so that the first tensor a training step materialises after the waveform is the token tensor.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N
from .plans import OUT_CODES, require_gpu

PATCH = 16
_COMPUTE = {torch.float16: N.OUT_F16, torch.bfloat16: N.OUT_BF16}


class PatchEmbed:
    """PatchEmbed(img_size=(H, W), patch_size=16, in_chans=1, embed_dim=E) with its Conv2d's weight and bias.

    weight: (E, 1, 16, 16) (Conv2d.weight) or (E, 256); bias: (E,) or None.
    compute_dtype: the 16-bit type the products are formed in (float16 = what torch's autocast uses on a GPU; bfloat16);
    the image and the weights are rounded to it (round to nearest even), sums are float32.
    out_dtype: dtype of the tokens (float32, or the 16-bit type the autocast convolution returns).
    Embedding sizes 384 and 768 (DeiT-small / -base, the configurations of vit_model.py) are built."""

    def __init__(self, img_size, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, compute_dtype: torch.dtype = torch.float16,
                 out_dtype: torch.dtype = torch.float32, device: Optional[int] = None):
        require_gpu()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.patch_size = (PATCH, PATCH)
        if compute_dtype not in _COMPUTE:
            raise ValueError("compute_dtype must be torch.float16 or torch.bfloat16")
        if out_dtype not in OUT_CODES:
            raise ValueError("out_dtype must be torch.float32, torch.float16 or torch.bfloat16")
        w = weight.detach().to("cpu", torch.float32)
        if w.dim() == 4:
            if w.shape[1] != 1 or tuple(w.shape[2:]) != (PATCH, PATCH):
                raise ValueError(f"expected a (E, 1, 16, 16) convolution weight, got {tuple(w.shape)}")
            w = w.reshape(w.shape[0], PATCH * PATCH)
        b = None if bias is None else bias.detach().to("cpu", torch.float32).numpy()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.out_dtype = out_dtype
        self.compute_dtype = compute_dtype
        self._w, self._b = w.contiguous().numpy(), b
        self._handles = {}                       # one per image dtype: float32 (rounded on the way in) or the products' own 16-bit type
        h = self._handle(torch.float32)
        self.num_patches = h.n_patches
        self.embed_dim = h.embed_dim

    def _handle(self, in_dtype: torch.dtype) -> N.PatchEmbedHandle:
        h = self._handles.get(in_dtype)
        if h is None:
            code = {torch.float32: N.IN_F32, torch.float16: N.IN_F16, torch.bfloat16: N.IN_BF16}[in_dtype]
            h = N.PatchEmbedHandle(self.img_size[0], self.img_size[1], self._w, self._b, _COMPUTE[self.compute_dtype],
                                   OUT_CODES[self.out_dtype], device=self.device, in_dtype=code)
            self._handles[in_dtype] = h
        return h

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: (B, 1, H, W) or (B, H, W) on this object's GPU (a CPU tensor is copied over and the tokens copied back), float32 - rounded
        to compute_dtype on the way into the matrix pipe - or already of compute_dtype (the spectrogram as
        `preprocessing_audio_batch(dtype=...)` stores it: the same values, half the bytes); any other dtype is cast to float32 first.
        out, when given: a (B, T, E) tensor of out_dtype with T >= num_patches whose LAST num_patches token rows are written
        (T = num_patches + 1 leaves row 0 for the class token that vit_model.py:382-383 concatenates in front)."""
        origin = x.device
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise ValueError("expected a single-channel image (B, 1, H, W)")
            x = x[:, 0]
        if x.dim() != 3 or tuple(x.shape[1:]) != self.img_size:
            raise ValueError(f"expected (B, {self.img_size[0]}, {self.img_size[1]}) images, got {tuple(x.shape)}")
        if x.dtype != torch.float32 and not (x.dtype == self.compute_dtype and self.img_size[1] % 8 == 0):
            x = x.to(torch.float32)
        if not x.is_cuda:
            x = x.cuda(self.device)
        if x.stride(2) != 1 or x.stride(1) != x.shape[2] or x.stride(0) % 8 != 0 or x.data_ptr() % 16 != 0:
            x = x.contiguous()
        if x.data_ptr() % 16 != 0:
            x = x.clone()
        B = x.shape[0]
        P, E = self.num_patches, self.embed_dim
        if out is None:
            out = torch.empty((B, P, E), dtype=self.out_dtype, device=x.device)
            first = 0
        else:
            if out.dim() != 3 or out.shape[0] != B or out.shape[2] != E or out.shape[1] < P or out.dtype != self.out_dtype \
                    or out.device != x.device or out.stride(2) != 1 or out.stride(1) != E:
                raise ValueError(f"out must be a ({B}, >= {P}, {E}) {self.out_dtype} tensor with contiguous token rows on {x.device}")
            first = out.shape[1] - P
        if B > 0:
            stream = torch.cuda.current_stream(x.device).cuda_stream
            self._handle(x.dtype).run_ptr(x.data_ptr(), x.stride(0) if B > 1 else x[0].numel(), B, out.data_ptr() + first * E * out.element_size(),
                            out.stride(0) if B > 1 else out[0].numel(), stream)
        return out if origin.type == "cuda" else out.to(origin)

    forward = __call__

    def close(self) -> None:
        for h in self._handles.values():
            h.close()
        self._handles.clear()
