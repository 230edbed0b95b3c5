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


class _PatchEmbedFn(torch.autograd.Function):
    """Forward on the matrix-pipe kernel; backward with torch's own matrix products (plain library GEMMs: the layer's gradients
    are outside the front-end path, they only have to exist so that a model that trains this layer can keep it)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        ctx.layer = layer
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return layer._kernel(x)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        layer = ctx.layer
        H, W = layer.img_size
        Ph, Pw = H // PATCH, W // PATCH
        B = x.shape[0]
        cd = layer.compute_dtype
        g = grad.reshape(B * Ph * Pw, -1).float()
        gw = gb = gx = None
        if ctx.needs_input_grad[1]:
            # the patches the kernel multiplied: the image rounded to the products' type, (B, Ph, 16, Pw, 16) -> (B Ph Pw, 256)
            img = x.reshape(B, H, W)[:, : Ph * PATCH, : Pw * PATCH].to(cd).float()
            patches = img.reshape(B, Ph, PATCH, Pw, PATCH).permute(0, 1, 3, 2, 4).reshape(B * Ph * Pw, PATCH * PATCH)
            gw = (g.t() @ patches).reshape(weight.shape).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        if ctx.needs_input_grad[0]:
            wr = weight.reshape(weight.shape[0], PATCH * PATCH).to(cd).float()
            gp = (g @ wr).reshape(B, Ph, Pw, PATCH, PATCH).permute(0, 1, 3, 2, 4).reshape(B, Ph * PATCH, Pw * PATCH)
            gx = torch.zeros(B, H, W, dtype=torch.float32, device=x.device)
            gx[:, : Ph * PATCH, : Pw * PATCH] = gp
            gx = gx.reshape(x.shape).to(x.dtype)
        return gx, gw, gb, None


class PatchEmbedLayer(torch.nn.Module):
    """The reference's PatchEmbed module (classification/vggsound/models/vit_model.py:179-199) after AudioTransformer has
    replaced its projection (vit_model.py:435-439): same attributes (`img_size`, `patch_size`, `num_patches`, `proj` = a
    Conv2d(1, embed_dim, 16, stride 16) whose `weight` / `bias` are the parameters), so a state_dict of the reference's module
    loads into it, and `forward(x)` = `self.proj(x).flatten(2).transpose(1, 2)` as @autocast() evaluates it.

    The forward pass runs on the matrix-pipe kernel.  After an optimizer step has changed `proj.weight` / `proj.bias` (their
    version counters move) the next forward re-packs them on the device (mmfe_patch_embed_update: one small kernel on the
    current stream, nothing goes through the host).  Gradients of weight, bias and image come from torch's own products."""

    def __init__(self, img_size=(384, 128), patch_size: int = 16, in_chans: int = 1, embed_dim: int = 768,
                 compute_dtype: torch.dtype = torch.float16, out_dtype: torch.dtype = torch.float32):
        super().__init__()
        if patch_size != PATCH and tuple(patch_size if not isinstance(patch_size, int) else (patch_size,) * 2) != (PATCH, PATCH):
            raise ValueError("the kernel is built for 16 x 16 patches")
        if in_chans != 1:
            raise ValueError("the audio projection has one input channel (vit_model.py:435)")
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.patch_size = (PATCH, PATCH)
        self.num_patches = (self.img_size[0] // PATCH) * (self.img_size[1] // PATCH)
        self.compute_dtype, self.out_dtype = compute_dtype, out_dtype
        self.proj = torch.nn.Conv2d(1, embed_dim, kernel_size=(PATCH, PATCH), stride=(PATCH, PATCH))
        self._pe = None
        self._seen = None

    def _kernel(self, x: torch.Tensor) -> torch.Tensor:
        w, b = self.proj.weight, self.proj.bias
        if not w.is_cuda:
            raise RuntimeError("PatchEmbedLayer runs on an MI355X: move the module to the GPU (there is no CPU fallback)")
        key = (w.data_ptr(), w._version, None if b is None else b.data_ptr(), None if b is None else b._version, w.device.index)
        if self._pe is None or self._pe.device != w.device.index:
            self._pe = PatchEmbed(self.img_size, w.detach(), None if b is None else b.detach(), compute_dtype=self.compute_dtype,
                                  out_dtype=self.out_dtype, device=w.device.index)
            self._seen = key
        elif key != self._seen:
            wc = w.detach().reshape(w.shape[0], PATCH * PATCH).float().contiguous()
            bc = None if b is None else b.detach().float().contiguous()
            stream = torch.cuda.current_stream(w.device).cuda_stream
            for h in self._pe._handles.values():
                h.update_ptr(wc.data_ptr(), 0 if bc is None else bc.data_ptr(), stream)
            self._host_stale = True
            self._seen = key
        in_dtype = x.dtype if (x.dtype == self.compute_dtype and self.img_size[1] % 8 == 0) else torch.float32
        if in_dtype not in self._pe._handles and getattr(self, "_host_stale", False):
            # a first call with this image dtype after the weights have moved: its handle is packed from the live parameters
            self._pe._w = w.detach().reshape(w.shape[0], PATCH * PATCH).float().cpu().contiguous().numpy()
            self._pe._b = None if b is None else b.detach().float().cpu().numpy()
            self._host_stale = False
        return self._pe(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or self.proj.weight.requires_grad):
            return _PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, self)
        return self._kernel(x)
