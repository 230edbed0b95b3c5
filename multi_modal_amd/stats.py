"""Normalisation statistics of a feature tensor on the device.

Stands for cal_norm (classification/vggsound/dataset/vggsound.py:85-107), which makes two DataLoader
passes over the data set to obtain the mean and the standard deviation used at vggsound.py:59.  Here the
sum and the sum of squares are reduced on the GPU in float64; with several ranks two scalars are
all-reduced instead of gathering features.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

from . import _native as N
from .plans import require_gpu


class RunningNorm:
    """Accumulates sum / sum of squares / count over batches of features."""

    def __init__(self, device: torch.device = None):
        require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.sums = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.count = 0

    def update(self, features: torch.Tensor) -> None:
        x = features
        if x.device != self.device:          # the sums live on self.device: a kernel on another GPU must not add into them
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        if x.numel() == 0:
            return
        if x.data_ptr() % 16 != 0:           # a contiguous view at an odd storage offset: the kernel reads 16-byte words
            x = x.clone()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        N.feature_sums_ptr(x.data_ptr(), x.numel(), self.sums.data_ptr(), x.device.index, stream)
        self.count += x.numel()

    def finalize(self, group=None) -> Tuple[float, float]:
        """(mean, std) over everything seen; std is the population standard deviation."""
        sums = self.sums.clone()
        count = torch.tensor([float(self.count)], dtype=torch.float64, device=self.device)
        if dist.is_initialized():
            packed = torch.cat([sums, count])
            dist.all_reduce(packed, group=group)
            sums, count = packed[:2], packed[2:]
        s, q = sums.tolist()
        n = float(count.item())
        if n == 0:
            return float("nan"), float("nan")
        mean = s / n
        var = max(q / n - mean * mean, 0.0)
        return mean, var ** 0.5


class BatchAveragedNorm:
    """The statistics pass at the bottom of classification/recognition/dataset/epic_kitchen.py (lines 275-290): per DataLoader
    batch the mean and the UNBIASED variance (`tensor.std() ** 2`) of the spectrogram batch, both averaged over the batches;
    the reported std is the square root of the averaged variance.  That differs from RunningNorm, which pools all values
    (population variance about the global mean): batch-to-batch drift of the mean is not part of this figure.
    Sums are reduced on the GPU in float64 (one kernel per batch, no host synchronisation until finalize)."""

    def __init__(self, device: torch.device = None):
        require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.acc = torch.zeros(3, dtype=torch.float64, device=self.device)      # sum of batch means, sum of batch variances, batches

    def update(self, batch: torch.Tensor) -> None:
        n = batch.numel()
        if n < 2:
            raise ValueError("a batch needs at least two values for an unbiased variance")
        r = RunningNorm(self.device)
        r.update(batch)
        s, q = r.sums[0], r.sums[1]
        self.acc += torch.stack([s / n, (q - s * s / n) / (n - 1), torch.ones((), dtype=torch.float64, device=self.device)])

    def finalize(self, group=None) -> Tuple[float, float]:
        """(mean of the batch means, sqrt of the mean of the batch variances); with several ranks every rank's batches count."""
        acc = self.acc.clone()
        if dist.is_initialized():
            dist.all_reduce(acc, group=group)
        m, v, k = acc.tolist()
        if k == 0:
            return float("nan"), float("nan")
        return m / k, max(v / k, 0.0) ** 0.5


def cal_norm(features: torch.Tensor) -> Tuple[float, float]:
    """Mean and standard deviation of one feature tensor."""
    r = RunningNorm(features.device if features.is_cuda else None)
    r.update(features)
    return r.finalize()
