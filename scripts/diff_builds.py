#!/usr/bin/env python3
"""Where do two builds of libmmfe.so disagree on the VGGSound log-mel?  usage: diff_builds.py a.so b.so [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ab_fbank import build_plan  # noqa: E402

a, b = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1234)
x = (0.1 * torch.randn(B, 160000, device="cuda", generator=g)).clamp_(-1, 1)
outs = []
for path in (a, b):
    plan = build_plan(os.path.abspath(path), 0)
    o = torch.full((B, 384, 128), float("nan"), device="cuda")
    plan.run_ptr(x.data_ptr(), x.stride(0), B, 160000, o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    outs.append(o)
d = (outs[0] - outs[1]).abs()
nz = (d > 0).nonzero()
print("cells differing:", nz.shape[0], "of", d.numel(), "max", float(d.max()))
if nz.shape[0]:
    print("rows", nz[:, 0].unique()[:20].tolist(), "frames", nz[:, 1].unique()[:40].tolist(), "mels", nz[:, 2].unique()[:40].tolist())
    top = d.flatten().topk(min(10, nz.shape[0]))
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        r, f, m = i // (384 * 128), (i // 128) % 384, i % 128
        print(f"  row {r} frame {f} mel {m}: {float(outs[0][r, f, m]):.6f} vs {float(outs[1][r, f, m]):.6f}  d={v:.3e}")
