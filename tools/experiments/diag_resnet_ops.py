#!/usr/bin/env python
"""Sub-op bisect of one ResnetBlock3D of the width / 5 model (down_blocks[level].resnets[0]): every kernel launch of the
block on the GPU against the exact stand-in (tests/_emu_kernels.py) ON IDENTICAL INPUTS (the stand-in's own intermediates),
so each line isolates one kernel; the expected figure is one output rounding (fp16 ~2.8e-4, bf16 ~2.3e-3).

    python tools/diag_resnet_ops.py [fp16|bf16] [level]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import _emu_kernels as E  # noqa: E402
from imagine360_amd import configs, kernels as K  # noqa: E402

torch.set_grad_enabled(False)
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-30)).item()
mv = configs.build_mv_model(5, device="cpu", dtype=dt, xformers=True)
blk = mv.pano_unet.down_blocks[level].resnets[0]
gblk = configs.build_mv_model(5, device="cuda", dtype=dt, xformers=True).pano_unet.down_blocks[level].resnets[0]
cin, cout = blk.in_channels, blk.out_channels
g = torch.Generator().manual_seed(5)
for name, N, H, W, frames in (("pers", 320, 16 >> level, 16 >> level, 8), ("pano", 16, 32 >> level, 64 >> level, 8)):
    x = (torch.randn(N, H, W, cin, generator=g) * 1.3 + 0.2).to(dt)
    temb = torch.randn(N // frames, 256, generator=g).to(dt)
    for pano in ((False, True) if name == "pano" else (False,)):
        pad = 2 if pano else 0
        with E.patched_kernels():
            s1 = E.group_norm_stats(x, blk.norm1.weight, blk.norm1.bias, 32, blk.norm1.eps, pad)
            h1 = E.group_norm_apply(x, *s1, True, pad)
            t = blk.time_emb_proj(F.silu(temb)).contiguous()
            h2 = blk.conv1.forward_cl(h1, temb=t, imgs_per_temb=frames)
            s2 = E.group_norm_stats(h2, blk.norm2.weight, blk.norm2.bias, 32, blk.norm2.eps)
            h3 = E.group_norm_apply(h2, *s2, True)
            sh = x if blk.conv_shortcut is None else blk.conv_shortcut.forward_cl(x)
            out = blk.conv2.forward_cl(h3, x_off=pad, wout=W, res=sh)
            whole = blk.forward_cl(x, temb, frames, pano)
        d = lambda v: v.cuda()
        gs1 = K.group_norm_stats(d(x), gblk.norm1.weight, gblk.norm1.bias, 32, gblk.norm1.eps, pad)
        gh1 = K.group_norm_apply(d(x), d(s1[0]), d(s1[1]), True, pad)
        gt = gblk.time_emb_proj(F.silu(d(temb))).contiguous()
        gh2 = gblk.conv1.forward_cl(d(h1), temb=d(t), imgs_per_temb=frames)
        gs2 = K.group_norm_stats(d(h2), gblk.norm2.weight, gblk.norm2.bias, 32, gblk.norm2.eps)
        gh3 = K.group_norm_apply(d(h2), d(s2[0]), d(s2[1]), True)
        gsh = d(x) if gblk.conv_shortcut is None else gblk.conv_shortcut.forward_cl(d(x))
        gout = gblk.conv2.forward_cl(d(h3), x_off=pad, wout=W, res=d(sh))
        gwhole = gblk.forward_cl(d(x), d(temb), frames, pano)
        print(f"{dt} level {level} {name}{' pano-pad' if pano else ''} {N}x{H}x{W} {cin}->{cout}: "
              f"gn1 scale {rel(gs1[0], s1[0]):.1e} shift {(gs1[1].cpu() - s1[1]).abs().max().item():.1e} | gn1 apply {rel(gh1, h1):.1e} | temb lin {rel(gt, t):.1e} | "
              f"conv1 {rel(gh2, h2):.1e} | gn2 scale {rel(gs2[0], s2[0]):.1e} | gn2 apply {rel(gh3, h3):.1e} | shortcut {rel(gsh, sh):.1e} | "
              f"conv2+res {rel(gout, out):.1e} | whole block {rel(gwhole, whole):.1e}")
