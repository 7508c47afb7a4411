#!/usr/bin/env python
"""Sub-op bisect of one WarpAttn block of the width / 5 model: every kernel launch on the GPU against the exact stand-in
(tests/_emu_kernels.py) ON IDENTICAL INPUTS (the stand-in's own intermediates).  Expected: one output rounding per line.

    python tools/diag_warp_ops.py [fp16|bf16]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import _emu_kernels as E  # noqa: E402
from imagine360_amd import configs, kernels as K, synthetic as S  # noqa: E402
from imagine360_amd.layers import layer_norm  # noqa: E402

torch.set_grad_enabled(False)
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-30)).item()
cm = configs.build_mv_model(5, device="cpu", dtype=dt, xformers=True)
gm = configs.build_mv_model(5, device="cuda", dtype=dt, xformers=True)
cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 128).items()}
gcams = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cams.items()}
g = torch.Generator().manual_seed(9)
frames, b, m = 8, 2, 20
for name, cblk, gblk, ph, eh in (("enc0", cm.cp_blocks_encoder[0], gm.cp_blocks_encoder[0], 8, 16), ("enc1", cm.cp_blocks_encoder[1], gm.cp_blocks_encoder[1], 4, 8),
                                 ("enc2", cm.cp_blocks_encoder[2], gm.cp_blocks_encoder[2], 2, 4), ("mid", cm.cp_blocks_mid, gm.cp_blocks_mid, 2, 4)):
    c = cblk.dim
    pers = (torch.randn(b * m * frames, ph, ph, c, generator=g) * 1.2).to(dt)
    equi = (torch.randn(b * frames, eh, 2 * eh, c, generator=g) * 1.2).to(dt)
    t, gt = cblk.transformer, gblk.transformer
    h = t.attn1.heads
    with E.patched_kernels():
        b_e2p, b_p2e, pers_pe, equi_pe, packed = cblk.geometry(ph, ph, eh, 2 * eh, cams, False, pers.device, dt)
        eq = equi.reshape(b * frames, eh * 2 * eh, c)
        pr = pers.reshape(b, m, frames, ph * ph, c).permute(0, 2, 1, 3, 4).reshape(b * frames, m * ph * ph, c).contiguous()
        eq_n, pr_n = layer_norm(t.norm1, eq, pre=equi_pe), layer_norm(t.norm1, pr, pre=pers_pe)
        qkv_e, qkv_p = t.attn1.qkv(eq_n), t.attn1.qkv(pr_n)
        a_e = E.attention(qkv_e[..., :c], qkv_p[..., c:2 * c], qkv_p[..., 2 * c:], h, bias=b_e2p, bias_packed=packed)
        a_p = E.attention(qkv_p[..., :c], qkv_e[..., c:2 * c], qkv_e[..., 2 * c:], h, bias=b_p2e, bias_packed=packed)
        eq1 = t.attn1.out_proj(a_e, residual=eq)
        n2 = layer_norm(t.norm2, eq1)
        ff = t.ff(n2, residual=eq1)
        wp, we = cblk.forward_cl(pers, equi, cams, frames, opposite=False)
    d = lambda v: v.cuda()
    gb_e2p, gb_p2e, gpers_pe, gequi_pe, gpacked = gblk.geometry(ph, ph, eh, 2 * eh, gcams, False, torch.device("cuda"), dt)
    geq_n, gpr_n = layer_norm(gt.norm1, d(eq), pre=gequi_pe), layer_norm(gt.norm1, d(pr), pre=gpers_pe)
    gqkv_e = gt.attn1.qkv(d(eq_n))
    dq_e, dq_p = d(qkv_e), d(qkv_p)
    ga_e = K.attention(dq_e[..., :c], dq_p[..., c:2 * c], dq_p[..., 2 * c:], h, bias=gb_e2p, bias_packed=gpacked)
    ga_p = K.attention(dq_p[..., :c], dq_e[..., c:2 * c], dq_e[..., 2 * c:], h, bias=gb_p2e, bias_packed=gpacked)
    geq1 = gt.attn1.out_proj(d(a_e), residual=d(eq))
    gn2 = layer_norm(gt.norm2, d(eq1))
    gff = gt.ff(d(n2), residual=d(eq1))
    gwp, gwe = gblk.forward_cl(d(pers), d(equi), gcams, frames, opposite=False)
    print(f"{dt} {name} C={c} pers {ph}x{ph} equi {eh}x{2 * eh} heads {h} packed={packed}/{gpacked}: pe {rel(gequi_pe, equi_pe):.1e} {rel(gpers_pe, pers_pe):.1e} | "
          f"bias {rel(gb_e2p.float(), b_e2p.float()):.1e} | LN(x+pe) {rel(geq_n, eq_n):.1e} {rel(gpr_n, pr_n):.1e} | qkv {rel(gqkv_e, qkv_e):.1e} | "
          f"attn e2p {rel(ga_e, a_e):.1e} p2e {rel(ga_p, a_p):.1e} | out-proj+res {rel(geq1, eq1):.1e} | LN2 {rel(gn2, n2):.1e} | FF+res {rel(gff, ff):.1e} | "
          f"whole: pers {rel(gwp, wp):.1e} equi {rel(gwe, we):.1e}")
