"""Parity of every HIP kernel (through the C ABI) against the CPU oracle / plain fp32 torch on the
same seeded inputs.  Tolerances are for 16-bit storage with fp32 accumulation: relative L2 error of
the whole tensor <= 1e-2 (bf16) / 3e-3 (fp16) unless noted, plus a loose max-abs bound."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from im360_oracle import geometry as OG, unet as OU  # noqa: E402
from imagine360_amd import kernels as K  # noqa: E402

DTYPES = [torch.bfloat16, torch.float16]
TOL = {torch.bfloat16: 1e-2, torch.float16: 3e-3}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def blockrel(a, b, rows=256):
    """max over blocks of `rows` consecutive rows (last dim = one row) of the block's relative L2 error: a wrong tile edge,
    a dropped row or one bad 32-row MFMA block cannot hide in a whole-tensor norm."""
    a, b = a.double().cpu().reshape(-1, a.shape[-1]), b.double().cpu().reshape(-1, b.shape[-1])
    n = (a.shape[0] + rows - 1) // rows * rows
    pad = lambda t: torch.cat([t, torch.zeros(n - t.shape[0], t.shape[1], dtype=t.dtype)]).reshape(n // rows, -1)
    num, den = pad(a - b).norm(dim=1), pad(b).norm(dim=1)
    return float((num / den.clamp_min(1e-30 + 1e-3 * float(den.max()))).max())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q16(t, dt):
    """Round to the 16-bit dtype and back so the reference sees exactly the kernel's inputs."""
    return t.to(dt).float()


def test_library_loads():
    assert K.lib().im360_abi_version() == K.ABI_VERSION == 5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 3, 256, 192, 64), (3, 2, 100, 77, 64), (2, 5, 16, 16, 64),
                                          (1, 4, 640, 1280, 32), (2, 2, 33, 130, 32), (1, 1, 2048, 2048, 64)])
def test_attention(dt, B, H, Nq, Nk, D):
    q, k, v = (q16(rnd(B, n, H * D, seed=s), dt) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    ref = OU.sdpa(q, k, v, H)
    out = K.attention(q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda(), H)
    assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt]       # no 32-query block may be off
    assert (out.float().cpu() - ref).abs().max() < 0.05


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_bias_and_strided_qkv(dt):
    """WarpAttn form: fused QKV rows (row stride 3C), shared [Nq, Nk] bias in [-1, 1], d = 32."""
    B, H, D, Nq, Nk = 3, 4, 32, 200, 320
    C = H * D
    qkv_q = q16(rnd(B, Nq, 3 * C, seed=4), dt)
    qkv_k = q16(rnd(B, Nk, 3 * C, seed=5), dt)
    bias = q16(torch.rand(Nq, Nk, generator=torch.Generator().manual_seed(6)) * 2 - 1, dt)
    ref = OU.sdpa(qkv_q[..., :C], qkv_k[..., C:2 * C], qkv_k[..., 2 * C:], H, bias=bias)
    dq, dk = qkv_q.to(dt).cuda(), qkv_k.to(dt).cuda()
    out = K.attention(dq[..., :C], dk[..., C:2 * C], dk[..., 2 * C:], H, bias=bias.to(dt).cuda())
    assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_two_kv_sets_accumulate(dt):
    """IP cross attention: attn(Q, K_text, V_text) + 1.0 * attn(Q, K_ip, V_ip), logit scale 1.0 or d^-1/2."""
    B, H, D, Nq = 2, 5, 64, 300
    C = H * D
    q = q16(rnd(B, Nq, C, seed=7, scale=0.3), dt)
    k1, v1, k2, v2 = (q16(rnd(B, n, C, seed=s), dt) for n, s in ((77, 8), (77, 9), (64, 10), (64, 11)))
    for scale in (1.0, D ** -0.5):
        ref = OU.sdpa(q, k1, v1, H, scale=scale) + OU.sdpa(q, k2, v2, H, scale=scale)
        dq = q.to(dt).cuda()
        out = K.attention(dq, k1.to(dt).cuda(), v1.to(dt).cuda(), H, scale=scale)
        K.attention(dq, k2.to(dt).cuda(), v2.to(dt).cuda(), H, scale=scale, out=out, accumulate=True)
        assert rel(out, ref) < 1.5 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,Nq,group", [(4, 300, 2), (2, 1024, 1), (6, 40, 3), (32, 64, 16), (4, 1056, 2), (64, 1024, 16)])
def test_attention_two_kv_sets_one_launch(dt, B, Nq, group):
    """im360_attn_fwd2: text (77 keys, ragged tile) + IP (64 keys) cross attention of one query in one launch, with the
    per-video key / value sharing (kv_group) and an IP scale, against the oracle's two attention calls.  Nq % 32 == 0 runs
    the resident-K/V kernel (both store forms, bit-identical), the other shapes the generic two-pass kernel; knob attn_x = 0
    forces the generic kernel everywhere."""
    H, D = 5, 64
    C = H * D
    q = q16(rnd(B, Nq, C, seed=70, scale=0.3), dt)
    k1, v1, k2, v2 = (q16(rnd(B // group, n, C, seed=s), dt) for n, s in ((77, 71), (77, 72), (64, 73), (64, 74)))
    rep = lambda t: t.repeat_interleave(group, 0)
    dq, dk1, dv1, dk2, dv2 = (t.to(dt).cuda() for t in (q, k1, v1, k2, v2))
    try:
        for scale, s2 in ((1.0, 1.0), (D ** -0.5, 0.7)):
            ref = OU.sdpa(q, rep(k1), rep(v1), H, scale=scale) + s2 * OU.sdpa(q, rep(k2), rep(v2), H, scale=scale)
            outs = []
            for x in (1, 2, 0, 3):          # 3: twelve-wave workgroups with LDS-DMA query rings (large problems; else = 1)
                K.tuning_set("attn_x", x)
                out = K.attention2(dq, dk1, dv1, dk2, dv2, H, scale=scale, out_scale2=s2, kv_group=group)
                assert rel(out, ref) < 1.5 * TOL[dt] and blockrel(out, ref, 32) < 3 * TOL[dt], (x, scale)
                outs.append(out)
            assert torch.equal(outs[0], outs[1])          # 16-byte and 8-byte stores of the same values
            assert torch.equal(outs[0], outs[3])          # the same arithmetic on ring-fetched queries
    finally:
        K.tuning_set("attn_x", 3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n1,n2", [(90, 48), (96, 64), (65, 33), (80, 64)])
def test_attention_resident_kv_sets_general_key_counts(dt, n1, n2):
    """The resident-K/V cross attention away from the model's 77 + 64 keys: both sets ragged (masked by the MFMA C operand),
    exactly full blocks, the smallest counts it accepts; strided (fused-projection) query rows; a workgroup range that
    crosses (video, head) pairs (3 videos x 2 frames x 10 blocks x 4 heads = 240 blocks over 5 workgroups)."""
    B, H, D, Nq, group = 6, 4, 64, 320, 2
    C = H * D
    qw = q16(rnd(B, Nq, 2 * C, seed=75, scale=0.3), dt)
    k1, v1, k2, v2 = (q16(rnd(B // group, n, C, seed=s), dt) for n, s in ((n1, 76), (n1, 77), (n2, 78), (n2, 79)))
    rep = lambda t: t.repeat_interleave(group, 0)
    q = qw[..., C:]
    ref = OU.sdpa(q, rep(k1), rep(v1), H) + 0.5 * OU.sdpa(q, rep(k2), rep(v2), H)
    out = K.attention2(qw.to(dt).cuda()[..., C:], k1.to(dt).cuda(), v1.to(dt).cuda(), k2.to(dt).cuda(), v2.to(dt).cuda(), H,
                       out_scale2=0.5, kv_group=group)
    assert rel(out, ref) < 1.5 * TOL[dt] and blockrel(out, ref, 32) < 3 * TOL[dt]
    assert (out.float().cpu() - ref).abs().max() < 0.05


def test_softmax_rows_and_single_head_attention():
    """The VAE's d = 512 single-head attention as GEMM -> fp32 row softmax -> GEMM (kernels.single_head_attention) and
    the row-softmax kernel alone, against fp32 torch."""
    g = torch.Generator().manual_seed(80)
    for dt in DTYPES:
        x = q16(torch.randn(37, 1000, generator=g) * 3, dt)
        y = K.softmax_rows(x.to(dt).cuda(), 0.37)
        assert rel(y, torch.softmax(x * 0.37, -1)) < TOL[dt]
        q, k, v = (q16(torch.randn(n, 512, generator=g), dt) for n in (320, 1184, 1184))
        s = (q @ k.t()).to(dt).float() * 512 ** -0.5                      # the reference rounds the scores to 16 bits
        ref = torch.softmax(s, -1).to(dt).float() @ v
        out = K.single_head_attention(q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda(), 512 ** -0.5)
        assert rel(out, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_two_query_blocks_per_wave_variant(dt):
    """One (knob attn_qb = 1) and two (= 2: one K fragment read feeds two query blocks; the default for d = 32 on large
    grids) query blocks per wave, forced: d = 32 with the shared bias and d = 64, ragged key count."""
    g = torch.Generator().manual_seed(85)
    try:
        for H, D, Nq, Nk, has_bias in ((4, 32, 600, 328, True), (2, 64, 512, 200, False)):
            q, k, v = (q16(torch.randn(2, n, H * D, generator=g), dt) for n in (Nq, Nk, Nk))
            bias = q16(torch.rand(Nq, Nk, generator=g) * 2 - 1, dt) if has_bias else None
            outs = []
            for qb in (1, 2):
                K.tuning_set("attn_qb", qb)
                outs.append(K.attention(q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda(), H, bias=None if bias is None else bias.to(dt).cuda()))
                assert rel(outs[-1], OU.sdpa(q, k, v, H, bias=bias)) < TOL[dt]
            assert rel(outs[0], outs[1].float().cpu()) < TOL[dt]       # (the variants rescale at different points: not bit-identical)
    finally:
        K.tuning_set("attn_qb", 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 4, 600, 328), (1, 2, 40, 80), (3, 3, 100, 1280), (2, 10, 2048, 512)])
def test_attention_packed_bias_through_the_matrix_pipe(dt, B, H, Nq, Nk):
    """WarpAttn's additive mask as fp16 * log2(e) fragments added by two f16 MFMAs per score block (dtype + 256,
    kernels.pack_attn_bias) instead of unpack + FMA on the vector ALU: against the fp32 oracle and the unpacked-bias kernel,
    ragged key counts (Nk % 8 == 0), 1 / 2 / 4 waves, one and two query blocks per wave, and the device-side mask switch."""
    g = torch.Generator().manual_seed(86)
    D = 32
    q, k, v = (q16(torch.randn(B, n, H * D, generator=g), dt) for n in (Nq, Nk, Nk))
    bias = q16(torch.where(torch.rand(Nq, Nk, generator=g) < 0.3, 1.0, -1.0) + 0.25 * torch.randn(Nq, Nk, generator=g), dt)
    alt = q16(-bias, dt)
    dq, dk, dv, db, da = (t.to(dt).cuda() for t in (q, k, v, bias, alt))
    pb, pa = K.pack_attn_bias(db), K.pack_attn_bias(da)
    assert pb.dtype == torch.float16 and torch.allclose(pb.float().cpu(), bias * 1.4426950408889634, rtol=1e-3, atol=1e-3)
    ref, ref_alt = OU.sdpa(q, k, v, H, bias=bias), OU.sdpa(q, k, v, H, bias=alt)
    try:
        for qb in (1, 2):
            K.tuning_set("attn_qb", qb)
            out = K.attention(dq, dk, dv, H, bias=pb, bias_packed=True)
            assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt], qb
            assert rel(out, K.attention(dq, dk, dv, H, bias=db).float().cpu()) < TOL[dt]
            if qb == 2 and K.ablate_build():      # knob attn_hl = 2: the wave's mask rows fetched coalesced and passed through its LDS patch instead of
                try:         # per-lane fragments straight from global memory -- the same arithmetic, so the same bits
                    K.tuning_set("attn_hl", 2)
                    assert torch.equal(K.attention(dq, dk, dv, H, bias=pb, bias_packed=True), out)
                finally:
                    K.tuning_set("attn_hl", 0)
            for flag, want in ((0, ref), (1, ref_alt)):
                sel = torch.tensor([flag], dtype=torch.int32, device="cuda")
                assert rel(K.attention(dq, dk, dv, H, bias=pb, bias_alt=pa, bias_sel=sel, bias_packed=True), want) < TOL[dt]
    finally:
        K.tuning_set("attn_qb", 0)
    with pytest.raises(RuntimeError, match="head dim 32"):
        x = torch.zeros(1, 64, 128, dtype=dt, device="cuda")
        K.attention(x, x, x, 2, bias=torch.zeros(64, 64, dtype=torch.float16, device="cuda"), bias_packed=True)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 4, 2048, 5120), (3, 2, 576, 1288), (1, 3, 64, 2112), (2, 2, 4096, 1024)])
def test_attention_block_map_skips_background_blocks(dt, B, H, Nq, Nk):
    """Round 6: WarpAttn's masks shifted so that the background is exactly zero (softmax is invariant under a per-row constant) +
    one bit per (32-query block, 32-key half) of the packed matrix (kernels.attn_bias_blocks): blocks with a clear bit skip their
    fragment loads and their two bias MFMAs.  A skipped MFMA would have added exact zeros, so the result with the map is BIT-identical
    to the result without it; both match the fp32 oracle on the UNSHIFTED mask.  Sizes: WarpAttn level 1, ragged query / key counts
    (map rows and words past the end), more than 1024 keys (a second map word; the look-ahead across the word boundary), the
    device-side switch between two matrices with different maps."""
    g = torch.Generator().manual_seed(91)
    D = 32
    q, k, v = (q16(torch.randn(B, n, H * D, generator=g), dt) for n in (Nq, Nk, Nk))

    def sparse_mask(seed):
        gg = torch.Generator().manual_seed(seed)
        m = torch.full((Nq, Nk), -1.0)
        for _ in range(max(3, Nq * Nk // 200000)):                      # a few soft foreground patches on a -1 background
            r0, c0 = int(torch.randint(0, Nq, (1,), generator=gg)), int(torch.randint(0, Nk, (1,), generator=gg))
            h, w = int(torch.randint(8, 80, (1,), generator=gg)), int(torch.randint(8, 200, (1,), generator=gg))
            m[r0:r0 + h, c0:c0 + w] = torch.rand(min(h, Nq - r0), min(w, Nk - c0), generator=gg) * 2 - 1
        return q16(m, dt)

    bias, alt = sparse_mask(1), sparse_mask(2)
    ref, ref_alt = OU.sdpa(q, k, v, H, bias=bias), OU.sdpa(q, k, v, H, bias=alt)
    dq, dk, dv = (t.to(dt).cuda() for t in (q, k, v))

    def pack(b):
        pm = ((b.float().cuda() + 1.0) * 1.4426950408889634).to(torch.float16).contiguous()
        return pm, K.attn_bias_blocks(pm)

    (pb, mb), (pa, ma) = pack(bias), pack(alt)
    want_bits = torch.zeros((-(-Nq // 32) * 32, -(-Nk // 32) * 32), dtype=torch.bool, device="cuda")
    want_bits[:Nq, :Nk] = pb != 0
    want_bits = want_bits.view(-1, 32, want_bits.shape[1] // 32, 32).any(3).any(1)
    words = mb.to(torch.int64) & 0xffffffff
    got_bits = ((words[:, :, None] >> torch.arange(32, device="cuda")) & 1).reshape(words.shape[0], -1)[:, :want_bits.shape[1]].bool()
    assert torch.equal(got_bits, want_bits) and 0.0 < float(want_bits.float().mean()) < 0.5
    try:
        for qb in (1, 2):                # (the map is used by the two-query-block kernel -- WarpAttn's; with one block per wave it is ignored)
            K.tuning_set("attn_qb", qb)
            plain = K.attention(dq, dk, dv, H, bias=pb, bias_packed=True)
            mapped = K.attention(dq, dk, dv, H, bias=pb, bias_packed=True, bias_blocks=mb)
            assert torch.equal(plain, mapped), qb
            assert rel(mapped, ref) < TOL[dt] and blockrel(mapped, ref, 32) < 2 * TOL[dt], qb
            for flag, want in ((0, ref), (1, ref_alt)):
                sel = torch.tensor([flag], dtype=torch.int32, device="cuda")
                out = K.attention(dq, dk, dv, H, bias=pb, bias_alt=pa, bias_sel=sel, bias_packed=True, bias_blocks=mb, bias_blocks_alt=ma)
                assert rel(out, want) < TOL[dt] and blockrel(out, want, 32) < 2 * TOL[dt], (qb, flag)
                assert torch.equal(out, K.attention(dq, dk, dv, H, bias=pb, bias_alt=pa, bias_sel=sel, bias_packed=True)), (qb, flag)
    finally:
        K.tuning_set("attn_qb", 0)
    with pytest.raises(RuntimeError, match="block map"):
        K.attention(dq, dk, dv, H, bias=bias.to(dt).cuda(), bias_blocks=mb)


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_dot_sum_variant(dt):
    """Knob attn_ds: row sums as dot2's over the packed (rounded) weights and the half-wave max exchange through
    v_permlane32_swap, on the four-wave kernels: d = 64 (one / two query blocks per wave, ragged keys, a spiked key that moves
    the running max late) and d = 32 with the packed bias -- against the oracle and close to the plain variant."""
    if not K.ablate_build():
        pytest.skip("rejected A/B variant: only in `make ablate` builds of the library")
    g = torch.Generator().manual_seed(87)
    try:
        for H, D, Nq, Nk, has_bias, qb in ((2, 64, 512, 200, False, 1), (2, 64, 512, 1096, False, 2), (4, 32, 600, 328, True, 2), (4, 32, 256, 640, True, 1)):
            q, k, v = (q16(torch.randn(2, n, H * D, generator=g), dt) for n in (Nq, Nk, Nk))
            k[:, Nk - 7] *= 6.0
            k = q16(k, dt)
            bias = q16(torch.rand(Nq, Nk, generator=g) * 2 - 1, dt) if has_bias else None
            db = None if bias is None else K.pack_attn_bias(bias.to(dt).cuda())
            ref = OU.sdpa(q, k, v, H, bias=bias)
            K.tuning_set("attn_qb", qb)
            outs = []
            for ds in (0, 1):
                K.tuning_set("attn_ds", ds)
                outs.append(K.attention(q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda(), H, bias=db, bias_packed=has_bias))
                assert rel(outs[-1], ref) < TOL[dt] and blockrel(outs[-1], ref, 32) < 2 * TOL[dt], (D, qb, ds)
            assert rel(outs[1], outs[0].float().cpu()) < TOL[dt]
    finally:
        K.tuning_set("attn_qb", 0)
        K.tuning_set("attn_ds", 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 10, 2048, 512), (4, 3, 100, 328), (1, 4, 64, 1280)])
def test_attention_head_groups_share_the_mask(dt, B, H, Nq, Nk):
    """Knob attn_hg: the four waves of a workgroup take four (batch, head) pairs over the same 64 query rows (mask fragments
    then come out of the CU's L1) instead of four row blocks of one pair; every wave stages its own K / V.  Same arithmetic
    per (pair, row): bit-identical to the two-query-block kernel, ragged rows / keys and the device-side mask switch included."""
    if not K.ablate_build():
        pytest.skip("rejected A/B variant: only in `make ablate` builds of the library")
    g = torch.Generator().manual_seed(88)
    D = 32
    q, k, v = (q16(torch.randn(B, n, H * D, generator=g), dt) for n in (Nq, Nk, Nk))
    bias = q16(torch.where(torch.rand(Nq, Nk, generator=g) < 0.3, 1.0, -1.0) + 0.25 * torch.randn(Nq, Nk, generator=g), dt)
    dq, dk, dv = (t.to(dt).cuda() for t in (q, k, v))
    pb, pa = K.pack_attn_bias(bias.to(dt).cuda()), K.pack_attn_bias((-bias).to(dt).cuda())
    sel = torch.tensor([1], dtype=torch.int32, device="cuda")
    try:
        K.tuning_set("attn_qb", 2)
        outs = []
        for hg in (0, 1):
            K.tuning_set("attn_hg", hg)
            outs.append((K.attention(dq, dk, dv, H, bias=pb, bias_packed=True),
                         K.attention(dq, dk, dv, H, bias=pb, bias_alt=pa, bias_sel=sel, bias_packed=True)))
        assert rel(outs[1][0], OU.sdpa(q, k, v, H, bias=bias)) < TOL[dt]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    finally:
        K.tuning_set("attn_qb", 0)
        K.tuning_set("attn_hg", 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("knob", [1, 2, 3, 4, 9, 10, 17, 25, 33, 65, 66, 81])
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 256, 192), (1, 2, 200, 128), (1, 1, 2048, 2048), (3, 1, 128, 640), (1, 5, 1024, 1024)])
def test_attention_pipelined_kernel(dt, knob, B, H, Nq, Nk):
    """attn_pipe.hip (software-pipelined d = 64 self-attention, every schedule, four- and eight-wave workgroups) against the
    fp32 oracle and against attn_fwd_kernel: odd and even tile counts, a ragged last query block, a spiked key that forces
    the deferred rescale in a late tile."""
    D = 64
    q, k, v = (q16(rnd(B, n, H * D, seed=s), dt) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    k[0, Nk - 70, :D] = q[0, 5, :D] * 5.0            # head 0: the running max jumps in the second to last tile
    k[0, 9, :D] = q[0, 40, :D] * 7.0                  # ... and in the first one
    ref = OU.sdpa(q, k, v, H)
    qd, kd, vd = q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda()
    try:
        K.tuning_set("attn_pipe", 0)
        base = K.attention(qd, kd, vd, H)             # attn_fwd_kernel
        K.tuning_set("attn_pipe", knob)
        out = K.attention(qd, kd, vd, H)
    finally:
        K.tuning_set("attn_pipe", K.ATTN_PIPE_DEFAULT)
    assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt]
    assert (out.float().cpu() - ref).abs().max() < 0.05
    assert rel(out, base) < 2e-3                      # same arithmetic up to the order of the fp32 row sums


def test_attention_softmax_rescale_branch():
    """Force the running max to jump in a late KV tile (spiked key) -- the online-softmax rescale path."""
    dt = torch.bfloat16
    B, H, D, Nq, Nk = 1, 1, 64, 64, 512
    q, k, v = (q16(rnd(B, n, D, seed=s), dt) for n, s in ((Nq, 12), (Nk, 13), (Nk, 14)))
    k[0, 300] = q[0, 5] * 6.0
    k[0, 450] = q[0, 9] * 9.0
    ref = OU.sdpa(q, k, v, H)
    out = K.attention(q.to(dt).cuda(), k.to(dt).cuda(), v.to(dt).cuda(), H)
    assert rel(out, ref) < TOL[dt]
    assert (out.float().cpu() - ref)[0, [5, 9]].abs().max() < 0.05


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,Fr,P,heads,d", [(2, 16, 96, 8, 40), (3, 8, 33, 8, 8), (1, 48, 20, 8, 16), (2, 16, 7, 8, 160),
                                            (2, 24, 50, 8, 40), (1, 33, 17, 8, 80), (1, 64, 9, 8, 160), (1, 17, 5, 4, 24)])
def test_temporal_attention(dt, B, Fr, P, heads, d):
    C = heads * d
    qkv = q16(rnd(B * Fr * P, 3 * C, seed=15), dt)
    t = qkv.reshape(B, Fr, P, 3 * C).permute(0, 2, 1, 3).reshape(B * P, Fr, 3 * C)      # (b d) f c
    ref = OU.sdpa(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], heads)
    ref = ref.reshape(B, P, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * P, C)
    out = K.temporal_attention(qkv.to(dt).cuda(), B, Fr, P, heads)
    assert rel(out, ref) < TOL[dt]
    try:                                     # the scalar LDS kernel (fallback / knob) on the same problem
        K.tuning_set("tattn_scalar", 1)
        assert rel(K.temporal_attention(qkv.to(dt).cuda(), B, Fr, P, heads), ref) < TOL[dt]
    finally:
        K.tuning_set("tattn_scalar", 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C,pad,silu", [(3, 8, 16, 64, 0, True), (2, 16, 32, 320, 2, True), (5, 4, 4, 1280, 0, False),
                                                (2, 6, 10, 2560, 2, True), (1, 64, 128, 320, 2, True)])
def test_group_norm(dt, N, H, W, C, pad, silu):
    x = q16(rnd(N, H, W, C, seed=16) + 0.5, dt)
    gamma, beta = q16(1 + 0.1 * rnd(C, seed=17), dt), q16(0.1 * rnd(C, seed=18), dt)
    xr = OG.pad_pano(x.permute(0, 3, 1, 2), pad)
    ref = F.group_norm(xr, 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    out = K.group_norm(x.to(dt).cuda(), gamma.to(dt).cuda(), beta.to(dt).cuda(), 32, 1e-5, silu=silu, pad=pad)
    assert out.shape == ref.shape
    assert rel(out, ref) < TOL[dt]


def _conv_ref(x, w, b, stride=1, up=False, wrap_pad=0, unpad=0):
    xr = x.permute(0, 3, 1, 2)
    if wrap_pad:
        xr = OG.pad_pano(xr, wrap_pad)
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xr, w, b, stride=stride, padding=w.shape[-1] // 2)
    return OG.unpad_pano(y, unpad).permute(0, 2, 3, 1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (3, 8, 12, 320, 640), (5, 4, 4, 256, 128), (1, 32, 64, 32, 4),
                                             (2, 9, 7, 96, 200)])
def test_conv3x3_plain(dt, N, H, W, Cin, Cout):
    x = q16(rnd(N, H, W, Cin, seed=19), dt)
    w = q16(rnd(Cout, Cin, 3, 3, seed=20, scale=(9 * Cin) ** -0.5), dt)
    b = q16(rnd(Cout, seed=21, scale=0.1), dt)
    ref = _conv_ref(x, w, b)
    wp = K.pack_conv_weight(w.to(dt).cuda())
    out = K.conv2d(x.to(dt).cuda(), wp, Cout, bias=b.to(dt).cuda())
    assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt]      # per 32 output pixels: tile / image edges included


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_variants_pano(dt):
    """The pano-branch addressing modes against pad_pano -> conv -> unpad_pano of the reference."""
    N, H, W, C, Co = 2, 8, 16, 64, 96
    x = q16(rnd(N, H, W, C, seed=22), dt)
    w = q16(rnd(Co, C, 3, 3, seed=23, scale=(9 * C) ** -0.5), dt)
    b = q16(rnd(Co, seed=24, scale=0.1), dt)
    dx, db = x.to(dt).cuda(), b.to(dt).cuda()
    wp = K.pack_conv_weight(w.to(dt).cuda())
    # conv_in / conv_out: pad 1 -> conv -> unpad 1  == circular conv
    assert rel(K.conv2d(dx, wp, Co, bias=db, wrap=True), _conv_ref(x, w, b, wrap_pad=1, unpad=1)) < TOL[dt]
    # downsampler: pad 2 -> conv stride 2 -> unpad 1
    assert rel(K.conv2d(dx, wp, Co, bias=db, wrap=True, stride=2), _conv_ref(x, w, b, stride=2, wrap_pad=2, unpad=1)) < TOL[dt]
    assert rel(K.conv2d(dx, wp, Co, bias=db, stride=2), _conv_ref(x, w, b, stride=2)) < TOL[dt]
    # upsampler: pad 1 -> nearest x2 -> conv -> unpad 2
    assert rel(K.conv2d(dx, wp, Co, bias=db, wrap=True, up=True), _conv_ref(x, w, b, up=True, wrap_pad=1, unpad=2)) < TOL[dt]
    assert rel(K.conv2d(dx, wp, Co, bias=db, up=True), _conv_ref(x, w, b, up=True)) < TOL[dt]
    # resnet conv2 on the pre-padded tensor: read columns 2 .. W+1 of a W+4 wide input
    xp = OG.pad_pano(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
    ref = OG.unpad_pano(F.conv2d(xp.permute(0, 3, 1, 2), w, b, padding=1), 2).permute(0, 2, 3, 1)
    assert rel(K.conv2d(xp.to(dt).cuda(), wp, Co, bias=db, x_off=2, wout=W), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_epilogue_temb_residual_and_1x1(dt):
    N, Fr, H, W, C, Co = 6, 3, 8, 8, 128, 64
    x = q16(rnd(N, H, W, C, seed=25), dt)
    w = q16(rnd(Co, C, 3, 3, seed=26, scale=(9 * C) ** -0.5), dt)
    w1 = q16(rnd(Co, C, 1, 1, seed=27, scale=C ** -0.5), dt)
    b = q16(rnd(Co, seed=28, scale=0.1), dt)
    temb = q16(rnd(N // Fr, Co, seed=29), dt)
    res = _conv_ref(x, w1, b)                      # 1x1 shortcut
    short = K.conv2d(x.to(dt).cuda(), K.pack_conv_weight(w1.to(dt).cuda()), Co, bias=b.to(dt).cuda())
    assert rel(short, res) < TOL[dt]
    ref = _conv_ref(x, w, b) + temb.repeat_interleave(Fr, 0)[:, None, None, :] + q16(res, dt)
    out = K.conv2d(x.to(dt).cuda(), K.pack_conv_weight(w.to(dt).cuda()), Co, bias=b.to(dt).cuda(),
                   temb=temb.to(dt).cuda(), imgs_per_temb=Fr, res=q16(res, dt).to(dt).cuda())
    assert rel(out, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,wrap", [(128, 32, 32, 64, 320, False), (33, 32, 64, 128, 640, True)])
def test_conv3x3_256x320_tiles(dt, N, H, W, Cin, Cout, wrap):
    """Shapes large enough (>= 512 workgroups of 256 pixels x 320 couts) to take the 8-wave big-tile kernel, including
    a ragged last pixel tile, the circular wrap, and the temb / residual epilogue."""
    Fr = N // 11 if N % 11 == 0 else 1
    x = q16(rnd(N, H, W, Cin, seed=40), dt)
    w = q16(rnd(Cout, Cin, 3, 3, seed=41, scale=(9 * Cin) ** -0.5), dt)
    b = q16(rnd(Cout, seed=42, scale=0.1), dt)
    temb = q16(rnd(N // Fr, Cout, seed=43), dt)
    res = q16(rnd(N, H, W, Cout, seed=44), dt)
    ref = _conv_ref(x, w, b, wrap_pad=1, unpad=1) if wrap else _conv_ref(x, w, b)
    ref = ref + temb.repeat_interleave(Fr, 0)[:, None, None, :] + res
    out = K.conv2d(x.to(dt).cuda(), K.pack_conv_weight(w.to(dt).cuda()), Cout, bias=b.to(dt).cuda(), wrap=wrap,
                   temb=temb.to(dt).cuda(), imgs_per_temb=Fr, res=res.to(dt).cuda())
    assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_conv3x3_256x320_tiles_addressing_modes(dt):
    """The stride-2 / nearest-upsample / pre-padded-window addressing modes at pixel counts that take the 8-wave
    256 x 320 tile (the cfg2 Downsample3D / Upsample3D / pano conv2 launches), against the reference composition."""
    C, Co = 64, 320
    w = q16(rnd(Co, C, 3, 3, seed=71, scale=(9 * C) ** -0.5), dt)
    b = q16(rnd(Co, seed=72, scale=0.1), dt)
    wp, db = K.pack_conv_weight(w.to(dt).cuda()), b.to(dt).cuda()
    x = q16(rnd(640, 32, 32, C, seed=73), dt)                       # stride 2: 640 x 16 x 16 outputs = 640 tiles
    assert rel(K.conv2d(x.to(dt).cuda(), wp, Co, bias=db, stride=2), _conv_ref(x, w, b, stride=2)) < TOL[dt]
    assert rel(K.conv2d(x.to(dt).cuda(), wp, Co, bias=db, stride=2, wrap=True),
               _conv_ref(x, w, b, stride=2, wrap_pad=2, unpad=1)) < TOL[dt]
    x = q16(rnd(128, 16, 16, C, seed=74), dt)                       # nearest x2: 128 x 32 x 32 outputs = 512 tiles
    assert rel(K.conv2d(x.to(dt).cuda(), wp, Co, bias=db, up=True), _conv_ref(x, w, b, up=True)) < TOL[dt]
    x = q16(rnd(64, 32, 64, C, seed=75), dt)                        # window of the W+4 padded tensor, 512 tiles
    xp = OG.pad_pano(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
    ref = OG.unpad_pano(F.conv2d(xp.permute(0, 3, 1, 2), w, b, padding=1), 2).permute(0, 2, 3, 1)
    assert rel(K.conv2d(xp.to(dt).cuda(), wp, Co, bias=db, x_off=2, wout=64), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,wrap", [(512, 16, 16, 64, 320, False),      # 256 x 320 tiles (512 of them)
                                                 (32, 32, 64, 64, 640, True),        # panorama: circular along W
                                                 (7, 5, 9, 128, 96, False),          # ragged, 128 x 128 tiles, Cout not a tile multiple
                                                 (3, 4, 6, 64, 64, True)])
def test_conv_up2_subpixel_upsample(dt, N, H, W, Cin, Cout, wrap):
    """Upsample3D's nearest-x2 + conv3x3 as four 2 x 2 convolutions of the low-resolution input (im360_conv_up2_fwd,
    pre-summed taps, 4 / 9 of the MACs) against the fp32 upsample + conv reference and against the 9-tap kernel path."""
    g = torch.Generator().manual_seed(92)
    x = q16(torch.randn(N, H, W, Cin, generator=g), dt)
    w = q16(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, dt)
    b = q16(torch.randn(Cout, generator=g) * 0.1, dt)
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    if wrap:
        ref = F.conv2d(F.pad(torch.cat([up[..., -1:], up, up[..., :1]], dim=-1), (0, 0, 1, 1)), w, b)
    else:
        ref = F.conv2d(up, w, b, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    dx, dw, db = x.to(dt).cuda(), w.to(dt).cuda(), b.to(dt).cuda()
    w4 = K.pack_conv_up2_weight(dw)
    assert w4.shape == (4, (Cout + 127) // 128 * 128, 4, Cin)
    out = K.conv_up2(dx, w4, Cout, bias=db, wrap=wrap)
    assert out.shape == (N, 2 * H, 2 * W, Cout) and rel(out, ref) < TOL[dt]
    old = K.conv2d(dx, K.pack_conv_weight(dw), Cout, bias=db, up=True, wrap=wrap)
    assert rel(out, old.float().cpu()) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_linear_residual_through_gemm_kernel(dt):
    """Token counts large enough for layers.linear_residual to take the implicit-GEMM kernel (bias + residual in the
    epilogue) instead of hipBLASLt + add; both must agree with the fp32 reference."""
    from imagine360_amd import layers
    M, Kd, N = 262144, 320, 320
    assert layers._gemm_kernel_pays(M, Kd, N) and not layers._gemm_kernel_pays(M // 8, Kd, N)
    lin = torch.nn.Linear(Kd, N).to(dt).cuda()
    g = torch.Generator().manual_seed(50)
    x = torch.randn(4, M // 4, Kd, generator=g).to(dt).cuda()
    res = torch.randn(4, M // 4, N, generator=g).to(dt).cuda()
    ref = F.linear(x.float(), lin.weight.float(), lin.bias.float()) + res.float()
    out = layers.linear_residual(lin, x, res)
    assert out.shape == res.shape and rel(out.cpu(), ref.cpu()) < TOL[dt] and blockrel(out, ref) < 2 * TOL[dt]
    small = layers.linear_residual(lin, x[:1, :1000], res[:1, :1000])        # hipBLASLt + add path
    assert rel(small.cpu(), ref[:1, :1000].cpu()) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_pipeline_variants_are_bit_identical(dt):
    """The persistent ring kernel (knob conv_ring 1: interleaved LDS-DMA, 3: plain ring, 4: staggered wave groups, 5: also
    for convolutions) accumulates in the same order as the two-stage kernel (0): identical bits, on a ragged token count,
    a GEGLU projection and a 3x3 convolution with temb + residual; and the result matches the fp32 reference."""
    g = torch.Generator().manual_seed(60)
    M, Kd, N = 131073, 128, 320
    x = torch.randn(M, 1, 1, Kd, generator=g).to(dt).cuda()
    w = (torch.randn(N, Kd, 1, 1, generator=g) * Kd ** -0.5).to(dt).cuda()
    b, r = torch.randn(N, generator=g).to(dt).cuda(), torch.randn(M, 1, 1, N, generator=g).to(dt).cuda()
    wp = K.pack_conv_weight(w)
    gx = torch.randn(65536, 64, generator=g).to(dt).cuda()
    gw, gb = (torch.randn(512, 64, generator=g) * 0.125).to(dt).cuda(), torch.randn(512, generator=g).to(dt).cuda()
    gwp, gbp = K.pack_geglu(gw, gb)
    cx = torch.randn(145, 30, 31, 64, generator=g).to(dt).cuda()
    cw = (torch.randn(320, 64, 3, 3, generator=g) * 576 ** -0.5).to(dt).cuda()
    cwp = K.pack_conv_weight(cw)
    temb, cres = torch.randn(29, 320, generator=g).to(dt).cuda(), torch.randn(145, 30, 31, 320, generator=g).to(dt).cuda()
    outs = {}
    try:
        K.tuning_set("conv_cm", 0)          # the ring kernel sums taps outermost: compare it with the tap-major two-stage kernel
        for v in (0, 1, 3, 4, 5, 8):
            K.tuning_set("conv_ring", v)
            outs[v] = (K.conv2d(x, wp, N, bias=b, res=r), K.linear_geglu(gx, gwp, gbp, 256),
                       K.conv2d(cx, cwp, 320, bias=b, temb=temb, imgs_per_temb=5, res=cres))
    finally:
        K.tuning_set("conv_ring", 1)
        K.tuning_set("conv_cm", 1)
    for v in (1, 3, 4, 5, 8):
        for a, ref in zip(outs[v], outs[0]):
            assert torch.equal(a, ref), v
    lin_ref = (x[:, 0, 0].float() @ w[:, :, 0, 0].float().t() + b.float()).to(dt).float() + r[:, 0, 0].float()
    assert rel(outs[1][0][:, 0, 0], lin_ref) < TOL[dt]
    conv_ref = (F.conv2d(cx.float().permute(0, 3, 1, 2), cw.float(), b.float(), padding=1).permute(0, 2, 3, 1)
                + temb.float().repeat_interleave(5, 0)[:, None, None, :]).to(dt).float() + cres.float()
    assert rel(outs[5][2], conv_ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("knob", ["conv_stag", "conv_persist"])
def test_conv_loop_variants_of_the_ablation_build_are_bit_identical(dt, knob):
    """Round 4's two rejected loops for the 3x3 convolutions (make ablate): staggered wave groups inside conv_igemm_kernel
    (conv_stag) and the chunk-major producer on the persistent shell (conv_persist) accumulate in the default kernel's order:
    identical outputs and GroupNorm partial sums, with time embedding + residual, stride 2, and a ragged image count."""
    if not K.ablate_build():
        pytest.skip("variants are compiled by `make ablate` only")
    g = torch.Generator().manual_seed(62)
    cases = [(160, 32, 32, 320, 320, dict()), (160, 32, 32, 128, 640, dict(stride=2)), (161, 16, 16, 64, 320, dict())]
    for N, H, W, Cin, Cout, kw in cases:
        x = torch.randn(N, H, W, Cin, generator=g).to(dt).cuda()
        wp = K.pack_conv_weight((torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dt).cuda())
        b = torch.randn(Cout, generator=g).to(dt).cuda()
        st = kw.get("stride", 1)
        temb = torch.randn(N, Cout, generator=g).to(dt).cuda()
        res = torch.randn(N, H // st, W // st, Cout, generator=g).to(dt).cuda()
        outs = []
        try:
            for v in (0, 1):
                K.tuning_set(knob, v)
                y = K.conv2d(x, wp, Cout, bias=b, temb=temb, res=res, gn_stats=True, **kw)
                gn = K._gn_of(y)
                outs.append((y.clone(), None if gn is None else gn[0].clone()))
        finally:
            K.tuning_set(knob, 0)
        assert torch.equal(outs[0][0], outs[1][0])
        assert (outs[0][1] is None) == (outs[1][1] is None) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,Kd", [(131073, 320), (65536 + 77, 64), (70000, 1280)])
def test_staggered_gemm_loop_every_epilogue_bit_identical_to_the_ring(dt, M, Kd):
    """Round 4's default loop of the token-major GEMMs (64-channel stages in two buffers, two wave groups one barrier interval
    apart: conv_ring 1) against round 3's (32-channel phases over a four-slot ring: conv_ring 8): same accumulation order, so
    identical bits from every epilogue -- bias + residual, + LayerNorm row statistics, + GroupNorm partial sums, LayerNorm
    folded in (with a positional table), GEGLU, GEGLU with LayerNorm folded in -- on ragged token counts with 1, 5 and 20 stages."""
    g = torch.Generator().manual_seed(61)
    N = 640
    x = (torch.randn(M, Kd, generator=g) + 0.3).to(dt).cuda()
    w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(dt).cuda()
    b, r = torch.randn(N, generator=g).to(dt).cuda(), torch.randn(M, N, generator=g).to(dt).cuda()
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    gam, bet = (1 + 0.1 * torch.randn(Kd, generator=g)).to(dt).cuda(), (0.1 * torch.randn(Kd, generator=g)).to(dt).cuda()
    wg, c1, c2 = K.fold_layer_norm(w, b, gam, bet)
    wgp = K.pack_conv_weight(wg.reshape(N, Kd, 1, 1).contiguous())
    tab = torch.randn(3, N, generator=g).cuda()
    gw, gb = (torch.randn(1024, Kd, generator=g) * Kd ** -0.5).to(dt).cuda(), torch.randn(1024, generator=g).to(dt).cuda()
    gwp, gbp = K.pack_geglu(gw, gb)
    gwf, gc1, gc2 = K.fold_layer_norm(gw, gb, gam, bet)
    gwfp, gc1p = K.pack_geglu(gwf, gc1)
    gc2p = K.interleave_geglu(gwf, gc2)[1].contiguous()
    # row statistics of x itself, in the producer's layout (per 160-column slice; the consumer sums the slices)
    sl = K.ROW_SLICE
    if Kd % sl == 0:
        xs = x.float().reshape(M, Kd // sl, sl)
        st = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()
    else:
        st = torch.stack([x.float().sum(-1, keepdim=True), (x.float() ** 2).sum(-1, keepdim=True)], dim=-1).contiguous()
    m256 = (M // 256) * 256
    outs = {}
    variants = (1, 8, 10) if K.ablate_build() else (1, 8)        # 10: two activation stages in flight (gemm_a3_kernel, make ablate)
    try:
        for v in variants:
            K.tuning_set("conv_ring", v)
            y5, s5 = K.linear(x, wp, N, bias=b, res=r, row_stats=True)
            yg = K.linear(x[:m256], wp, N, bias=b, res=r[:m256], gn_hw=256)
            outs[v] = (K.linear(x, wp, N, bias=b, res=r), y5, s5, yg, K._gn_of(yg)[0] if K._gn_of(yg) is not None else yg,
                       K.linear_ln(x, wgp, c1, c2, st, 1e-5, N, tab=tab, tab_div=256),
                       K.linear_geglu(x, gwp, gbp, 512), K.linear_geglu_ln(x, gwfp, gc1p.contiguous(), gc2p, st, 1e-5, 512))
    finally:
        K.tuning_set("conv_ring", 1)
    for v in variants[1:]:
        for i, (a, ref) in enumerate(zip(outs[1], outs[v])):
            assert torch.equal(a, ref), (v, i)
    ref = (x.float() @ w.float().t() + b.float()).to(dt).float() + r.float()
    assert rel(outs[1][0], ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,kw", [(640, 32, 32, 128, 320, dict()),                       # 256 x 320 tiles
                                               (32, 64, 132, 64, 320, dict(x_off=2, wout=128)),       # window of the W+4 tensor
                                               (320, 32, 32, 64, 320, dict(stride=2)),                # stride 2 (uniform tap offsets too)
                                               (145, 30, 31, 192, 96, dict()),                        # ragged, 128 x 128 tiles
                                               (3, 5, 7, 64, 64, dict())])                            # images smaller than a tile
def test_conv3x3_chunk_major_k_order(dt, N, H, W, Cin, Cout, kw):
    """Taps-innermost K order (knob conv_cm, default for 3x3 convs without wrap / upsample addressing: eight of the nine
    shifted reads of a channel chunk hit L2) against the tap-major order and the fp32 reference, with bias, time embedding
    and residual; the two orders differ only in fp32 summation order."""
    g = torch.Generator().manual_seed(91)
    x = q16(torch.randn(N, H, W, Cin, generator=g), dt)
    w = q16(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, dt)
    b = q16(torch.randn(Cout, generator=g) * 0.1, dt)
    stride, x_off, wout = kw.get("stride", 1), kw.get("x_off", 0), kw.get("wout", W)
    xr = x.permute(0, 3, 1, 2)
    if x_off:
        ref = F.conv2d(xr, w, b, padding=1)[..., x_off:x_off + wout].permute(0, 2, 3, 1)
    else:
        ref = F.conv2d(xr, w, b, padding=1, stride=stride).permute(0, 2, 3, 1)
    res = q16(torch.randn(ref.shape, generator=g), dt)
    ref = q16(ref, dt) + res
    wp = K.pack_conv_weight(w.to(dt).cuda())
    outs = []
    try:
        for cm in (1, 0):
            K.tuning_set("conv_cm", cm)
            outs.append(K.conv2d(x.to(dt).cuda(), wp, Cout, bias=b.to(dt).cuda(), res=res.to(dt).cuda(), **kw))
    finally:
        K.tuning_set("conv_cm", 1)
    assert rel(outs[0], ref) < TOL[dt] and rel(outs[1], ref) < TOL[dt]
    assert rel(outs[0], outs[1].float().cpu()) < TOL[dt] / 4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,kw", [(160, 8, 8, 256, 1280, dict()),                  # 160 tiles (level-3 like): below one round of the chip
                                               (100, 16, 16, 192, 640, dict()),                 # 200 tiles, three chunks: parts of 1 + 1 + 1 / 1 + 2 chunks
                                               (77, 16, 16, 128, 320, dict()),                  # ragged last pixel tile
                                               (320, 16, 16, 128, 640, dict(stride=2)),         # stride 2
                                               (32, 16, 36, 256, 1280, dict(wrap=True)),        # panorama level-2 like: circular wrap = tap-major K order, parts start inside a tap
                                               (32, 8, 20, 192, 1280, dict(wrap=True))])        # level-3 like, 27 steps over 2 - 3 parts
def test_conv3x3_k_split_equals_the_unsplit_launch(dt, N, H, W, Cin, Cout, kw):
    """Knob conv_ksplit (round 6): every 256 x 320 tile of a 3 x 3 convolution computed by 2 - 4 workgroups over contiguous ranges of the
    64-channel chunks, partial sums through a scratch buffer in a fixed order -- against the unsplit launch (fp32 summation order only), the
    fp32 reference, itself (deterministic), with bias, time embedding and residual; the planner's own choice (knob 1) as well."""
    g = torch.Generator().manual_seed(97)
    x = q16(torch.randn(N, H, W, Cin, generator=g), dt)
    w = q16(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, dt)
    b = q16(torch.randn(Cout, generator=g) * 0.1, dt)
    stride, wrap = kw.get("stride", 1), kw.get("wrap", False)
    ref = _conv_ref(x, w, b, wrap_pad=1, unpad=1) if wrap else F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1, stride=stride).permute(0, 2, 3, 1)
    temb = q16(torch.randn(N, Cout, generator=g), dt)
    res = q16(torch.randn(ref.shape, generator=g), dt)
    ref = ref + temb[:, None, None, :] + res
    wp = K.pack_conv_weight(w.to(dt).cuda())
    args = dict(bias=b.to(dt).cuda(), temb=temb.to(dt).cuda(), imgs_per_temb=1, res=res.to(dt).cuda(), **kw)
    xd = x.to(dt).cuda()
    lib = K.lib()
    ho, wo = H // stride, W // stride
    try:
        K.tuning_set("conv_ksplit", 0)
        assert lib.im360_conv_ksplit_plan(N, ho, wo, Cin, Cout, 9, 0, int(wrap), 0) == 1
        base = K.conv2d(xd, wp, Cout, **args)
        assert rel(base, ref) < TOL[dt]
        for knob in (2, 3, 4, 1, 5, 9):
            K.tuning_set("conv_ksplit", knob)
            parts = lib.im360_conv_ksplit_plan(N, ho, wo, Cin, Cout, 9, 0, int(wrap), 0)
            if 2 <= knob <= 4:
                assert parts == (knob if knob <= Cin // 64 else 1), (knob, parts)
            assert lib.im360_conv_ksplit_plan(N, ho, wo, Cin, Cout, 9, 1, int(wrap), 0) == 1        # upsample addressing: no split
            out = K.conv2d(xd, wp, Cout, **args)
            assert rel(out, ref) < TOL[dt] and blockrel(out, ref, 32) < 2 * TOL[dt], (knob, parts)
            assert rel(out, base.float().cpu()) < TOL[dt] / 8, (knob, parts)
            assert torch.equal(out, K.conv2d(xd, wp, Cout, **args)), (knob, parts)           # fixed order; the counters came back to zero
    finally:
        K.tuning_set("conv_ksplit", 1)          # the library's default: the planner's rule


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Win,Cin,Cout,x_off", [(128, 32, 32, 32, 64, 320, 0),      # 8 rows x 32 of one image per tile
                                                       (512, 16, 16, 16, 96, 320, 0),      # a whole 16 x 16 image per tile
                                                       (2048, 8, 8, 8, 64, 320, 0),        # four 8 x 8 images per tile
                                                       (16, 64, 128, 132, 64, 320, 2),     # window of the W+4 tensor, 2 rows per tile
                                                       (256, 16, 16, 16, 64, 640, 0)])     # two cout tiles
def test_conv3x3_halo_patch_kernel(dt, N, H, W, Win, Cin, Cout, x_off):
    """The halo-patch kernel (knob conv_halo): rectangular pixel tiles, the (rows + 2) x (W + 2) input patch of a
    32-channel chunk in LDS once, nine taps out of it -- against the fp32 reference and against the streaming kernel."""
    if not K.ablate_build():
        pytest.skip("rejected A/B variant: only in `make ablate` builds of the library")
    g = torch.Generator().manual_seed(90)
    x = q16(torch.randn(N, H, Win, Cin, generator=g), dt)
    w = q16(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, dt)
    b = q16(torch.randn(Cout, generator=g) * 0.1, dt)
    fr = 4
    temb = q16(torch.randn(N // fr, Cout, generator=g), dt)
    res = q16(torch.randn(N, H, W, Cout, generator=g), dt)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)[..., x_off:x_off + W].permute(0, 2, 3, 1)
    ref = (ref + temb.repeat_interleave(fr, 0)[:, None, None, :]).to(dt).float() + res
    wp = K.pack_conv_weight(w.to(dt).cuda())
    args = dict(bias=b.to(dt).cuda(), temb=temb.to(dt).cuda(), imgs_per_temb=fr, res=res.to(dt).cuda(), x_off=x_off, wout=W)
    try:
        K.tuning_set("conv_halo", 1)
        out = K.conv2d(x.to(dt).cuda(), wp, Cout, **args)
        K.tuning_set("conv_halo", 0)
        old = K.conv2d(x.to(dt).cuda(), wp, Cout, **args)
    finally:
        K.tuning_set("conv_halo", 0)
    assert rel(out, ref) < TOL[dt] and rel(old, ref) < TOL[dt]
    assert rel(out, old) < (3e-3 if dt == torch.bfloat16 else 4e-4)


@pytest.mark.parametrize("dt", DTYPES)
def test_linear_geglu_fused(dt):
    """GEGLU projection + activation in one GEMM launch (interleaved value / gate weight rows) vs Linear -> chunk ->
    a * gelu(gate) in fp32, including a ragged last token tile; and the module-level switch in layers.GEGLU."""
    from imagine360_amd import layers
    M, Kd, I = 40000 + 77, 128, 256
    g = torch.Generator().manual_seed(60)
    x = q16(torch.randn(M, Kd, generator=g), dt)
    w = q16(torch.randn(2 * I, Kd, generator=g) * Kd ** -0.5, dt)
    b = q16(torch.randn(2 * I, generator=g) * 0.1, dt)
    h = q16(F.linear(x, w, b), dt)                      # the two-kernel path rounds the projection to 16 bits
    ref = h[:, :I] * F.gelu(h[:, I:])
    wp, bp = K.pack_geglu(w.to(dt).cuda(), b.to(dt).cuda())
    out = K.linear_geglu(x.to(dt).cuda(), wp, bp, I)
    assert out.shape == (M, I) and rel(out.cpu(), ref) < TOL[dt] and blockrel(out, ref) < 2 * TOL[dt]
    mod = layers.GEGLU(320, 1280).to(dt).cuda()
    xx = torch.randn(2, 131072, 320, generator=g).to(dt).cuda()
    fused = mod(xx)                                     # 1024 x 10 tiles: fused path
    unfused = K.geglu(mod.proj(xx))
    assert fused.shape == unfused.shape and rel(fused.cpu(), unfused.cpu()) < TOL[dt]


def test_circular_pad_and_cfg_ddim():
    dt = torch.bfloat16
    x = q16(rnd(3, 5, 16, 8, seed=30), dt)
    ref = OG.pad_pano(x.permute(0, 1, 3, 2), 4).permute(0, 1, 3, 2)
    assert torch.equal(K.circular_pad_w(x.to(dt).cuda(), 4).float().cpu(), ref)      # pure copy: bit exact
    u, c, s = (q16(rnd(1, 4, 16, 8, 16, seed=i), dt) for i in (31, 32, 33))
    g, cx, cv = 7.5, 0.83, -0.41
    ref = cx * s + cv * (u + g * (c - u))
    out = K.cfg_ddim_update(u.to(dt).cuda(), c.to(dt).cuda(), s.to(dt).cuda(), g, cx, cv)
    assert rel(out, ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32, torch.uint8, torch.float64])
def test_sr_close_loop_circular_pad(dt):
    """SURVEY row N4 (sr/video_to_video_model.py:16-29, 99, 160-162): im360_circular_pad_hw against the reference fixture,
    the oracle and torch's own circular pad, bit-exact, for every unit width the launcher can pick (16 / 8 / 4 / 2 / 1
    bytes: odd widths and pads), both axes, pads equal to the axis size, and the module API."""
    import numpy as np
    import os
    from helpers import GOLDEN
    from imagine360_amd import sr_patch
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "sr_pad.npz")).items()}
    if dt == torch.float16:
        assert torch.equal(sr_patch.padding_pano(g["lat"].cuda(), latent=True).cpu(), g["lat_pad16"])
    if dt == torch.float32:
        vp = sr_patch.padding_pano(g["vid"].cuda())
        assert torch.equal(vp.cpu(), g["vid_pad128"])
        assert torch.equal(sr_patch.unpadding_pano(vp).cpu(), g["vid"])
        assert torch.equal(sr_patch.circular_pad(g["fr"].cuda(), (3, 5, 2, 4)).cpu(), g["fr_fit"])
    gen = torch.Generator().manual_seed(5)
    for shape, pad in [((2, 3, 4, 16, 64), (16, 16, 0, 0)), ((1, 3, 9, 31), (5, 2, 3, 1)), ((3, 7, 13), (13, 13, 7, 7)),
                       ((2, 2, 6, 24), (8, 0, 0, 6)), ((1, 1, 5, 33), (1, 0, 0, 0))]:
        x = torch.randint(0, 255, shape, generator=gen).to(dt) if dt == torch.uint8 else torch.randn(shape, generator=gen).to(dt)
        ref = OG.circular_pad(x, pad)
        got = K.circular_pad_hw(x.cuda(), *pad).cpu()
        assert got.shape == ref.shape and torch.equal(got.view(torch.uint8), ref.contiguous().view(torch.uint8)), (shape, pad)
    with pytest.raises(RuntimeError, match="pads"):
        K.circular_pad_hw(torch.zeros(1, 4, 8, device="cuda"), 9, 0)


def test_sr_close_loop_pad_full_size_round_trip():
    """At the SR stage's real size (2x upscaled 1024 x 2048 frames, fp16): pad 128 columns -> the wrap columns equal the
    opposite border, unpad returns the input bit-exactly (size-independent property; no reference at this size)."""
    from imagine360_amd import sr_patch
    x = torch.randn(1, 3, 8, 1024, 2048, device="cuda").half()
    y = sr_patch.padding_pano(x)
    assert y.shape[-1] == 2048 + 256
    assert torch.equal(y[..., :128], x[..., -128:]) and torch.equal(y[..., -128:], x[..., :128])
    assert torch.equal(sr_patch.unpadding_pano(y), x)


def test_bad_arguments_fail_loudly():
    x = torch.zeros(1, 16, 40, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="head dim"):
        K.attention(x, x, x, heads=1)                 # d = 40 unsupported by attn_fwd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.attention(x.cpu(), x.cpu(), x.cpu(), heads=1)
    with pytest.raises(TypeError):
        K.group_norm(torch.zeros(1, 2, 2, 32, device="cuda"), torch.ones(32, device="cuda"), torch.zeros(32, device="cuda"), 32, 1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,C", [(1000, 320), (1003, 320), (5, 320), (999, 640), (2, 640), (77, 64), (513, 1280), (40, 1024), (9, 2048)])
def test_layer_norm_plain_pre_post(dt, rows, C):
    x = q16(rnd(rows, C, seed=40) * 1.3 + 0.2, dt)
    g, b = q16(1 + 0.1 * rnd(C, seed=41), dt), q16(0.1 * rnd(C, seed=42), dt)
    dx, dg, db = x.to(dt).cuda(), g.to(dt).cuda(), b.to(dt).cuda()
    assert rel(K.layer_norm(dx, dg, db, 1e-5), F.layer_norm(x, (C,), g, b, 1e-5)) < TOL[dt]
    pre = q16(rnd(7, C, seed=43), dt)                    # WarpAttn form: LN(x + pe[row % P])
    r = torch.arange(rows)
    assert rel(K.layer_norm(dx, dg, db, 1e-5, pre=pre.to(dt).cuda()), F.layer_norm(x + pre[r % 7], (C,), g, b, 1e-5)) < TOL[dt]
    post = q16(rnd(5, C, seed=44), dt)                   # motion-module form: LN(x) + pe[(row // P) % F]
    ref = F.layer_norm(x, (C,), g, b, 1e-5) + post[(r // 3) % 5]
    assert rel(K.layer_norm(dx, dg, db, 1e-5, post=post.to(dt).cuda(), post_div=3), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_geglu(dt):
    h = q16(rnd(333, 2 * 1280, seed=45) * 2, dt)
    a, g = h.chunk(2, dim=-1)
    assert rel(K.geglu(h.to(dt).cuda()), a * F.gelu(g)) < TOL[dt]


# ------------------------------------------------------------------ round 3: skip pairs, row statistics, folded LayerNorm
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C1,C2,pad", [(3, 8, 16, 64, 64, 0), (2, 16, 32, 640, 320, 2), (4, 8, 8, 1280, 640, 0), (2, 6, 10, 320, 320, 2)])
def test_group_norm_of_a_skip_pair_equals_the_concatenation(dt, N, H, W, C1, C2, pad):
    """GroupNorm statistics / apply reading (x, skip) in place == the same kernels on torch.cat([x, skip]) (statistics to
    fp32 summation order, the apply pass bit for bit), incl. group boundaries that straddle the two tensors
    (960 channels: 30 per group) and the pad-aware statistics."""
    xa, xb = q16(rnd(N, H, W, C1, seed=70) + 0.3, dt).to(dt).cuda(), q16(rnd(N, H, W, C2, seed=71) * 1.5, dt).to(dt).cuda()
    C = C1 + C2
    gamma, beta = (1 + 0.1 * rnd(C, seed=72)).to(dt).cuda(), (0.1 * rnd(C, seed=73)).to(dt).cuda()
    cat = torch.cat([xa, xb], dim=-1).contiguous()
    s1, h1 = K.group_norm_stats((xa, xb), gamma, beta, 32, 1e-5, pad=pad)
    s2, h2 = K.group_norm_stats(cat, gamma, beta, 32, 1e-5, pad=pad)
    assert rel(s1, s2) < 2e-6 and (h1 - h2).abs().max() < 1e-5          # (fp32 partial sums split over threads by channel count)
    y1, y2 = K.group_norm_apply((xa, xb), s2, h2, True, pad=pad), K.group_norm_apply(cat, s2, h2, True, pad=pad)
    assert y1.shape == (N, H, W + 2 * pad, C) and torch.equal(y1, y2)      # same scale / shift -> the same bits
    ref = F.silu(F.group_norm(OG.pad_pano(cat.float().cpu().permute(0, 3, 1, 2), pad), 32, gamma.float().cpu(), beta.float().cpu(), 1e-5)).permute(0, 2, 3, 1)
    assert rel(y1, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C1,C2,Cout", [(640, 16, 16, 640, 320, 320),      # 256 x 320 tiles, 64-channel steps
                                               (3, 8, 12, 128, 64, 128),          # 128 x 128 tiles
                                               (2, 9, 7, 64, 192, 200),           # odd Cout, ragged pixel tile
                                               (5, 4, 4, 1280, 1280, 1280)])
def test_conv1x1_of_a_skip_pair_equals_the_concatenation(dt, N, H, W, C1, C2, Cout):
    """conv_shortcut on (x, skip) read in place: same K order as the 1x1 conv of the materialised concatenation, so the
    results agree bit for bit; + residual + bias; against fp32 too."""
    xa, xb = q16(rnd(N, H, W, C1, seed=74), dt).to(dt).cuda(), q16(rnd(N, H, W, C2, seed=75), dt).to(dt).cuda()
    w = q16(rnd(Cout, C1 + C2, 1, 1, seed=76, scale=(C1 + C2) ** -0.5), dt).to(dt).cuda()
    b, r = q16(rnd(Cout, seed=77, scale=0.1), dt).to(dt).cuda(), q16(rnd(N, H, W, Cout, seed=78), dt).to(dt).cuda()
    wp = K.pack_conv_weight(w)
    cat = torch.cat([xa, xb], dim=-1).contiguous()
    y1 = K.conv1x1_cat(xa, xb, wp, Cout, bias=b, res=r)
    y2 = K.conv2d(cat, wp, Cout, bias=b, res=r)
    assert torch.equal(y1, y2)
    ref = F.linear(cat.float().cpu(), w.float().cpu().reshape(Cout, -1), b.float().cpu()) + r.float().cpu()
    assert rel(y1, ref) < TOL[dt] and blockrel(y1, ref) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,Kd,N", [(70000 + 77, 320, 320), (66000, 640, 640), (300, 320, 960), (65536, 1280, 320)])
def test_linear_row_statistics(dt, M, Kd, N):
    """im360_linear_fwd: same output as the conv-kernel route of the Linear (bit for bit), and the (sum, sum of squares)
    slices it writes in the epilogue == sums over the STORED output, ragged last token tile included."""
    x = q16(rnd(M, Kd, seed=80), dt).to(dt).cuda()
    w = q16(rnd(N, Kd, seed=81, scale=Kd ** -0.5), dt).to(dt).cuda()
    b, r = q16(rnd(N, seed=82, scale=0.2), dt).to(dt).cuda(), q16(rnd(M, N, seed=83) + 0.7, dt).to(dt).cuda()
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    y, st = K.linear(x, wp, N, bias=b, res=r, row_stats=True)
    y0 = K.conv2d(x.reshape(M, 1, 1, Kd), wp, N, bias=b, res=r.reshape(M, 1, 1, N)).reshape(M, N)
    assert torch.equal(y, y0) and torch.equal(K.linear(x, wp, N, bias=b, res=r), y0)
    t = y.double().reshape(M, N // 160, 160)
    ref = torch.stack([t.sum(-1), (t * t).sum(-1)], dim=-1)
    assert st.shape == (M, N // 160, 2) and (st.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    fref = F.linear(x.float(), w.float(), b.float()) + r.float()
    assert rel(y, fref) < TOL[dt] and blockrel(y, fref) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,C,N,frames,pixels,offset", [(70000 + 13, 320, 960, 0, 0, 0.0), (66000, 640, 1920, 0, 0, 2.0),
                                                         (2 * 16 * 2304, 320, 960, 16, 2304, 0.5), (2 * 8 * 256, 320, 320, 8, 256, 4.0)])
def test_linear_with_folded_layer_norm(dt, M, C, N, frames, pixels, offset):
    """Linear(LayerNorm(y) [+ PE[frame]]) as ONE GEMM on the raw rows (im360_linear_ln_fwd) with the statistics the
    producer wrote (im360_linear_fwd), against fp32 torch and against the LayerNorm-kernel + GEMM route; row means up to
    4 sigma (variance from E[x^2] - mu^2)."""
    from imagine360_amd import layers
    g = torch.Generator().manual_seed(90)
    prod = torch.nn.Linear(C, C).to(dt).cuda()
    x0 = (torch.randn(M, C, generator=g)).to(dt).cuda()
    res = (torch.randn(M, C, generator=g) + offset).to(dt).cuda()
    y, st = layers.gemm_linear(prod.weight, prod.bias, x0, res=res, cache=layers.DerivedCache(), row_stats=True)
    saved = layers.ROUTE_MIN_TOKENS
    layers.ROUTE_MIN_TOKENS = 0
    try:
        if st is None:        # small M: below the routing threshold, force the MFMA route
            y, st = layers.gemm_linear(prod.weight, prod.bias, x0, res=res, cache=layers.DerivedCache(), row_stats=True)
        assert st is not None and st.shape == (M, C // 160, 2)
        norm = torch.nn.LayerNorm(C).to(dt).cuda()
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
            norm.bias.copy_(0.2 * torch.randn(C, generator=g))
        w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dt).cuda()
        bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda()
        post = (torch.randn(frames, C, generator=g) * 0.5).to(dt).cuda() if frames else None
        out = layers.ln_linear(norm, w, bias, y, st, layers.DerivedCache(), "t", post=post, post_div=max(pixels, 1))
        unf = layers.ln_linear(norm, w, bias, y, None, layers.DerivedCache(), "t", post=post, post_div=max(pixels, 1))
    finally:
        layers.ROUTE_MIN_TOKENS = saved
    n32 = F.layer_norm(y.float(), (C,), norm.weight.float(), norm.bias.float(), norm.eps)
    if frames:
        n32 = n32 + post.float()[(torch.arange(M, device="cuda") // pixels) % frames]
    ref = F.linear(n32, w.float(), bias.float())
    e_f, e_u = rel(out, ref), rel(unf, ref)
    assert out.shape == (M, N) and e_f < TOL[dt] and blockrel(out, ref) < 2 * TOL[dt], (e_f, e_u)
    assert e_f < 1.5 * e_u + 1e-4, (e_f, e_u)            # no worse than normalising first (one 16-bit rounding fewer)


@pytest.mark.parametrize("dt", DTYPES)
def test_linear_geglu_fused_large_magnitude_gates(dt):
    """The fused GEGLU epilogue's transcendental-free GELU (gelu_poly_pk: Phi clamped at |x| = 4.2) on gate pre-activations
    far outside the clamp: the result must tend to 0 for very negative gates -- round 3's form returned
    x * 1.3e-5, i.e. grew linearly -- and to the value itself for very positive ones."""
    M, Kd, I = 66000, 64, 128
    g = torch.Generator().manual_seed(61)
    x = q16(torch.randn(M, Kd, generator=g), dt)
    w = q16(torch.randn(2 * I, Kd, generator=g) * Kd ** -0.5, dt)
    b = torch.zeros(2 * I)
    b[I:I + 32] = -30.0           # gates around -30 +- 1
    b[I + 32:I + 64] = -6.0
    b[I + 64:I + 96] = 25.0
    b = q16(b, dt)
    h = F.linear(x, w, b)
    ref = h[:, :I] * F.gelu(h[:, I:])
    wp, bp = K.pack_geglu(w.to(dt).cuda(), b.to(dt).cuda())
    out = K.linear_geglu(x.to(dt).cuda(), wp, bp, I).float().cpu()
    assert rel(out, ref) < TOL[dt]
    # absolute error per gate band: |value| <= ~5, exact GELU(-30) = -0, GELU(-6) = -6e-9; the clamp leaves <= 5.6e-5 * |value|
    assert (out[:, :32] - ref[:, :32]).abs().max() < 5e-4, (out[:, :32] - ref[:, :32]).abs().max()
    assert (out[:, 32:64] - ref[:, 32:64]).abs().max() < 5e-4
    assert rel(out[:, 64:96], ref[:, 64:96]) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_linear_with_folded_layer_norm_rows_far_from_zero_mean(dt):
    """the folded LayerNorm takes the variance as E[x^2] - mu^2 in fp32 from per-slice (sum, sum of squares): rows whose
    mean is far larger than their spread (mean 50, std 0.5 -- about the largest ratio 16-bit storage of the rows can carry: a
    bf16 value near 50 has a spacing of 0.25) lose digits there.  Bounded against fp32 torch.layer_norm on the STORED rows."""
    from imagine360_amd import layers
    M, C, N = 66000, 320, 960
    g = torch.Generator().manual_seed(92)
    saved = layers.ROUTE_MIN_TOKENS
    layers.ROUTE_MIN_TOKENS = 0
    try:
        prod = torch.nn.Linear(C, C).to(dt).cuda()
        with torch.no_grad():
            prod.weight.mul_(0.02)
            prod.bias.zero_()
        x0 = torch.randn(M, C, generator=g).to(dt).cuda()
        res = (50.0 + 0.5 * torch.randn(M, C, generator=g)).to(dt).cuda()
        y, st = layers.gemm_linear(prod.weight, prod.bias, x0, res=res, cache=layers.DerivedCache(), row_stats=True)
        norm = torch.nn.LayerNorm(C).to(dt).cuda()
        w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dt).cuda()
        bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda()
        out = layers.ln_linear(norm, w, bias, y, st, layers.DerivedCache(), "t")
        unf = layers.ln_linear(norm, w, bias, y, None, layers.DerivedCache(), "t")
    finally:
        layers.ROUTE_MIN_TOKENS = saved
    ref = F.linear(F.layer_norm(y.float(), (C,), norm.weight.float(), norm.bias.float(), norm.eps), w.float(), bias.float())
    e_f, e_u = rel(out, ref), rel(unf, ref)
    assert float(y.float().mean()) > 45 and float(y.float().std(dim=1).mean()) < 1.0
    assert e_f < TOL[dt] and e_f < 1.5 * e_u + 5e-4, (e_f, e_u)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,C", [(70000 + 5, 320), (66000, 640)])
def test_geglu_with_folded_layer_norm(dt, M, C):
    """GEGLU(LayerNorm(y)) in one launch (im360_linear_geglu_ln) against fp32 torch and the LayerNorm + fused-GEGLU route."""
    from imagine360_amd import layers
    g = torch.Generator().manual_seed(91)
    prod = torch.nn.Linear(C, C).to(dt).cuda()
    norm = torch.nn.LayerNorm(C).to(dt).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        norm.bias.copy_(0.2 * torch.randn(C, generator=g))
    mod = layers.GEGLU(C, 4 * C).to(dt).cuda()
    saved = layers.ROUTE_MIN_TOKENS
    layers.ROUTE_MIN_TOKENS = 0             # (the token counts here are below the routing threshold of the model code)
    try:
        y, st = layers.gemm_linear(prod.weight, prod.bias, torch.randn(M, C, generator=g).to(dt).cuda(),
                                   res=(torch.randn(M, C, generator=g) + 1.0).to(dt).cuda(), cache=layers.DerivedCache(), row_stats=True)
        assert st is not None
        out = mod(y, norm, st)
        unf = mod(y, norm, None)
    finally:
        layers.ROUTE_MIN_TOKENS = saved
    h = F.linear(F.layer_norm(y.float(), (C,), norm.weight.float(), norm.bias.float(), norm.eps), mod.proj.weight.float(), mod.proj.bias.float())
    ref = h[:, :4 * C] * F.gelu(h[:, 4 * C:])
    e_f, e_u = rel(out, ref), rel(unf, ref)
    assert out.shape == (M, 4 * C) and e_f < TOL[dt] and blockrel(out, ref) < 2 * TOL[dt], (e_f, e_u)
    assert e_f < 1.5 * e_u + 1e-4, (e_f, e_u)


def test_ring_kernel_cout_groups_are_bit_identical():
    """The tile walk of the persistent GEMM kernel with the cout tiles split over XCD groups (the level-1 GEGLU projection,
    6.5 MB of weights: two groups by default) produces the same bits as the plain walk; forced 2 / 4 groups on a Linear too."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(92)
    x = torch.randn(66000, 640, generator=g).to(dt).cuda()
    w = (torch.randn(5120, 640, generator=g) * 640 ** -0.5).to(dt).cuda()
    b = (torch.randn(5120, generator=g) * 0.1).to(dt).cuda()
    wp, bp = K.pack_geglu(w, b)
    lw = K.pack_conv_weight((torch.randn(1280, 640, generator=g) * 640 ** -0.5).to(dt).cuda().reshape(1280, 640, 1, 1))
    try:
        K.tuning_set("ring_groups", 1)
        ref, lref = K.linear_geglu(x, wp, bp, 2560), K.linear(x, lw, 1280)
        for ng in (0, 2, 4):
            K.tuning_set("ring_groups", ng)
            assert torch.equal(K.linear_geglu(x, wp, bp, 2560), ref), ng
            assert torch.equal(K.linear(x, lw, 1280), lref), ng
    finally:
        K.tuning_set("ring_groups", 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C1,C2,pad,silu", [(640, 32, 32, 320, 0, 0, True), (32, 64, 128, 320, 0, 2, True), (640, 16, 16, 1280, 640, 0, True),
                                                   (3, 8, 16, 64, 64, 0, False), (5, 4, 4, 1280, 0, 0, False), (2, 6, 10, 320, 320, 2, True)])
def test_group_norm_single_launch_equals_the_three_kernel_path(dt, N, H, W, C1, C2, pad, silu):
    """im360_groupnorm_fused (statistics, per-image arrival counter, in-kernel reduction, normalisation: one launch, the
    slab's second read from the caches) produces the SAME BITS as statistics kernel + finalize + apply, at grid sizes far
    above what is resident at once (2560 workgroups: the waiting scheme's liveness), on skip pairs and with the pad-aware
    statistics; run twice (the counter is re-zeroed by every call)."""
    xa = (rnd(N, H, W, C1, seed=95) * 1.3 + 0.4).to(dt).cuda()
    xb = (rnd(N, H, W, C2, seed=96) * 0.7).to(dt).cuda() if C2 else None
    C = C1 + C2
    gamma, beta = (1 + 0.1 * rnd(C, seed=97)).to(dt).cuda(), (0.1 * rnd(C, seed=98)).to(dt).cuda()
    x = (xa, xb) if C2 else xa
    try:
        K.GN_FUSED = False
        want = K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad)
        K.GN_FUSED = True
        for _ in range(2):
            got = K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad)
            assert got.shape == (N, H, W + 2 * pad, C) and torch.equal(got, want)
    finally:
        K.GN_FUSED = False


# ------------------------------------------------------------------ round 6: normalisation straight from the partial sums
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,C1,C2,pad,silu", [(3, 8, 16, 64, 0, 0, True), (2, 16, 32, 320, 0, 2, True), (5, 4, 4, 1280, 0, 0, False), (1, 64, 128, 320, 0, 2, True),
                                                   (2, 16, 32, 640, 320, 2, True), (4, 8, 8, 1280, 640, 0, False), (2, 6, 10, 320, 320, 2, True), (40, 32, 32, 320, 0, 0, True)])
def test_group_norm_from_partial_sums_equals_the_three_launch_path(dt, N, H, W, C1, C2, pad, silu):
    """im360_groupnorm_apply_partials (every workgroup rebuilds its image's scale / shift from the partial sums, then normalises:
    ONE launch behind the statistics) == statistics + finalize + apply, bit for bit on one tensor (the same reduction order), to
    the fp64 order of a group that straddles the two tensors of a pair (30 channels per group at 960); and against fp32 torch."""
    xa = q16(rnd(N, H, W, C1, seed=160) + 0.3, dt).to(dt).cuda()
    xb = q16(rnd(N, H, W, C2, seed=161) * 1.5, dt).to(dt).cuda() if C2 else None
    C = C1 + C2
    gamma, beta = (1 + 0.1 * rnd(C, seed=162)).to(dt).cuda(), (0.1 * rnd(C, seed=163)).to(dt).cuda()
    x = xa if xb is None else (xa, xb)
    assert K.GN_MODE == "partials"
    got = K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad)
    try:
        K.GN_MODE = "three"
        want = K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad)
    finally:
        K.GN_MODE = "partials"
    assert got.shape == want.shape == (N, H, W + 2 * pad, C)
    if xb is None:
        assert torch.equal(got, want)
    else:
        assert rel(got, want) < 1e-6 and (got.float() - want.float()).abs().max() <= 2 * float(want.float().abs().max()) * 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    cat = xa if xb is None else torch.cat([xa, xb], dim=-1)
    ref = F.group_norm(OG.pad_pano(cat.float().cpu().permute(0, 3, 1, 2), pad), 32, gamma.float().cpu(), beta.float().cpu(), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
    assert rel(got, ref) < TOL[dt]
    for v in (0, 1, 3, 6):                  # the A/B variants of the apply loop (plain stores, loads in flight, non-temporal loads): the same bits
        try:
            K.tuning_set("gn_apply", v)
            assert torch.equal(K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad), got)
        finally:
            K.tuning_set("gn_apply", 2)


@pytest.mark.parametrize("dt", DTYPES)
def test_group_norm_from_the_producers_partial_sums_in_one_launch(dt):
    """The tagged output of conv2d(..., gn_stats=True) normalised by ONE launch (no statistics pass, no finalize) == the
    finalize + apply pair on the same partial sums, bit for bit; a pair of one tagged and one untagged tensor likewise to fp64 order."""
    g = torch.Generator().manual_seed(164)
    N, H, W, Cin, Cout = 160, 32, 32, 320, 320
    x = torch.randn(N, H, W, Cin, generator=g).to(dt).cuda()
    wp = K.pack_conv_weight((torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dt).cuda())
    y = K.conv2d(x, wp, Cout, gn_stats=True)
    assert K._gn_of(y) is not None
    gamma, beta = (1 + 0.1 * torch.randn(Cout, generator=g)).to(dt).cuda(), (0.1 * torch.randn(Cout, generator=g)).to(dt).cuda()
    K.STATS = {}
    try:
        got = K.group_norm(y, gamma, beta, 32, 1e-5, silu=True)
        assert "gn_stats" not in K.STATS and K.STATS["gn_apply"][2] == 1          # one launch, no statistics launch
    finally:
        K.STATS = None
    try:
        K.GN_MODE = "three"
        want = K.group_norm(y, gamma, beta, 32, 1e-5, silu=True)
    finally:
        K.GN_MODE = "partials"
    assert torch.equal(got, want)
    skip = torch.randn(N, H, W, 640, generator=g).to(dt).cuda()
    g2, b2 = (1 + 0.1 * torch.randn(960, generator=g)).to(dt).cuda(), (0.1 * torch.randn(960, generator=g)).to(dt).cuda()
    got = K.group_norm((y, skip), g2, b2, 32, 1e-5, silu=True)
    ref = F.silu(F.group_norm(torch.cat([y, skip], dim=-1).float().cpu().permute(0, 3, 1, 2), 32, g2.float().cpu(), b2.float().cpu(), 1e-5)).permute(0, 2, 3, 1)
    assert rel(got, ref) < TOL[dt]


# ------------------------------------------------------------------ round 4: GroupNorm statistics from the producer's epilogue
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,taps", [(160, 32, 32, 320, 320, 9), (80, 32, 32, 320, 640, 9), (640, 16, 16, 640, 640, 1), (16, 64, 128, 320, 320, 9)])
def test_group_norm_statistics_from_the_conv_epilogue(dt, N, H, W, Cin, Cout, taps):
    """conv2d(..., gn_stats=True): the 256 x 320 tile's epilogue writes per tile and channel (sum, sum of squares) of the values it
    stores (bias, time embedding and residual included); group_norm_stats() of the tagged tensor then runs only the finalize.
    Against the statistics pass over the same tensor (fp32 summation order apart) and fp32 torch; and the tag dies with an
    in-place write."""
    g = torch.Generator().manual_seed(120)
    x = torch.randn(N, H, W, Cin, generator=g).to(dt).cuda()
    w = (torch.randn(Cout, Cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) * (taps * Cin) ** -0.5).to(dt).cuda()
    wp = K.pack_conv_weight(w)
    b = (torch.randn(Cout, generator=g) * 0.3 + 0.5).to(dt).cuda()
    temb = torch.randn(N // 8, Cout, generator=g).to(dt).cuda()
    res = torch.randn(N, H, W, Cout, generator=g).to(dt).cuda()
    y = K.conv2d(x, wp, Cout, bias=b, temb=temb, imgs_per_temb=8, res=res, gn_stats=True)
    assert K._gn_of(y) is not None and K._gn_of(y)[1] == H * W // 256
    assert torch.equal(y, K.conv2d(x, wp, Cout, bias=b, temb=temb, imgs_per_temb=8, res=res))        # same output bits as the plain epilogue
    gamma, beta = (1 + 0.1 * torch.randn(Cout, generator=g)).to(dt).cuda(), (0.1 * torch.randn(Cout, generator=g)).to(dt).cuda()
    s1, h1 = K.group_norm_stats(y, gamma, beta, 32, 1e-5)
    try:
        K.GN_FROM_PRODUCER = False
        s2, h2 = K.group_norm_stats(y, gamma, beta, 32, 1e-5)
    finally:
        K.GN_FROM_PRODUCER = True
    assert rel(s1, s2) < 1e-5 and (h1 - h2).abs().max() < 1e-4 * (1 + float(h2.abs().max()))
    ref = F.silu(F.group_norm(y.float().cpu().permute(0, 3, 1, 2), 32, gamma.float().cpu(), beta.float().cpu(), 1e-5)).permute(0, 2, 3, 1)
    assert rel(K.group_norm(y, gamma, beta, 32, 1e-5, silu=True), ref) < TOL[dt]
    y.add_(1.0)                       # an in-place write invalidates the tag
    assert K._gn_of(y) is None


@pytest.mark.parametrize("dt", DTYPES)
def test_group_norm_statistics_from_the_linear_epilogue_and_skip_pairs(dt):
    """linear(..., gn_hw = pixels per image) on the persistent ring kernel (with and without the LayerNorm row statistics), and
    the decoder's skip pairs: (tagged, untagged), (tagged, tagged) and mixed slab counts, each against the statistics pass."""
    g = torch.Generator().manual_seed(121)
    N, HW, C = 80, 1024, 320
    x = torch.randn(N * HW, C, generator=g).to(dt).cuda()
    wl = (torch.randn(C, C, generator=g) * C ** -0.5).to(dt).cuda()
    wp = K.pack_conv_weight(wl.reshape(C, C, 1, 1))
    b = torch.randn(C, generator=g).to(dt).cuda()
    res = (torch.randn(N * HW, C, generator=g) + 0.7).to(dt).cuda()
    y = K.linear(x, wp, C, bias=b, res=res, gn_hw=HW)
    assert K._gn_of(y) is not None and torch.equal(y, K.linear(x, wp, C, bias=b, res=res))
    y5, st = K.linear(x, wp, C, bias=b, res=res, row_stats=True, gn_hw=HW)
    assert K._gn_of(y5) is not None and torch.equal(y5, y) and st is not None
    gamma, beta = (1 + 0.1 * torch.randn(2 * C, generator=g)).to(dt).cuda(), (0.1 * torch.randn(2 * C, generator=g)).to(dt).cuda()
    ya = K.carry_gn(y.reshape(N, 32, 32, C), y)
    yb5 = K.carry_gn(y5.reshape(N, 32, 32, C), y5)
    plain = (torch.randn(N, 32, 32, C, generator=g) * 2).to(dt).cuda()
    assert K._gn_of(ya) is not None and K._gn_of(plain) is None

    def both(xx, ga, be):
        a = K.group_norm_stats(xx, ga, be, 32, 1e-5)
        try:
            K.GN_FROM_PRODUCER = False
            r = K.group_norm_stats(xx, ga, be, 32, 1e-5)
        finally:
            K.GN_FROM_PRODUCER = True
        assert rel(a[0], r[0]) < 1e-5 and (a[1] - r[1]).abs().max() < 1e-4 * (1 + float(r[1].abs().max())), (rel(a[0], r[0]), (a[1] - r[1]).abs().max())
    both(ya, gamma[:C], beta[:C])
    both((ya, plain), gamma, beta)          # 640 channels in 32 groups of 20: no group straddles the two sources
    both((plain, yb5), gamma, beta)
    both((ya, yb5), gamma, beta)
    # groups that straddle the sources: 320 + 640 channels = 32 groups of 30
    wide = torch.randn(N, 32, 32, 2 * C, generator=g).to(dt).cuda()
    g3, b3 = (1 + 0.1 * torch.randn(3 * C, generator=g)).to(dt).cuda(), (0.1 * torch.randn(3 * C, generator=g)).to(dt).cuda()
    both((ya, wide), g3, b3)
    both((wide, ya), g3, b3)
