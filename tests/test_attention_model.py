"""CPU model of attn_fwd.hip's per-wave algorithm (numbers, not lanes): Q pre-multiplied by scale*log2(e) and rounded
to 16 bits, the running max entering the QK^T accumulation as the MFMA C operand, ONE max / rescale decision per
64-key tile taken wave-wide (`__any`) with threshold RESCALE_THR, the first tile always setting the max, exp2 of the
accumulator, P rounded to 16 bits for the PV product but summed unrounded.  Checked against exact softmax attention
on the cases that stress the bookkeeping: late spikes (rescale path), strongly negative logits (first-tile path with
a negative max), ragged key counts, the additive bias.  The HIP kernel itself is checked on the GPU; this pins the
algorithm its comments describe."""
import math

import pytest
import torch

LOG2E = 1.4426950408889634
THR = 5.0


def r16(x, dt):
    return x.to(dt).float()


def model_attention(q, k, v, scale, bias=None, dt=torch.bfloat16, out_scale=1.0):
    Nq, D = q.shape
    Nk = k.shape[0]
    qs = r16(q * (scale * LOG2E), dt)
    out = torch.empty(Nq, D)
    stats = {"rescales": 0, "tiles": 0}
    for q0 in range(0, Nq, 32):
        qb = qs[q0:q0 + 32]
        n = qb.shape[0]
        m_sc = torch.zeros(n)
        l = torch.zeros(n)
        o = torch.zeros(n, D)
        for t, k0 in enumerate(range(0, Nk, 64)):
            kt, vt = k[k0:k0 + 64], v[k0:k0 + 64]
            s = qb @ kt.T - m_sc[:, None]                       # the MFMA result: C operand = -max
            if bias is not None:
                s = s + bias[q0:q0 + 32, k0:k0 + 64] * LOG2E
            mloc = s.max(dim=1).values
            first = t == 0
            if first or bool((mloc > THR).any()):               # one wave-wide decision per tile
                delta = mloc if first else mloc.clamp_min(0.0)
                alpha = torch.ones(n) if first else torch.exp2(-delta)
                m_sc = m_sc + delta
                l, o = l * alpha, o * alpha[:, None]
                s = s - delta[:, None]
                stats["rescales"] += 0 if first else 1
            p = torch.exp2(s)
            l = l + p.sum(dim=1)
            o = o + r16(p, dt) @ vt
            stats["tiles"] += 1
        out[q0:q0 + 32] = o / l[:, None] * out_scale
    return out, stats


def exact(q, k, v, scale, bias=None):
    s = (q.double() @ k.double().T) * scale
    if bias is not None:
        s = s + bias.double()
    return (torch.softmax(s, dim=-1) @ v.double()).float()


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1.5e-3)])
@pytest.mark.parametrize("Nq,Nk,D", [(64, 512, 64), (40, 77, 64), (96, 130, 32)])
def test_model_matches_exact_attention(dt, tol, Nq, Nk, D):
    g = torch.Generator().manual_seed(Nq + Nk)
    q, k, v = (r16(torch.randn(n, D, generator=g), dt) for n in (Nq, Nk, Nk))
    out, _ = model_attention(q, k, v, D ** -0.5, dt=dt)
    assert rel(out, exact(q, k, v, D ** -0.5)) < tol
    bias = r16(torch.rand(Nq, Nk, generator=g) * 2 - 1, dt)
    out, _ = model_attention(q, k, v, D ** -0.5, bias=bias, dt=dt)
    assert rel(out, exact(q, k, v, D ** -0.5, bias=bias)) < tol


def test_late_spikes_take_the_rescale_path():
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16
    q, k, v = (r16(torch.randn(n, 64, generator=g), dt) for n in (64, 512, 512))
    k[300] = q[5] * 6.0
    k[450] = q[9] * 9.0
    out, stats = model_attention(q, k, v, 64 ** -0.5, dt=dt)
    assert stats["rescales"] >= 2 and stats["rescales"] < stats["tiles"] // 2      # rare, as designed
    ref = exact(q, k, v, 64 ** -0.5)
    assert rel(out, ref) < 6e-3 and (out - ref)[[5, 9]].abs().max() < 0.05


def test_negative_logits_and_unit_scale():
    """All logits far below zero (the running max must come from the first tile, not from its initial 0), and the
    IPCrossAttention quirk's logit scale 1.0 with larger magnitudes."""
    g = torch.Generator().manual_seed(6)
    dt = torch.bfloat16
    q = r16(torch.randn(32, 64, generator=g).abs() + 1.0, dt)
    k = r16(-(torch.randn(200, 64, generator=g).abs() + 1.0), dt)          # q . k ~ -150
    v = r16(torch.randn(200, 64, generator=g), dt)
    out, _ = model_attention(q, k, v, 1.0, dt=dt)
    assert torch.isfinite(out).all() and rel(out, exact(q, k, v, 1.0)) < 2e-2
    q2, k2 = r16(torch.randn(64, 64, generator=g) * 0.3, dt), r16(torch.randn(141, 64, generator=g), dt)
    v2 = r16(torch.randn(141, 64, generator=g), dt)
    out, _ = model_attention(q2, k2, v2, 1.0, dt=dt)
    assert rel(out, exact(q2, k2, v2, 1.0)) < 8e-3


def test_threshold_bounds_p():
    """Between rescales P never exceeds 2^THR (so the 16-bit P and the fp32 sums stay well scaled)."""
    g = torch.Generator().manual_seed(7)
    q, k = torch.randn(32, 64, generator=g), torch.randn(4096, 64, generator=g)
    qs = q * (64 ** -0.5 * LOG2E)
    m = (qs @ k[:64].T).max(dim=1).values
    for k0 in range(64, 4096, 64):
        s = qs @ k[k0:k0 + 64].T - m[:, None]
        mloc = s.max(dim=1).values
        if bool((mloc > THR).any()):
            m = m + mloc.clamp_min(0.0)
            s = s - mloc.clamp_min(0.0)[:, None]
        assert float(torch.exp2(s).max()) <= 2.0 ** THR * (1 + 1e-6)
    assert math.isfinite(float(m.max()))
