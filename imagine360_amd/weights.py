"""Deterministic synthetic weights ("filler") for the random-init architecture.

No checkpoint of the reference is available offline, so parity tests
and the benchmark use formula-based weights that are bit-identical on every device and do
not depend on ``torch.manual_seed``: element ``i`` of the parameter named ``n`` is an integer
hash of ``(crc32(n), i)`` mapped to [-1, 1) and scaled by the layer's fan-in.  Crucially the
filler also overwrites the reference's zero-initialised layers (WarpAttn ``to_out`` /
``ff.net.2``, motion-module ``proj_out``, ``fps_embedding.linear_2`` --
src/modules/transformer.py:30-32,55-57; animatediff/models/motion_module.py:88-89;
animatediff/models/unet.py:168-169), which would otherwise make those paths identities.
"""
import math
import zlib

import torch

_M32 = 0xFFFFFFFF


def _hash_uniform(seed: int, numel: int, device) -> torch.Tensor:
    """lowbias32-style integer hash of (seed + index) -> float32 in [-1, 1); exact int64 math."""
    x = torch.arange(numel, dtype=torch.int64, device=device) + (seed & _M32)
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    # 24 mantissa-exact bits
    return ((x >> 8).to(torch.float32) * (1.0 / 8388608.0)) - 1.0


def filler_tensor(name: str, shape, device="cpu", gain: float = 1.0) -> torch.Tensor:
    shape = tuple(shape)
    numel = 1
    for s in shape:
        numel *= s
    u = _hash_uniform(zlib.crc32(name.encode()), numel, device).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "bias":
        return u * 0.05
    if len(shape) == 1:                       # norm scales
        return 1.0 + 0.1 * u
    if leaf == "latents":                     # Resampler queries [1, n, dim]
        return u * (shape[-1] ** -0.5) * math.sqrt(3.0)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return u * (gain * math.sqrt(3.0 / max(fan_in, 1)))


@torch.no_grad()
def fill_state_dict_(sd, gain: float = 1.0):
    """Overwrite every floating-point *parameter-like* entry of ``sd`` in place.  Buffers that
    carry fixed tables (``pe``, ``freq_bands``) are left untouched."""
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("pe", "freq_bands"):
            continue
        v = filler_tensor(name, t.shape, t.device, gain)
        t.copy_(v.to(t.dtype))
    return sd


@torch.no_grad()
def fill_module_(module: torch.nn.Module, gain: float = 1.0):
    fill_state_dict_(module.state_dict(), gain)
    return module
