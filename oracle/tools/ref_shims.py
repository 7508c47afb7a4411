"""Import harness for the *reference* (3DTopia/Imagine360 at /root/reference).

TEST INFRASTRUCTURE ONLY.  Used in the authoring container (where /root/reference is
mounted) to (i) check oracle/im360_oracle against the real reference and (ii) emit the
golden fixtures under tests/golden/.  Nothing here ships to, or runs on, the GPU box;
no reference source is copied -- this file only installs stand-in *modules* for
third-party packages the reference imports but that are absent from this image
(SURVEY.md section 8c lists them), then imports the reference from where it lies.

Semantic stand-ins (arithmetic that lives in un-vendored third parties; "parity
unpinned" by the reference itself):
  xformers.ops.memory_efficient_attention(q,k,v,attn_bias,scale) = softmax(q k^T s + b) v
  kornia.geometry.transform.remap(align_corners=True)            = grid_sample, zeros pad
  kornia.filters.gaussian_blur2d((5,5),(1,1),'replicate')         = separable normalised
  kornia.utils.create_meshgrid(h,w,False)                         = [1,h,w,2] (x,y)
  cv2.Rodrigues(v)                                                = I + sin K + (1-cos) K^2
"""
import importlib.machinery
import importlib.metadata
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _rodrigues(v):
    v = np.asarray(v, dtype=np.float64).reshape(3)
    th = float(np.linalg.norm(v))
    if th < 1e-15:
        return np.eye(3), None
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K), None


def _remap(img, map_x, map_y, mode="bilinear", padding_mode="zeros", align_corners=True,
           normalized_coordinates=False):
    h, w = img.shape[-2:]
    gx = 2.0 * map_x / (w - 1) - 1.0
    gy = 2.0 * map_y / (h - 1) - 1.0
    grid = torch.stack([gx, gy], dim=-1).to(img.dtype)
    return F.grid_sample(img, grid, mode=mode, padding_mode=padding_mode, align_corners=True)


def _gauss1d(k, s):
    x = torch.arange(k, dtype=torch.float64) - (k - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2.0 * s * s))
    return g / g.sum()


def _gaussian_blur2d(x, kernel_size, sigma, border_type="reflect", separable=True):
    ky, kx = kernel_size
    sy, sx = sigma
    gy = _gauss1d(ky, sy).to(x.dtype).to(x.device)
    gx = _gauss1d(kx, sx).to(x.dtype).to(x.device)
    c = x.shape[1]
    mode = {"replicate": "replicate", "reflect": "reflect", "constant": "constant"}[border_type]
    xp = F.pad(x, (kx // 2, kx // 2, ky // 2, ky // 2), mode=mode)
    xp = F.conv2d(xp, gx.view(1, 1, 1, kx).repeat(c, 1, 1, 1), groups=c)
    xp = F.conv2d(xp, gy.view(1, 1, ky, 1).repeat(c, 1, 1, 1), groups=c)
    return xp


def _create_meshgrid(h, w, normalized_coordinates=True, device=None, dtype=None):
    xs = torch.linspace(0, w - 1, w, device=device, dtype=torch.float32)
    ys = torch.linspace(0, h - 1, h, device=device, dtype=torch.float32)
    if normalized_coordinates:
        xs = (xs / (w - 1) - 0.5) * 2
        ys = (ys / (h - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    g = torch.stack([gx, gy], dim=-1).unsqueeze(0)
    return g.to(dtype) if dtype is not None else g


def _mea(q, k, v, attn_bias=None, p=0.0, scale=None, op=None):
    if scale is None:
        scale = q.shape[-1] ** -0.5

    def one(q, k, v, b):
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
        if b is not None:
            s = s + b.float()
        return torch.matmul(s.softmax(-1), v.float()).to(q.dtype)
    # (batch * head) chunks that keep the score matrix under ~2 GB: the cfg2-sized fixtures (8192^2 self-attention,
    # 8192 x 20480 cross-view attention) would otherwise need 43 / 215 GB.  Batch entries are independent: same numbers.
    n = q.shape[0]
    per = max(1, int(2e9 // (4 * q.shape[-2] * k.shape[-2])))
    if q.dim() != 3 or per >= n:
        return one(q, k, v, attn_bias)
    out = torch.empty(q.shape[:-1] + (v.shape[-1],), dtype=q.dtype)
    for i in range(0, n, per):
        b = None if attn_bias is None else (attn_bias[i:i + per] if attn_bias.dim() == 3 and attn_bias.shape[0] == n else attn_bias)
        out[i:i + per] = one(q[i:i + per], k[i:i + per], v[i:i + per], b)
    return out


class _SamStub:
    """SamPredictor stand-in: hands out caller-provided features (``preset`` [N,256,64,64]) in
    order, n per ``set_torch_image`` call, so the harness controls the SAM features exactly."""

    class _T:
        def apply_image(self, image):
            return image

    preset = None

    def __init__(self, model=None):
        self.transform = self._T()
        self._n = 0
        self._cursor = 0

    def set_torch_image(self, x, hw):
        self._n = x.shape[0]

    def get_image_embedding(self):
        out = type(self).preset[self._cursor:self._cursor + self._n]
        self._cursor += self._n
        return out


_installed = False


def install():
    """Install shims and put /root/reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    _installed = True
    import huggingface_hub
    import huggingface_hub.constants as hc
    if not hasattr(hc, "hf_cache_home"):
        hc.hf_cache_home = "/tmp/hf_cache"
    if not hasattr(huggingface_hub, "HfFolder"):
        class HfFolder:  # noqa
            @staticmethod
            def get_token():
                return None
        huggingface_hub.HfFolder = HfFolder
    if not hasattr(huggingface_hub, "cached_download"):
        huggingface_hub.cached_download = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))

    # real transformers classes must be resolved before torchvision is faked
    from transformers import (CLIPTextModel, CLIPTokenizer, CLIPImageProcessor,  # noqa
                              CLIPVisionModelWithProjection)

    _orig_version = importlib.metadata.version

    def _hide_tf(name):
        if name == "transformers":
            raise importlib.metadata.PackageNotFoundError(name)
        return _orig_version(name)

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    tv.utils = _mod("torchvision.utils", save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)
    cv2 = _mod("cv2", Rodrigues=_rodrigues, INTER_LINEAR=1, INTER_CUBIC=2, INTER_NEAREST=0,
               BORDER_WRAP=3, remap=None)
    kor = _mod("kornia")
    kor.utils = _mod("kornia.utils", create_meshgrid=_create_meshgrid)
    kor.filters = _mod("kornia.filters", gaussian_blur2d=_gaussian_blur2d)
    kor.geometry = _mod("kornia.geometry")
    kor.geometry.transform = _mod("kornia.geometry.transform", remap=_remap)
    xf = _mod("xformers")
    xf.ops = _mod("xformers.ops", memory_efficient_attention=_mea)
    fs = _mod("fairscale")
    fs.nn = _mod("fairscale.nn")
    fs.nn.checkpoint = _mod("fairscale.nn.checkpoint", checkpoint_wrapper=lambda m, *a, **k: m)
    _mod("imageio")
    _mod("decord", VideoReader=None)
    _mod("loguru", logger=None)
    _mod("segment_anything", SamPredictor=_SamStub, sam_model_registry={})

    sys.path.insert(0, REF_ROOT)
    importlib.metadata.version = _hide_tf
    try:
        import diffusers  # noqa  (the vendored one under /root/reference)
    finally:
        importlib.metadata.version = _orig_version

    # the vendored diffusers probes xformers through package metadata (absent for the stand-in):
    # hand it the stand-in module so `_memory_efficient_attention_xformers` can be exercised
    import diffusers.models.attention_processor as _ap
    _ap.xformers = xf

    import src.models.MVGenModel as mvm
    import animatediff.pipelines.pipeline_animation_inference_dual as pip
    mvm.flush = lambda: None
    pip.flush = lambda: None


def ref_modules():
    """Return the reference classes the goldens are generated from."""
    install()
    from animatediff.models.unet import UNet3DConditionModel
    from src.models.MVGenModel import MultiViewBaseModel
    from animatediff.pipelines.pipeline_animation_inference_dual import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    return dict(UNet3DConditionModel=UNet3DConditionModel, MultiViewBaseModel=MultiViewBaseModel,
                AnimationPipeline=AnimationPipeline, AutoencoderKL=AutoencoderKL,
                DDIMScheduler=DDIMScheduler)
