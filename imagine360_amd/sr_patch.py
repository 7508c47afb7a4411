"""The 360-degree close-loop patch of the super-resolution stage on the MI355X.

The reference patches VEnhancer (an external model, not in the checkout) in three places so that the left and right
image borders of the equirectangular video stay continuous (sr/video_to_video_model.py):

  :16-29    ``padding_pano`` / ``unpadding_pano``: ``pad_pano`` / ``unpad_pano`` of src/utils/pano.py:75-101 with the
            padding given in latent columns (x 8 in pixel space)
  :99       ``F.pad(video_data, (w1, w2, h1, h2), "circular")`` instead of VEnhancer's constant pad_to_fit
  :160-162  latent -> ``padding_pano(latent=True)`` -> tiled decode -> ``unpadding_pano``

All three are one circular gather over W-last (NCHW / NCFHW) tensors; here they are one launch of
``im360_circular_pad_hw`` (HBM-bound: every output byte is read once and written once) instead of
rearrange -> F.pad -> rearrange.  Same names, argument meaning and errors as the reference helpers; a maintainer swaps
the two helper definitions for ``from imagine360_amd.sr_patch import padding_pano, unpadding_pano`` and line 99 for
``circular_pad(video_data, (w1, w2, h1, h2))``.
"""
from . import kernels


def circular_pad(x, pad):
    """``torch.nn.functional.pad(x, pad, mode="circular")`` for ``pad = (left, right)`` or ``(left, right, top, bottom)``
    on the last axes of a device tensor."""
    if len(pad) == 2:
        left, right, top, bottom = pad[0], pad[1], 0, 0
    elif len(pad) == 4:
        left, right, top, bottom = pad
    else:
        raise NotImplementedError("circular_pad: pad must have 2 or 4 entries (last one or two axes)")
    if min(left, right, top, bottom) < 0:
        raise NotImplementedError("circular_pad: negative (cropping) pads are not supported")
    if left == right == top == bottom == 0:
        return x
    return kernels.circular_pad_hw(x.contiguous(), int(left), int(right), int(top), int(bottom))


def pad_pano(pano, padding):
    """src/utils/pano.py:75-95 for W-last tensors (4 or 5 dims, as the reference accepts)."""
    if padding <= 0:
        return pano
    if pano.ndim not in (4, 5):
        raise NotImplementedError('pano should be 4 or 5 dim')
    return circular_pad(pano, (padding, padding))


def unpad_pano(pano_pad, padding):
    """src/utils/pano.py:98-101 (a view, like the reference's slice)."""
    return pano_pad if padding <= 0 else pano_pad[..., padding:-padding]


def padding_pano(pano, padding=16, latent=False):
    """sr/video_to_video_model.py:21-24: ``padding`` counts latent columns; pixel-space tensors get 8x as many."""
    if not latent:
        padding *= 8
    return pad_pano(pano, padding=padding)


def unpadding_pano(pano_pad, padding=16, latent=False):
    """sr/video_to_video_model.py:26-29."""
    if not latent:
        padding *= 8
    return unpad_pano(pano_pad, padding=padding)
