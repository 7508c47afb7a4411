"""Host-preprocessing geometry of the inference script on the MI355X.

The reference warps every frame on the CPU, one ``cv2.remap`` per (frame, view), with the sampling maps rebuilt for every
call although they depend on the camera only:

  ``process_equi``      inference_dual_p2e.py:113-144   panorama frames -> 20 perspective views   (Equirec2Perspec.py:18-62)
  ``pers2pano_vid``     inference_dual_p2e.py:291-304   input frames -> equirectangular canvas + mask (Perspec2Equirec.py:27-72)
  ``get_anchor_target`` animatediff/utils/video_mask.py:158-217   anchor crops, masks, relative positions, pitches
  ``get_maxrec_cord``   src/modules/utils.py:39-73      largest rectangle of a mask (pure-Python loops)

Here the maps are built once per camera on the host in the reference's own float64 numpy arithmetic (pinned bit-exactly on
the real modules, tests/golden/preproc.npz), cached, and ONE launch of ``im360_remap_cubic_wrap_u8`` warps all frames through
all maps on the GPU; the rectangle search is the C function ``im360_max_rect``.  Same names, arguments and return
conventions as the reference functions.  The bicubic arithmetic of cv2.remap is restated from OpenCV's published source
(fixed-point, 1/32-pixel quantisation): OpenCV is not available here to check it -- parity unpinned for that one function.
"""
import functools
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import kernels


# ------------------------------------------------------------------------------------------------ maps (host, float64)
def _axis_angle(v):
    """cv2.Rodrigues (vector -> matrix): I + sin(t) K + (1 - cos t) K^2."""
    v = np.asarray(v, np.float64).reshape(3)
    t = float(np.linalg.norm(v))
    if t < 1e-15:
        return np.eye(3)
    k = v / t
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * (K @ K)


def _yaw_pitch(theta, phi):
    """R1 = yaw about z, R2 = pitch about the yawed y axis (Equirec2Perspec.py:42-45)."""
    R1 = _axis_angle(np.array([0.0, 0.0, 1.0], np.float32) * np.radians(theta))
    R2 = _axis_angle(np.dot(R1, np.array([0.0, 1.0, 0.0], np.float32)) * np.radians(-phi))
    return R1, R2


@functools.lru_cache(maxsize=256)
def e2p_maps(fov, theta, phi, height, width, equ_h, equ_w):
    """Source coordinates (lon, lat), float32 [height, width], of the perspective view (fov, theta, phi) in an
    equ_h x equ_w panorama: what Equirectangular.GetPerspective hands to cv2.remap (Equirec2Perspec.py:18-58)."""
    cx, cy = (equ_w - 1) / 2.0, (equ_h - 1) / 2.0
    h_fov = float(height) / width * fov
    w_len, h_len = np.tan(np.radians(fov / 2.0)), np.tan(np.radians(h_fov / 2.0))
    xs = np.ones([height, width], np.float32)
    ys = np.tile(np.linspace(-w_len, w_len, width), [height, 1])
    zs = -np.tile(np.linspace(-h_len, h_len, height), [width, 1]).T
    norm = np.sqrt(xs ** 2 + ys ** 2 + zs ** 2)
    rays = np.stack((xs, ys, zs), axis=2) / np.repeat(norm[:, :, np.newaxis], 3, axis=2)
    R1, R2 = _yaw_pitch(theta, phi)
    rays = np.dot(R2, np.dot(R1, rays.reshape([height * width, 3]).T)).T
    lat = -np.arcsin(rays[:, 2]).reshape([height, width]) / np.pi * 180
    lon = np.arctan2(rays[:, 1], rays[:, 0]).reshape([height, width]) / np.pi * 180
    lon = lon / 180 * cx + cx
    lat = lat / 90 * cy + cy
    return lon.astype(np.float32), lat.astype(np.float32)


@functools.lru_cache(maxsize=256)
def p2e_maps(fov, theta, phi, pers_h, pers_w, height, width):
    """(lon_map, lat_map float32, mask int) [height, width] of Perspective.GetEquirec (Perspec2Equirec.py:27-72): where every
    panorama pixel samples the perspective image, and which pixels it covers."""
    h_fov = float(pers_h) / pers_w * fov
    w_len, h_len = np.tan(np.radians(fov / 2.0)), np.tan(np.radians(h_fov / 2.0))
    lon, lat = np.meshgrid(np.linspace(-180, 180, width), np.linspace(90, -90, height))
    rays = np.stack((np.cos(np.radians(lon)) * np.cos(np.radians(lat)), np.sin(np.radians(lon)) * np.cos(np.radians(lat)),
                     np.sin(np.radians(lat))), axis=2)
    R1, R2 = _yaw_pitch(theta, phi)
    rays = np.dot(np.linalg.inv(R1), np.dot(np.linalg.inv(R2), rays.reshape([height * width, 3]).T)).T.reshape([height, width, 3])
    front = np.where(rays[:, :, 0] > 0, 1, 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        rays[:, :] = rays[:, :] / np.repeat(rays[:, :, 0][:, :, np.newaxis], 3, axis=2)
    inside = (-w_len < rays[:, :, 1]) & (rays[:, :, 1] < w_len) & (-h_len < rays[:, :, 2]) & (rays[:, :, 2] < h_len)
    lon_map = np.where(inside, (rays[:, :, 1] + w_len) / 2 / w_len * pers_w, 0)
    lat_map = np.where(inside, (-rays[:, :, 2] + h_len) / 2 / h_len * pers_h, 0)
    return lon_map.astype(np.float32), lat_map.astype(np.float32), np.where(inside, 1, 0) * front


# ------------------------------------------------------------------------------------------------ cv2.remap on the GPU
@functools.lru_cache(maxsize=1)
def cubic_weight_table():
    """int16 [1024, 16]: OpenCV's fixed-point bicubic table (A = -0.75, 32 x 32 fractional positions, weights x 2^15, the
    rounding residue of every entry folded into the smallest / largest of its taps (2..3, 2..3) so the entry sums to 2^15 --
    the centre tap of an integer position saturates at 32767 and tap (2, 2) takes the missing 1)."""
    a = np.float32(-0.75)
    x = (np.arange(32, dtype=np.float32) * np.float32(1.0 / 32))
    one = np.float32(1)
    c0 = ((a * (x + one) - np.float32(5) * a) * (x + one) + np.float32(8) * a) * (x + one) - np.float32(4) * a
    c1 = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + one
    c2 = ((a + np.float32(2)) * (one - x) - (a + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    t1 = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)                        # [32, 4]
    v = (t1[:, None, :, None] * t1[None, :, None, :]).astype(np.float32)              # [fy, fx, k1, k2]
    it = np.clip(np.rint(v * np.float32(32768)), -32768, 32767).astype(np.int64).reshape(1024, 4, 4)
    diff = it.reshape(1024, 16).sum(1) - 32768
    taps = it[:, 2:4, 2:4].reshape(1024, 4)                                            # scan order (2,2) (2,3) (3,2) (3,3)
    for e in np.nonzero(diff)[0]:
        lo = hi = 0
        for k in range(1, 4):
            if taps[e, k] < taps[e, lo]:
                lo = k
            elif taps[e, k] > taps[e, hi]:
                hi = k
        k = hi if diff[e] < 0 else lo
        it[e, 2 + k // 2, 2 + k % 2] -= diff[e]
    return it.reshape(1024, 16).astype(np.int16)


_dev_cache = {}


def _on_device(key, build, device):
    k = (key, str(device))
    if k not in _dev_cache:
        _dev_cache[k] = build().to(device)
    return _dev_cache[k]


def remap(img, map_x, map_y):
    """``cv2.remap(img, map_x, map_y, INTER_CUBIC, borderMode=BORDER_WRAP)`` batched: img uint8 [N, H, W, C] (device
    tensor), maps float32 [M, h, w] (device tensors) -> uint8 [N, M, h, w, C]."""
    wtab = _on_device("cubic_table", lambda: torch.from_numpy(cubic_weight_table()), img.device)
    return kernels.remap_cubic_wrap(img.contiguous(), map_x.contiguous(), map_y.contiguous(), wtab)


def _as_u8_frames(img, device):
    t = torch.as_tensor(img)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    assert t.dtype == torch.uint8 and t.dim() == 4, "uint8 images [H, W, C] or [N, H, W, C]"
    return t.to(device)


class Equirectangular:
    """Equirec2Perspec.Equirectangular for an in-memory uint8 panorama [H, W, C] or a stack [N, H, W, C]."""

    def __init__(self, img, text2light=False, device="cuda"):
        if isinstance(img, str):
            raise NotImplementedError("file input is the reference's cv2.imread path; pass the decoded uint8 array")
        self._img = _as_u8_frames(img, device)
        if text2light:                                                   # Equirec2Perspec.py:12-13
            self._img = torch.roll(self._img, -60, dims=1)
        self._single = torch.as_tensor(img).dim() == 3
        _, self._height, self._width, _ = self._img.shape

    def GetPerspective(self, FOV, THETA, PHI, height, width):
        """One view (scalars) -> uint8 numpy [height, width, C] like the reference, or [N, ...] for a stack."""
        out = self.GetPerspectives(FOV, [THETA], [PHI], height, width)[:, 0]
        out = out.cpu().numpy()
        return out[0] if self._single else out

    def GetPerspectives(self, FOV, thetas, phis, height, width):
        """All views of all frames in one launch: device uint8 [N, M, height, width, C]."""
        maps = [e2p_maps(float(FOV), float(t), float(p), int(height), int(width), self._height, self._width) for t, p in zip(thetas, phis)]
        mx = torch.from_numpy(np.stack([m[0] for m in maps])).to(self._img.device)
        my = torch.from_numpy(np.stack([m[1] for m in maps])).to(self._img.device)
        return remap(self._img, mx, my)


class Perspective:
    """Perspec2Equirec.Perspective for an in-memory uint8 image [H, W, 3]."""

    def __init__(self, img, FOV, THETA, PHI, device="cuda"):
        if isinstance(img, str):
            raise NotImplementedError("file input is the reference's cv2.imread path; pass the decoded uint8 array")
        self._img = _as_u8_frames(img, device)
        _, self._height, self._width, _ = self._img.shape
        self.wFOV, self.THETA, self.PHI = FOV, THETA, PHI

    def GetEquirec(self, height, width):
        """-> (panorama canvas * mask, mask) uint8 / int numpy [height, width, 3], like the reference."""
        lon, lat, mask = p2e_maps(float(self.wFOV), float(self.THETA), float(self.PHI), self._height, self._width, int(height), int(width))
        dev = self._img.device
        warped = remap(self._img[:1], torch.from_numpy(lon)[None].to(dev), torch.from_numpy(lat)[None].to(dev))[0, 0].cpu().numpy()
        mask3 = np.repeat(mask[:, :, np.newaxis], 3, axis=2)
        return warped * mask3, mask3


# ------------------------------------------------------------------------------------------------ the script's helpers
def process_equi(panovid_data, thetas, phis, pers_resolution=256, back_norm=True, device="cuda", keep_on_device=False):
    """inference_dual_p2e.py:113-144: panorama video [f, c, h, w] float (in (-1, 1) when back_norm) -> perspective views
    [f, m, c, h, w] float32 (a CPU tensor like the reference's, or the device tensor with ``keep_on_device``: the script
    moves it to the GPU next anyway).  One launch for all f x m warps."""
    thetas = np.asarray(torch.as_tensor(thetas).squeeze().cpu(), np.float64).reshape(-1)
    phis = np.asarray(torch.as_tensor(phis).squeeze().cpu(), np.float64).reshape(-1)
    pano = torch.as_tensor(panovid_data).float().to(device)                               # fp32 add / mul round identically on the device
    pano = (pano + 1) * 127.5 if back_norm else pano * 255
    frames = pano.permute(0, 2, 3, 1).to(torch.uint8).contiguous()                        # numpy astype(uint8) of in-range values truncates
    views = Equirectangular(frames, device=device).GetPerspectives(90, thetas, phis, pers_resolution, pers_resolution)   # [f, m, h, w, c]
    if back_norm:
        # (img.astype(float32) / 127.5) - 1 through a 256-entry table computed with the reference's numpy expression:
        # bit-identical by construction, whatever the device's division rounding
        lut = _on_device("u8_to_unit", lambda: torch.from_numpy((np.arange(256).astype(np.float32) / 127.5) - 1), views.device)
        out = lut[views.long()]
    else:
        out = (views > 0).any(dim=-1, keepdim=True).float()
    out = out.permute(0, 1, 4, 2, 3).contiguous()
    return out if keep_on_device else out.cpu()


def pers2pano_frames(persframes, ph_list, pano_H=256, pano_W=512, fov=90, th=0, device="cuda"):
    """The warp loop of pers2pano_vid (inference_dual_p2e.py:291-304) for frames whose pitch is already known: uint8 frames
    [f, h, w, 3] -> (pano_frames uint8 [f, H, W, 3], pano_mask uint8 [f, H, W, 1], 1 = to be generated)."""
    fr = _as_u8_frames(np.asarray(persframes), device)
    f, h, w, _ = fr.shape
    maps = [p2e_maps(float(fov), float(th), float(ph), h, w, int(pano_H), int(pano_W)) for ph in ph_list]
    assert len(maps) == f
    uniq = {}
    for i, ph in enumerate(ph_list):
        uniq.setdefault(float(ph), []).append(i)
    frames = torch.empty((f, pano_H, pano_W, 3), dtype=torch.uint8, device=fr.device)
    masks = np.empty((f, pano_H, pano_W, 1), np.uint8)
    for ph, idx in uniq.items():                                                          # one launch per distinct pitch
        lon, lat, mask = maps[idx[0]]
        warped = remap(fr[idx], torch.from_numpy(lon)[None].to(fr.device), torch.from_numpy(lat)[None].to(fr.device))[:, 0]
        frames[idx] = warped * torch.from_numpy(mask.astype(np.uint8)).to(fr.device)[None, :, :, None]
        masks[idx] = (1 - mask).astype(np.uint8)[None, :, :, None]
    return frames.cpu().numpy(), masks


def get_maxrec_cord(input):
    """src/modules/utils.py:39-73: (top, left, width, height) of the largest all-ones rectangle of a [h, w] mask."""
    if isinstance(input, torch.Tensor):
        input = input.cpu().numpy()
    return kernels.max_rect(input)


def get_anchor_target(pixel_values, ph_list, fov=90, th=0):
    """animatediff/utils/video_mask.py:158-217 with the reference's return tuple: (anchor_pixels_values [b, f, c, 256, 256],
    anchor_pixels_values_pers [b, f, c, h/2, h/2], target_pixels_values, masks [b, f, 1, h, w], relative_positions [b, f, 6],
    pitchs [1, f])."""
    if pixel_values.dim() == 4:
        pixel_values = pixel_values.unsqueeze(0)
    b, f, c, h, w = pixel_values.shape
    dev = pixel_values.device
    ps = int(h / 2)
    frames = ((pixel_values[0].permute(0, 2, 3, 1).float().cpu().numpy() + 1) / 2 * 255).astype(np.uint8)
    equ = Equirectangular(frames, device=dev if dev.type == "cuda" else "cuda")
    pers = np.stack([equ._view_of_frame(i, fov, th, ph_list[i], ps) for i in range(f)])            # [f, ps, ps, c]
    anchor_pers = torch.from_numpy((pers / 127.5) - 1).permute(0, 3, 1, 2).unsqueeze(0).expand(b, -1, -1, -1, -1).to(dev)
    masks, anchors, rels, pitchs = [], [], [], []
    for i in range(f):
        _, _, m = p2e_maps(float(fov), float(th), float(ph_list[i]), ps, ps, h, w)
        mask = torch.from_numpy((1 - m)[None, None].astype(np.float32)).expand(b, -1, -1, -1).to(dev)
        masks.append(mask)
        top, left, rw, rh = get_maxrec_cord(m)
        crop = pixel_values[:, i, :, top:top + rh, left:left + rw]
        anchors.append(F.interpolate(crop, size=(256, 256), mode="bilinear", align_corners=False))
        pitchs.append(torch.tensor([ph_list[i]], device=dev))
        rels.append(torch.tensor([int(h / 2 - (top + top + rh) / 2), int(w / 2 - (left + left + rw) / 2), rh, rw, h, w], device=dev))
    return (torch.stack(anchors, dim=1), anchor_pers, pixel_values.clone(), torch.stack(masks, dim=1),
            torch.stack(rels, dim=0).unsqueeze(0).repeat(b, 1, 1), torch.stack(pitchs, dim=1))


def _view_of_frame(self, i, fov, th, ph, size):
    lon, lat = e2p_maps(float(fov), float(th), float(ph), int(size), int(size), self._height, self._width)
    dev = self._img.device
    return remap(self._img[i:i + 1], torch.from_numpy(lon)[None].to(dev), torch.from_numpy(lat)[None].to(dev))[0, 0].cpu().numpy()


Equirectangular._view_of_frame = _view_of_frame
