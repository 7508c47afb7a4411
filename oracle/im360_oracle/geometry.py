"""Panorama geometry of the cross-view attention: circular pad, gnomonic maps, cross-view
masks and spherical coordinates.  TEST INFRASTRUCTURE (see package docstring).

Third-party semantics restated here (un-vendored, "parity unpinned" by the reference):
cv2.Rodrigues, kornia remap(align_corners=True) = bilinear/nearest sampling at pixel
coordinates with zero padding, kornia gaussian_blur2d((5,5),(1,1),'replicate').
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def pad_pano(x, p):
    """Circular pad of the last (longitude) axis (src/utils/pano.py:75-95)."""
    if p <= 0:
        return x
    return torch.cat([x[..., -p:], x, x[..., :p]], dim=-1)


def unpad_pano(x, p):
    """src/utils/pano.py:98-101."""
    return x if p <= 0 else x[..., p:-p]


def padding_pano(pano, padding=16, latent=False):
    """The SR stage's close-loop pad (sr/video_to_video_model.py:21-24): ``padding`` counts latent columns, pixel-space
    tensors get 8x as many; 4- or 5-dim W-last tensors only, as the reference's pad_pano (src/utils/pano.py:79-86)."""
    if not latent:
        padding *= 8
    if padding > 0 and pano.ndim not in (4, 5):
        raise NotImplementedError("pano should be 4 or 5 dim")
    return pad_pano(pano, padding)


def unpadding_pano(pano_pad, padding=16, latent=False):
    """sr/video_to_video_model.py:26-29."""
    if not latent:
        padding *= 8
    return unpad_pano(pano_pad, padding)


def circular_pad(x, pad):
    """``F.pad(x, (w1, w2, h1, h2), "circular")`` of the SR stage's pad_to_fit replacement (sr/video_to_video_model.py:99),
    restated with index arithmetic: out[..., i, j] = x[..., (i - h1) mod H, (j - w1) mod W]."""
    w1, w2, h1, h2 = pad
    H, W = x.shape[-2], x.shape[-1]
    iy = (torch.arange(H + h1 + h2) - h1) % H
    ix = (torch.arange(W + w1 + w2) - w1) % W
    return x[..., iy, :][..., ix]


def rodrigues(v):
    """Axis-angle -> rotation matrix (cv2.Rodrigues restated: I + sin(t) K + (1-cos t) K^2)."""
    v = np.asarray(v, dtype=np.float64).reshape(3)
    th = float(np.linalg.norm(v))
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


def _view_rotations(theta, phi):
    y_axis = np.array([0.0, 1.0, 0.0])
    z_axis = np.array([0.0, 0.0, 1.0])
    R1 = rodrigues(z_axis * np.radians(theta))
    R2 = rodrigues(np.dot(R1, y_axis) * np.radians(-phi))
    return R1, R2


def pers_lonlat(fov, theta, phi, h, w):
    """Longitude/latitude (radians) of every perspective pixel
    (src/utils/Perspective_and_Equirectangular/e2p.py:9-40)."""
    hfov = float(h) / w * fov
    w_len = np.tan(np.radians(fov / 2.0))
    h_len = np.tan(np.radians(hfov / 2.0))
    x_map = np.ones([h, w], np.float32)
    y_map = np.tile(np.linspace(-w_len, w_len, w), [h, 1])
    z_map = -np.tile(np.linspace(-h_len, h_len, h), [w, 1]).T
    D = np.sqrt(x_map ** 2 + y_map ** 2 + z_map ** 2)
    xyz = np.stack((x_map, y_map, z_map), axis=2) / D[:, :, None]
    R1, R2 = _view_rotations(theta, phi)
    xyz = xyz.reshape(h * w, 3).T
    xyz = (R2 @ (R1 @ xyz)).T
    lat = np.arcsin(xyz[:, 2])
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    return lon.reshape(h, w), -lat.reshape(h, w)


def pers_to_equi_pixmap(eh, ew, fov, theta, phi, h, w):
    """Equirect pixel coordinates sampled by every perspective pixel (e2p.py:43-56)."""
    lon, lat = pers_lonlat(fov, theta, phi, h, w)
    cx, cy = (ew - 1) / 2.0, (eh - 1) / 2.0
    lon = lon / np.pi * 180
    lat = lat / np.pi * 180
    return lon / 180 * cx + cx, lat / 90 * cy + cy


def equi_to_pers_pixmap(ph, pw, fov, theta, phi, h, w):
    """Perspective pixel coordinates sampled by every equirect pixel + validity mask
    (src/utils/Perspective_and_Equirectangular/p2e.py:9-53)."""
    hfov = float(ph) / pw * fov
    w_len = np.tan(np.radians(fov / 2.0))
    h_len = np.tan(np.radians(hfov / 2.0))
    x, y = np.meshgrid(np.linspace(-180, 180, w), np.linspace(90, -90, h))
    xyz = np.stack((np.cos(np.radians(x)) * np.cos(np.radians(y)),
                    np.sin(np.radians(x)) * np.cos(np.radians(y)),
                    np.sin(np.radians(y))), axis=2)
    R1, R2 = _view_rotations(theta, phi)
    R1, R2 = np.linalg.inv(R1), np.linalg.inv(R2)
    xyz = xyz.reshape(h * w, 3).T
    xyz = (R1 @ (R2 @ xyz)).T.reshape(h, w, 3)
    front = xyz[:, :, 0] > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        xyz = xyz / xyz[:, :, 0:1]
    inside = (-w_len < xyz[:, :, 1]) & (xyz[:, :, 1] < w_len) & (-h_len < xyz[:, :, 2]) & (xyz[:, :, 2] < h_len)
    lon_map = np.where(inside, (xyz[:, :, 1] + w_len) / 2 / w_len * pw, 0)
    lat_map = np.where(inside, (-xyz[:, :, 2] + h_len) / 2 / h_len * ph, 0)
    return lon_map, lat_map, inside & front


def remap(img, map_x, map_y, mode="bilinear"):
    """kornia remap(align_corners=True): sample img[n,c,H,W] at pixel coords, zero padding."""
    H, W = img.shape[-2:]
    grid = torch.stack([2.0 * map_x / (W - 1) - 1.0, 2.0 * map_y / (H - 1) - 1.0], dim=-1).to(img.dtype)
    return F.grid_sample(img, grid, mode=mode, padding_mode="zeros", align_corners=True)


def _cam_list(cameras, key):
    v = cameras[key]
    return [float(x) for x in (v.reshape(-1).tolist() if torch.is_tensor(v) else v)]


def e2p(e_img, fovs, thetas, phis, out_hw, mode="bilinear"):
    """Equirect -> perspective views, one camera per batch item (e2p.py:59-77)."""
    he, we = e_img.shape[-2:]
    lons, lats = [], []
    for fov, u, v in zip(fovs, thetas, phis):
        lon, lat = pers_to_equi_pixmap(he, we, fov, u, v, out_hw[0], out_hw[1])
        lons.append(lon)
        lats.append(lat)
    lons = torch.from_numpy(np.stack(lons)).to(e_img.dtype)
    lats = torch.from_numpy(np.stack(lats)).to(e_img.dtype)
    return remap(e_img, lons, lats, mode)


def p2e(p_img, fovs, thetas, phis, out_hw, mode="bilinear"):
    """Perspective views -> equirect, masked by validity (p2e.py:56-72)."""
    hp, wp = p_img.shape[-2:]
    lons, lats, masks = [], [], []
    for fov, u, v in zip(fovs, thetas, phis):
        lon, lat, m = equi_to_pers_pixmap(hp, wp, fov, u, v, out_hw[0], out_hw[1])
        lons.append(lon)
        lats.append(lat)
        masks.append(m[None])
    lons = torch.from_numpy(np.stack(lons)).to(p_img.dtype)
    lats = torch.from_numpy(np.stack(lats)).to(p_img.dtype)
    mask = torch.from_numpy(np.stack(masks))
    return remap(p_img, lons, lats, mode) * mask


def _gauss5():
    x = torch.arange(5, dtype=torch.float64) - 2.0
    g = torch.exp(-(x ** 2) / 2.0)
    return (g / g.sum()).float()


def gaussian_blur5(x):
    """kornia gaussian_blur2d(x, (5,5), (1,1), border_type='replicate') on [n,1,h,w]."""
    g = _gauss5()
    xp = F.pad(x, (2, 2, 2, 2), mode="replicate")
    xp = F.conv2d(xp, g.view(1, 1, 1, 5))
    return F.conv2d(xp, g.view(1, 1, 5, 1))


def _raw_masks(ph, pw, eh, ew, cameras, opposite):
    """get_masks / get_oppo_masks (src/utils/utils.py:43-89, 91-142): one-hot pixel images
    warped through e2p / p2e, then the 'missing pixel' symmetrisation."""
    fov, theta, phi = (_cam_list(cameras, k) for k in ("FoV", "theta", "phi"))
    m = len(fov)
    ne, npx = eh * ew, ph * pw
    eye_p = torch.eye(npx).reshape(1, npx, ph, pw).expand(m, -1, -1, -1)
    eye_e = torch.eye(ne).reshape(ne, eh, ew)
    if opposite:
        # one-hot at the antipodal column (x + ew/2) of every source pixel
        eye_e = torch.roll(eye_e, shifts=ew // 2, dims=2)
    eye_e = eye_e.reshape(1, ne, eh, ew).expand(m, -1, -1, -1)
    pers_masks = e2p(eye_e, fov, theta, phi, (ph, pw))                        # [m, ne, ph, pw]
    th2 = [t + 180 for t in theta] if opposite else theta
    equi_masks = p2e(eye_p, fov, th2, phi, (eh, ew))                          # [m, np, eh, ew]
    pers_masks = pers_masks.reshape(m, ne, npx)
    equi_masks = equi_masks.reshape(m, npx, ne)
    pers_masks = torch.clamp(pers_masks + equi_masks.transpose(1, 2), 0, 1)
    equi_masks = torch.clamp(equi_masks + pers_masks.transpose(1, 2), 0, 1)
    return pers_masks.reshape(m, eh, ew, ph, pw), equi_masks.reshape(m, ph, pw, eh, ew)


def merged_masks(ph, pw, eh, ew, cameras, opposite):
    """get_merged_masks (src/utils/utils.py:12-41) with the coin flip taken by the caller:
    ``opposite`` = (random.random() < 0.4).  Returns pers_masks [m,eh,ew,ph,pw] and
    equi_masks [m,ph,pw,eh,ew], values in [-1, 1]."""
    pers_masks, equi_masks = _raw_masks(ph, pw, eh, ew, cameras, opposite)
    m = pers_masks.shape[0]
    pm = gaussian_blur5(pers_masks.reshape(-1, 1, ph, pw))
    em = unpad_pano(gaussian_blur5(pad_pano(equi_masks.reshape(-1, 1, eh, ew), 2)), 2)

    def norm(t):
        mx = torch.amax(t, dim=(1, 2, 3), keepdim=True)
        mx[mx == 0] = 1.0
        return t / mx * 2 - 1

    return norm(pm).reshape(m, eh, ew, ph, pw), norm(em).reshape(m, ph, pw, eh, ew)


def coords(ph, pw, eh, ew, cameras):
    """get_coords (src/utils/utils.py:145-164): (lon, lat) of every pixel."""
    x, y = np.meshgrid(np.linspace(-np.pi, np.pi, ew), np.linspace(np.pi / 2, -np.pi / 2, eh))
    equi = torch.tensor(np.stack([x, y]), dtype=torch.float32).permute(1, 2, 0)
    fov, theta, phi = (_cam_list(cameras, k) for k in ("FoV", "theta", "phi"))
    pers = []
    for f, t, p in zip(fov, theta, phi):
        lon, lat = pers_lonlat(f, t, p, ph, pw)
        pers.append(torch.tensor(np.stack([lon, lat]), dtype=torch.float32))
    return torch.stack(pers).permute(0, 2, 3, 1), equi


def spherical_pe(coords_, n_freqs):
    """SphericalPE (src/modules/transformer.py:170-206)."""
    base = 2 if n_freqs <= 80 else 5000 ** (1 / (n_freqs / 2.5))
    freq = base ** torch.linspace(0, n_freqs - 1, n_freqs)
    shape = coords_.shape[:-1]
    enc = coords_.reshape(-1, 2, 1) * freq
    pe = torch.cat([enc.sin(), enc.cos()], dim=1)
    return pe.reshape(*shape, -1)
