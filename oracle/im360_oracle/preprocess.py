"""TEST INFRASTRUCTURE -- CPU restatement of the reference's host preprocessing geometry.

What is pinned and what is not:
  * the sampling MAPS and masks of Equirec2Perspec.GetPerspective (src/utils/pano_utils/Equirec2Perspec.py:18-58) and
    Perspec2Equirec.GetEquirec (Perspec2Equirec.py:27-72) ARE pinned: oracle/tools/gen_golden.py preproc imports the real
    modules through ref_shims (cv2.Rodrigues restated, cv2.remap replaced by a recorder) and stores the maps the
    reference hands to cv2.remap (tests/golden/preproc.npz);
  * get_maxrec_cord (src/modules/utils.py:39-73) IS pinned: the real function is pure Python and runs in the authoring
    container;
  * cv2.remap(uint8, float32 maps, INTER_CUBIC, BORDER_WRAP) itself is **parity unpinned**: OpenCV is neither installed in
    this image nor vendored in the checkout.  `remap_cubic_wrap_u8` restates the algorithm of OpenCV 4.x
    modules/imgproc/src/imgwarp.cpp from its published source (fixed-point bicubic: 5 fractional bits per axis, a 1024 x 16
    table of int16 weights scaled by 2^15 from the A = -0.75 cubic kernel with the rounding residue of every entry folded
    into the smallest / largest of its taps (2..3, 2..3) -- as remembered from initInterTab2D, and the only reading under
    which an integer-coordinate sample, whose centre tap saturates at 32767, still returns the source pixel --, round-half-
    even coordinate quantisation, modulo border, (sum + 2^14) >> 15 saturated to uint8).  None of this can be checked
    against OpenCV here; the residue rule moves single results by at most 1 LSB.  What IS checked (round 3,
    tests/test_oracle_golden.py::test_remap_restatement_tracks_an_independent_bicubic): agreement to 1 LSB with torch's
    float bicubic grid_sample (same A = -0.75 kernel) at 1/32-pixel coordinates incl. wrap-around taps.
"""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
COEF_BITS = 15
COEF_SCALE = 1 << COEF_BITS


# ------------------------------------------------------------------------------------------------ rotations / maps
def rodrigues(v):
    """cv2.Rodrigues restated (axis-angle -> matrix): I + sin(t) K + (1 - cos t) K^2."""
    v = np.asarray(v, dtype=np.float64).reshape(3)
    th = float(np.linalg.norm(v))
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def _rotations(theta, phi):
    y_axis = np.array([0.0, 1.0, 0.0], np.float32)
    z_axis = np.array([0.0, 0.0, 1.0], np.float32)
    R1 = rodrigues(z_axis * np.radians(theta))
    R2 = rodrigues(np.dot(R1, y_axis) * np.radians(-phi))
    return R1, R2


def e2p_maps(fov, theta, phi, height, width, equ_h, equ_w):
    """(lon, lat) float32 pixel maps Equirectangular.GetPerspective passes to cv2.remap (Equirec2Perspec.py:18-58)."""
    equ_cx, equ_cy = (equ_w - 1) / 2.0, (equ_h - 1) / 2.0
    w_fov = fov
    h_fov = float(height) / width * w_fov
    w_len, h_len = np.tan(np.radians(w_fov / 2.0)), np.tan(np.radians(h_fov / 2.0))
    x_map = np.ones([height, width], np.float32)
    y_map = np.tile(np.linspace(-w_len, w_len, width), [height, 1])
    z_map = -np.tile(np.linspace(-h_len, h_len, height), [width, 1]).T
    D = np.sqrt(x_map ** 2 + y_map ** 2 + z_map ** 2)
    xyz = np.stack((x_map, y_map, z_map), axis=2) / np.repeat(D[:, :, np.newaxis], 3, axis=2)
    R1, R2 = _rotations(theta, phi)
    xyz = xyz.reshape([height * width, 3]).T
    xyz = np.dot(R1, xyz)
    xyz = np.dot(R2, xyz).T
    lat = np.arcsin(xyz[:, 2])
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    lon = lon.reshape([height, width]) / np.pi * 180
    lat = -lat.reshape([height, width]) / np.pi * 180
    lon = lon / 180 * equ_cx + equ_cx
    lat = lat / 90 * equ_cy + equ_cy
    return lon.astype(np.float32), lat.astype(np.float32)


def p2e_maps(fov, theta, phi, pers_h, pers_w, height, width):
    """(lon_map, lat_map float32, mask int [h, w]) of Perspective.GetEquirec (Perspec2Equirec.py:27-72)."""
    w_fov = fov
    h_fov = float(pers_h) / pers_w * fov
    w_len, h_len = np.tan(np.radians(w_fov / 2.0)), np.tan(np.radians(h_fov / 2.0))
    x, y = np.meshgrid(np.linspace(-180, 180, width), np.linspace(90, -90, height))
    x_map = np.cos(np.radians(x)) * np.cos(np.radians(y))
    y_map = np.sin(np.radians(x)) * np.cos(np.radians(y))
    z_map = np.sin(np.radians(y))
    xyz = np.stack((x_map, y_map, z_map), axis=2)
    R1, R2 = _rotations(theta, phi)
    R1, R2 = np.linalg.inv(R1), np.linalg.inv(R2)
    xyz = xyz.reshape([height * width, 3]).T
    xyz = np.dot(R2, xyz)
    xyz = np.dot(R1, xyz).T
    xyz = xyz.reshape([height, width, 3])
    inverse_mask = np.where(xyz[:, :, 0] > 0, 1, 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        xyz[:, :] = xyz[:, :] / np.repeat(xyz[:, :, 0][:, :, np.newaxis], 3, axis=2)
    inside = (-w_len < xyz[:, :, 1]) & (xyz[:, :, 1] < w_len) & (-h_len < xyz[:, :, 2]) & (xyz[:, :, 2] < h_len)
    lon_map = np.where(inside, (xyz[:, :, 1] + w_len) / 2 / w_len * pers_w, 0)
    lat_map = np.where(inside, (-xyz[:, :, 2] + h_len) / 2 / h_len * pers_h, 0)
    mask = np.where(inside, 1, 0) * inverse_mask
    return lon_map.astype(np.float32), lat_map.astype(np.float32), mask


# ------------------------------------------------------------------------------------------------ cv2.remap restated
def _cubic_coeffs(x):
    """interpolateCubic, float32 arithmetic, A = -0.75."""
    A = np.float32(-0.75)
    x = np.float32(x)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], np.float32)


def cubic_weight_table():
    """int16 [1024, 16]: entry (fy * 32 + fx) holds the 4 x 4 weights (row-major, scaled by 2^15) for fractions fy / 32, fx / 32."""
    tab1 = np.stack([_cubic_coeffs(np.float32(i) * np.float32(1.0 / INTER_TAB_SIZE)) for i in range(INTER_TAB_SIZE)])
    out = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 16), np.int16)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            v = (tab1[i][:, None] * tab1[j][None, :]).astype(np.float32)              # vy * vx in float32
            it = np.clip(np.rint(v * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int64)      # saturate_cast<short>
            isum = int(it.sum())
            if isum != COEF_SCALE:
                diff = isum - COEF_SCALE
                mk = Mk = (2, 2)
                for k1 in (2, 3):
                    for k2 in (2, 3):                                                   # taps ksize/2 .. ksize/2 + 1 of each axis
                        if it[k1, k2] < it[mk]:
                            mk = (k1, k2)
                        elif it[k1, k2] > it[Mk]:
                            Mk = (k1, k2)
                if diff < 0:
                    it[Mk] -= diff
                else:
                    it[mk] -= diff
            out[i * INTER_TAB_SIZE + j] = it.reshape(16).astype(np.int16)
    return out


_TABLE = None


def remap_cubic_wrap_u8(img, map_x, map_y):
    """cv2.remap(img uint8 [H, W, C], map_x, map_y float32 [h, w], INTER_CUBIC, borderMode=BORDER_WRAP) -> uint8 [h, w, C]."""
    global _TABLE
    if _TABLE is None:
        _TABLE = cubic_weight_table()
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    sx = np.rint(np.asarray(map_x, np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)      # cvRound: half to even
    sy = np.rint(np.asarray(map_y, np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767) - 1                                       # saturate_cast<short>, first tap
    iy = np.clip(sy >> INTER_BITS, -32768, 32767) - 1
    w = _TABLE[((sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1)))].astype(np.int64)   # [h, w, 16]
    acc = np.zeros(map_x.shape + (img.shape[2],), np.int64)
    for k1 in range(4):
        yy = np.mod(iy + k1, H)                                                            # BORDER_WRAP
        for k2 in range(4):
            xx = np.mod(ix + k2, W)
            acc += img[yy, xx].astype(np.int64) * w[..., k1 * 4 + k2][..., None]
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ compositions
def get_perspective(img, fov, theta, phi, height, width):
    """Equirectangular(img).GetPerspective (Equirec2Perspec.py:18-62)."""
    lon, lat = e2p_maps(fov, theta, phi, height, width, img.shape[0], img.shape[1])
    return remap_cubic_wrap_u8(img, lon, lat)


def get_equirec(img, fov, theta, phi, height, width):
    """Perspective(img, fov, theta, phi).GetEquirec(height, width) -> (persp * mask, mask [h, w, 3]) (Perspec2Equirec.py:27-72)."""
    lon, lat, mask = p2e_maps(fov, theta, phi, img.shape[0], img.shape[1], height, width)
    persp = remap_cubic_wrap_u8(img, lon, lat)
    mask3 = np.repeat(mask[:, :, np.newaxis], 3, axis=2)
    return persp * mask3, mask3


def process_equi(panovid, thetas, phis, pers_resolution=256, back_norm=True):
    """inference_dual_p2e.py:113-144: panovid [f, c, h, w] float in (-1, 1) -> perspective views [f, m, c, h, w] float."""
    pano = (np.asarray(panovid, np.float32) + 1) * 127.5 if back_norm else np.asarray(panovid, np.float32) * 255
    out = []
    for i in range(pano.shape[0]):
        frame = pano[i].transpose(1, 2, 0).astype(np.uint8)
        views = []
        for th, ph in zip(thetas, phis):
            img = get_perspective(frame, 90, th, ph, pers_resolution, pers_resolution)
            views.append((img.astype(np.float32) / 127.5) - 1 if back_norm else np.expand_dims(np.any(img > 0, axis=-1), axis=-1))
        out.append(np.stack(views))
    return np.stack(out, axis=0).astype(np.float32).transpose(0, 1, 4, 2, 3)


def pers2pano_frames(persframes, ph_list, pano_h=256, pano_w=512, fov=90, th=0):
    """The warp loop of pers2pano_vid (inference_dual_p2e.py:291-304): uint8 frames [f, h, w, 3] -> (pano frames uint8
    [f, H, W, 3], masks uint8 [f, H, W, 1] with 1 = to be generated)."""
    frames, masks = [], []
    for i in range(persframes.shape[0]):
        pano, mask = get_equirec(persframes[i], fov, th, ph_list[i], pano_h, pano_w)
        frames.append(pano.astype(np.uint8))
        m = np.any((1 - mask) > 0, axis=-1).astype(np.uint8)
        masks.append(m[..., None])
    return np.stack(frames, axis=0), np.stack(masks, axis=0)


def get_maxrec_cord(mask):
    """Largest all-ones rectangle, (top, left, width, height), with the reference's scan order and tie-breaking
    (src/modules/utils.py:39-73: column heights, then a monotone stack per row; the first strictly larger area wins)."""
    mask = np.asarray(mask)
    height, width = mask.shape
    dp = np.zeros((height, width), dtype=int)
    for i in range(height):
        for j in range(width):
            if mask[i, j] == 1:
                dp[i, j] = dp[i - 1, j] + 1 if i > 0 else 1
    max_area, max_rect = 0, (0, 0, 0, 0)
    for i in range(height):
        stack = []
        for j in range(width + 1):
            h = dp[i, j] if j < width else 0
            while stack and h < dp[i, stack[-1]]:
                top = stack.pop()
                hv = dp[i, top]
                wv = j if not stack else j - stack[-1] - 1
                if hv * wv > max_area:
                    max_area = hv * wv
                    max_rect = (i - hv + 1, stack[-1] + 1 if stack else 0, wv, hv)
            stack.append(j)
    return tuple(int(v) for v in max_rect)
