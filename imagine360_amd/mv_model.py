"""Dual-branch coupled model: MultiViewBaseModel + WarpAttn (src/models/MVGenModel.py:16-481,
src/modules/attn_perspano.py:10-99, src/modules/transformer.py:43-206), channels-last on HIP kernels.

What changed relative to the reference, deliberately:
  * masks / spherical coordinates / positional encodings are step-invariant: built once per resolution
    (both the normal and the antipodal variant) and cached on the device; per call only the reference's
    ``random.random() < 0.4`` coin is drawn, so Python-RNG parity holds;
  * both WarpAttn directions share LN(x + pe) of each side and one fused QKV GEMM per side;
  * the IP-adapter conditioning (TemporalProjection + Resampler) is hoisted out of the loop; the
    per-step ``randn_like`` noise (MVGenModel.py:186-187) is still drawn every call, pano first;
  * no empty_cache()/flush() syncs, no circular-pad copies (folded into the kernels, ``pano=True``).
"""
import random

import torch

LOG2E = 1.4426950408889634
import torch.nn as nn
import torch.nn.functional as F

from . import kernels
from . import pano_geometry as G
from .layers import DerivedCache, FeedForward, QKVAttention, layer_norm, publish_to_all_streams, to_cl


class SphericalPE(nn.Module):
    """[sin(lon f), sin(lat f), cos(lon f), cos(lat f)], f = base^k (transformer.py:170-206).  Evaluated
    in fp32 exactly as the reference's CPU path does (the top frequency at C=320 is 2^79, so a 16-bit
    evaluation is numerically arbitrary -- SURVEY.md section 7) and cast afterwards."""

    def __init__(self, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = N_freqs
        base = 2 if N_freqs <= 80 else 5000 ** (1 / (N_freqs / 2.5))
        self.register_buffer("freq_bands", base ** torch.linspace(0, N_freqs - 1, N_freqs))

    def forward(self, coords):
        shape = coords.shape[:-1]
        base = 2 if self.N_freqs <= 80 else 5000 ** (1 / (self.N_freqs / 2.5))
        # fp32 and built ON THE HOST like the reference's buffer (transformer.py:184-188, at module construction), not from
        # the (castable) buffer and not on the device: the GPU's pow() is not exact for 2^31 ... 2^63, and a one-ulp
        # frequency turns sin / cos of such arguments into different numbers (measured: 50 % relative difference of the
        # table at 128 / 256 channels, 5e-3 on the panorama prediction).  sin / cos are evaluated on the host too -- arguments
        # up to 2^79 * pi only reproduce the CPU reference with the same libm; once per resolution (WarpAttn caches it)
        freq = base ** torch.linspace(0, self.N_freqs - 1, self.N_freqs)
        enc = coords.float().cpu().reshape(-1, 2, 1) * freq
        return torch.cat([enc.sin(), enc.cos()], dim=1).reshape(*shape, -1).to(coords.device)


class _CrossViewBlock(nn.Module):
    """src/modules/transformer.py:135-167 with its parameter names (attn1.to_q/to_k/to_v/to_out, ff, norm1, norm2);
    the output projections are zero-initialised like the reference (transformer.py:30-32, 55-57)."""

    def __init__(self, dim, n_heads, d_head):
        super().__init__()
        self.attn1 = QKVAttention(dim, dim, n_heads, d_head, out_list=False)
        self.ff = FeedForward(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        for p in (self.attn1.to_out.weight, self.attn1.to_out.bias, self.ff.net[2].weight, self.ff.net[2].bias):
            nn.init.zeros_(p)


class WarpAttn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.transformer = _CrossViewBlock(dim, dim // 32, 32)
        self.mv_attn = _CrossViewBlock(dim, dim // 32, 32)        # present in reference checkpoints, never used
        self.pe = SphericalPE(dim // 4)
        self.dim = dim
        self._geom = {}
        self._geom_extra = {}
        self.mask_block_maps = True     # (A/B, tests) False: the kernel adds the shifted masks everywhere, no block skipping

    def geometry(self, ph, pw, eh, ew, cameras, opposite, device, dtype):
        fov, theta, phi = G.camera_lists(cameras)
        key = (ph, pw, eh, ew, bool(opposite), str(device), dtype, tuple(fov), tuple(theta), tuple(phi))
        if key not in self._geom:
            with torch.no_grad():
                b_e2p, b_p2e = G.cross_view_bias(ph, pw, eh, ew, cameras, opposite, device)
                pc, ec = G.spherical_coords(ph, pw, eh, ew, cameras)
                pers_pe = self.pe(pc.to(device)).reshape(-1, self.dim)          # (m h w) c
                equi_pe = self.pe(ec.to(device)).reshape(-1, self.dim)          # (h w) c
                b_e2p, b_p2e = b_e2p.to(dtype), b_p2e.to(dtype)
                # head dim 32: the masks go to the kernel as fp16 * log2(e) and are added by the matrix pipe.  Round 6: SHIFTED by minus
                # their minimum -- softmax is invariant under a per-row constant, and the ~97 % of a mask that is background (-1) becomes
                # exactly zero -- with a block map that lets the kernel skip the all-zero 32 x 32 blocks (2 of its 6 MFMAs per block and
                # the scattered fragment loads); ``geometry_extra`` returns the maps and the shifts
                packed = all(kernels.can_pack_attn_bias(32, b.shape[1]) for b in (b_e2p, b_p2e))
                extra = {}
                if packed:
                    def pack(b):
                        shift = -float(b.float().min())
                        pm = ((b.float() + shift) * LOG2E).clamp(-60000.0, 60000.0).to(torch.float16).contiguous()
                        return pm, kernels.attn_bias_blocks(pm), shift
                    b_e2p, extra["blocks_e2p"], extra["shift_e2p"] = pack(b_e2p)
                    b_p2e, extra["blocks_p2e"], extra["shift_p2e"] = pack(b_p2e)
                self._geom[key] = (b_e2p, b_p2e, pers_pe.to(dtype), equi_pe.to(dtype), packed)
                self._geom_extra[key] = extra
                publish_to_all_streams(self._geom[key][:4] + tuple(v for v in extra.values() if torch.is_tensor(v)))
        return self._geom[key]

    def geometry_extra(self, ph, pw, eh, ew, cameras, opposite, device, dtype):
        """Block maps (``kernels.attn_bias_blocks``) and shifts of the packed masks ``geometry`` returns for the same arguments
        ({} when the masks are not packed): packed = fp16((mask + shift) * log2 e)."""
        self.geometry(ph, pw, eh, ew, cameras, opposite, device, dtype)
        fov, theta, phi = G.camera_lists(cameras)
        return self._geom_extra[(ph, pw, eh, ew, bool(opposite), str(device), dtype, tuple(fov), tuple(theta), tuple(phi))]

    def forward_cl(self, pers, equi, cameras, frames, opposite=None, sel=None, side=None, diag=(False, False)):
        """pers [(b m) f, ph, pw, C], equi [b f, eh, ew, C] channels-last -> same shapes.  ``sel``: device int32
        scalar holding the normal (0) / antipodal (1) mask choice; then both variants are handed to the kernel and
        the choice is made on the device, which keeps the whole denoising step replayable from a hipGraph."""
        t = self.transformer
        nf, ph, pw, c = pers.shape
        ne_img, eh, ew, _ = equi.shape
        b = ne_img // frames
        m = nf // ne_img
        alt_e2p = alt_p2e = None
        xa = {}
        if sel is not None:
            b_e2p, b_p2e, pers_pe, equi_pe, packed = self.geometry(ph, pw, eh, ew, cameras, False, pers.device, pers.dtype)
            alt_e2p, alt_p2e, _, _, _ = self.geometry(ph, pw, eh, ew, cameras, True, pers.device, pers.dtype)
            xg = self.geometry_extra(ph, pw, eh, ew, cameras, False, pers.device, pers.dtype)
            xa = self.geometry_extra(ph, pw, eh, ew, cameras, True, pers.device, pers.dtype)
        else:
            if opposite is None:
                opposite = random.random() < 0.4                 # the reference's coin, one draw per call
            b_e2p, b_p2e, pers_pe, equi_pe, packed = self.geometry(ph, pw, eh, ew, cameras, opposite, pers.device, pers.dtype)
            xg = self.geometry_extra(ph, pw, eh, ew, cameras, opposite, pers.device, pers.dtype)
        if not self.mask_block_maps:
            xg, xa = {}, {}
        eq = equi.reshape(b * frames, eh * ew, c)
        # (b m) f (h w) c -> (b f) (m h w) c
        pr = pers.reshape(b, m, frames, ph * pw, c).permute(0, 2, 1, 3, 4).reshape(b * frames, m * ph * pw, c)
        eq_n = layer_norm(t.norm1, eq, pre=equi_pe)             # LN(x + pe), the PE add fused into the norm
        pr_n = layer_norm(t.norm1, pr, pre=pers_pe)
        qkv_e, qkv_p = t.attn1.qkv(eq_n), t.attn1.qkv(pr_n)
        h = t.attn1.heads
        # residual adds ride in the GEMM epilogues where the token count takes the MFMA kernel (same rounding sequence as
        # Linear -> + residual; hipBLASLt + add otherwise)
        def equi_chain(eq):
            a_e = kernels.attention(qkv_e[..., :c], qkv_p[..., c:2 * c], qkv_p[..., 2 * c:], h, bias=b_e2p, bias_alt=alt_e2p, bias_sel=sel, bias_packed=packed,
                                    bias_blocks=xg.get("blocks_e2p"), bias_blocks_alt=xa.get("blocks_e2p"))
            eq, st = t.attn1.out_proj(a_e, residual=eq, row_stats=True)
            return t.ff(eq, residual=eq, ln=t.norm2, stats=st)          # LayerNorm folded into the GEGLU GEMM where both take the MFMA kernel

        def pers_chain(pr):
            a_p = kernels.attention(qkv_p[..., :c], qkv_e[..., c:2 * c], qkv_e[..., 2 * c:], h, bias=b_p2e, bias_alt=alt_p2e, bias_sel=sel, bias_packed=packed,
                                    bias_blocks=xg.get("blocks_p2e"), bias_blocks_alt=xa.get("blocks_p2e"))
            pr, st = t.attn1.out_proj(a_p, residual=pr, row_stats=True)
            return t.ff(pr, residual=pr, ln=t.norm2, stats=st)

        if side is not None and eq.is_cuda:
            # once both projections exist the two directions are independent: the panorama's (the smaller grids) on the side
            # stream.  eq / qkv_e / qkv_p were allocated on this stream and stay referenced by this frame until after the join.
            main = torch.cuda.current_stream()
            eager = not torch.cuda.is_current_stream_capturing()
            if eager and diag[1]:
                torch.cuda.synchronize()
            side.wait_stream(main)
            if eager and diag[0]:
                for tns in (eq, qkv_e, qkv_p, b_e2p, alt_e2p):          # main-stream tensors the side stream reads
                    if torch.is_tensor(tns):
                        tns.record_stream(side)
            with torch.cuda.stream(side):
                eq_out = equi_chain(eq)
            pr = pers_chain(pr)
            main.wait_stream(side)
            if eager and diag[0]:
                eq_out.record_stream(main)                              # side-stream tensor the main stream reads from here on
            if eager and diag[1]:
                torch.cuda.synchronize()
            eq = eq_out
        else:
            eq = equi_chain(eq)
            pr = pers_chain(pr)
        pers_out = pr.reshape(b, frames, m, ph, pw, c).permute(0, 2, 1, 3, 4, 5).reshape(nf, ph, pw, c)
        return pers_out.contiguous(), eq.reshape(ne_img, eh, ew, c)

    def forward(self, pers_x, equi_x, cameras):
        p, f = to_cl(pers_x)
        e, _ = to_cl(equi_x)
        po, eo = self.forward_cl(p, e, cameras, f)
        from .layers import from_cl
        return from_cl(po, f), from_cl(eo, f)


class MultiViewBaseModel(nn.Module):
    """Interleaved block-by-block forward of the perspective and panorama UNets with 7 WarpAttn
    (src/models/MVGenModel.py:16-481)."""

    def __init__(self, unet, pano_unet, pano_pad=True, device="cuda"):
        super().__init__()
        self.unet, self.pano_unet, self.pano_pad = unet, pano_unet, pano_pad
        self.cp_blocks_encoder = nn.ModuleList([WarpAttn(blk.downsamplers[-1].out_channels)
                                                for blk in unet.down_blocks if blk.downsamplers is not None])
        self.cp_blocks_mid = WarpAttn(unet.mid_block.resnets[-1].out_channels)
        self.cp_blocks_decoder = nn.ModuleList([WarpAttn(blk.upsamplers[0].channels)
                                                for blk in unet.up_blocks if blk.upsamplers is not None])
        self.trainable_parameters = [(list(self.cp_blocks_mid.parameters()) + list(self.cp_blocks_decoder.parameters())
                                      + list(self.cp_blocks_encoder.parameters()), 1.0)]
        self.noise_on_host = False      # True: draw the per-step IP noise from the CPU generator (CPU-reference RNG parity)
        self.taps = None                # dict -> records (pers, equi) after each WarpAttn, for tests
        self._rig_cache = {}
        self._coins_dev = None          # int32[8] on the device: the 7 WarpAttn coins of the current step
        self.coins_preloaded = False    # True while a captured graph replays: the driver draws + uploads the coins
        self._ip_noise_half = None      # (half index, 2) when this rank runs one CFG half (BASELINE config 5 layout)
        self.dual_stream = True         # the panorama branch's segments between WarpAttn calls run on a side stream (GPU, unsharded)
        self._sharded = False
        self._shard_two_comms = False
        self.dual_stream_eager = False  # (tests / A-B only) use the side stream for eagerly issued steps too; see _two_streams
        self.dual_stream_shard = False  # opt-in: keep the side stream under a frame shard WHEN the panorama UNet has its own communicator
                                        # (set_frame_shard(shard, pano_shard)); never measured on more than one GPU, off by default
        self.warp_streams = True        # with dual_stream: the two directions of every WarpAttn on the two streams as well
        self._streams = {}
        self.tap_fn = None              # (diagnostics, tools/dual_stream_race.py) taps / debug_taps keep tap_fn(tensor) instead of the tensor
        self.record_streams = False     # (diagnostics, eager issue only) Tensor.record_stream on everything that crosses the two streams
        self.sync_joins = False         # (diagnostics, eager issue only) device synchronise at every fork / join of the side stream

    def _run_pair(self, pers_fn, pano_fn, inputs):
        """One segment of each branch between two WarpAttn calls (they share nothing but read-only conditioning).  With
        ``dual_stream`` on a GPU the panorama segment is issued on a side stream forked from / joined back into the current
        one, so its small grids (16 - 512 workgroups at levels 1 - 3) fill the CUs the perspective kernels' tails leave idle;
        captured into the step's hipGraph as two parallel branches.  Allocator safety without record_stream: ``inputs`` (the
        main-stream tensors the side stream reads first) stay referenced until the join, conv_in's output (the last skip the
        panorama decoder pops) by _trunk's arguments until all segments are done, and everything the side stream allocates is
        next touched by the main stream only after the join / by the side stream only after the next fork.  Measured on cfg2:
        327.9 -> 312.9 ms per step (-4.6 %), bit-identical results (test_dual_stream_forward_is_bit_identical_eager_and_graphed)."""
        x0 = inputs[0]
        if not (self._two_streams() and torch.is_tensor(x0) and x0.is_cuda):
            pers_fn()
            pano_fn()
            return
        main = torch.cuda.current_stream()
        side = self._stream(0, x0.device)
        eager = not torch.cuda.is_current_stream_capturing()
        if eager and self.sync_joins:
            torch.cuda.synchronize()
        side.wait_stream(main)
        if eager and self.record_streams:
            for t in inputs:                       # main-stream tensors the side stream reads
                if torch.is_tensor(t):
                    t.record_stream(side)
        with torch.cuda.stream(side):
            pano_fn()
        pers_fn()
        main.wait_stream(side)
        if eager and self.sync_joins:
            torch.cuda.synchronize()
        del inputs

    def _two_streams(self):
        """The side stream is used inside a hipGraph capture (where stream dependencies become graph edges and memory comes from
        the graph's private pool) -- the default, timed path; verified bit-identical to the one-stream eager step at full size by
        bench.py's parity_check and tests/test_model_gpu.py.  Eagerly issued steps stay on one stream unless ``dual_stream_eager``
        is set.  History: round 4 saw the eager two-stream step at cfg2 size not bit-identical in one bench run of two; round 5 could
        NOT reproduce it (60 instrumented runs on two boxes bit-identical, profiles/r05_dual_stream_race.log) after removing two debug
        dictionaries that pinned every down-block output -- whether that was the cause is unknown, so the hazard counts as "not
        reproduced", not as "fixed": eager issue stays on one stream (it is host-bound anyway), GraphedDenoiseStep's warm-up sets
        ``dual_stream_eager`` so that both allocator pools exist before the capture (its outputs are discarded)."""
        if not self.dual_stream or (self._sharded and not (self.dual_stream_shard and self._shard_two_comms)):
            return False
        return self.dual_stream_eager or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())

    def draw_coins(self, device):
        """The reference draws ``random.random() < 0.4`` once per WarpAttn call (src/utils/utils.py:15), 7 per step in
        execution order enc0..2, mid, dec0..2; nothing else consumes Python's RNG in between, so drawing the 7 up
        front keeps the stream identical.  They go to a small device tensor the attention kernels read."""
        if self._coins_dev is None or self._coins_dev.device != torch.device(device):
            self._coins_dev = torch.zeros(8, dtype=torch.int32, device=device)
        coins = [1 if random.random() < 0.4 else 0 for _ in range(7)] + [0]
        if self._coins_dev.is_cuda:
            # pinned staging ring + non-blocking copy: no host-device sync per step (a copy from pageable memory is one)
            if getattr(self, "_coin_ring", None) is None:
                self._coin_ring = ([torch.empty(8, dtype=torch.int32, pin_memory=True) for _ in range(8)], [None] * 8, 0)
            slots, events, i = self._coin_ring
            k = i % len(slots)
            if events[k] is not None:
                events[k].synchronize()
            slots[k].copy_(torch.tensor(coins, dtype=torch.int32))
            self._coins_dev.copy_(slots[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events[k] = ev
            self._coin_ring = (slots, events, i + 1)
        else:
            self._coins_dev.copy_(torch.tensor(coins, dtype=torch.int32))
        return self._coins_dev

    def _rig(self, cameras, m):
        """Camera dict flattened to [m, ...] plus host-side FoV/theta/phi lists, cached so a denoising loop
        does not sync on the (device-resident) camera tensors every step."""
        th = cameras["theta"]
        key = (th.data_ptr(), th._version, m) if torch.is_tensor(th) else (id(th), 0, m)
        hit = self._rig_cache.get(key)
        if hit is None:
            cams = {k: (v.reshape(-1, *v.shape[2:])[:m] if torch.is_tensor(v) and v.dim() >= 2 else v)
                    for k, v in cameras.items()}
            cams["_lists"] = G.camera_lists(cams)
            self._rig_cache = {key: cams}
            hit = cams
        return hit

    def set_frame_shard(self, shard, pano_shard=None):
        """Frame-chunk sharding (imagine360_amd.dist.FrameShard): the model is fed this rank's frames and every
        motion-module attention exchanges tokens with one all-to-all each way.  ``pano_shard``: a second FrameShard over the
        same ranks on ITS OWN process group (``dist.frame_shard_pair``) for the panorama UNet's motion modules -- collectives of
        one communicator have to stay on one stream, so only with two communicators may the panorama branch keep its side
        stream under a shard (``dual_stream_shard``, opt-in)."""
        from .unet3d import TemporalTransformer3DModel, VersatileAttention
        self._sharded = shard is not None       # (collectives of one communicator stay on one stream: no side stream then)
        self._shard_two_comms = (shard is not None and pano_shard is not None and pano_shard.group is not None
                                 and pano_shard.group is not shard.group)
        pano_mods = {id(m) for m in self.pano_unet.modules()}
        for mod in self.modules():
            if isinstance(mod, (VersatileAttention, TemporalTransformer3DModel)):
                sh = (pano_shard if (pano_shard is not None and shard is not None and id(mod) in pano_mods) else shard)
                # the exchange sits at the module boundary (default) or around every attention: exactly one of the two holds the shard
                at_module = sh is not None and sh.boundary == "module"
                mod.frame_shard = sh if at_module == isinstance(mod, TemporalTransformer3DModel) else None

    def _ip_noise(self, like):
        half = self._ip_noise_half
        if half is not None:
            # this rank runs ONE of the CFG halves (dist.cfg_half_inputs): draw the whole CFG batch's noise, keep our rows,
            # so the two rank groups together reproduce the CFG-batched run's stream
            idx, n = half
            full = torch.empty((like.shape[0] * n, *like.shape[1:]), dtype=like.dtype, device=like.device)
            return self._ip_noise_full(full).chunk(n)[idx]
        return self._ip_noise_full(like)

    def _ip_noise_full(self, like):
        if self.noise_on_host:
            return torch.randn(like.shape, dtype=torch.float32).to(device=like.device, dtype=like.dtype)
        return torch.randn_like(like)

    def _stream(self, k, device):
        st = self._streams.get(k)
        if st is None or st.device != torch.device(device):
            st = self._streams[k] = torch.cuda.Stream(device=device)
        return st

    def _trunk(self, x, px, emb, pemb, ctx, pctx, cams, f, coins):
        """Everything between conv_in and the output layout for the given rows of the batch: both UNets block by block, the
        7 WarpAttn, conv_out.  (Measured and dropped: the two rows of the CFG batch as two concurrent trunks on their own streams --
        half-sized grids per kernel cost more than the filled tails give back: 323 vs 304 ms per cfg2 step.)"""
        un, pu = self.unet, self.pano_unet
        pano = self.pano_pad
        taps = self.taps
        order = {"enc0": 0, "enc1": 1, "enc2": 2, "mid": 3, "dec0": 4, "dec1": 5, "dec2": 6}

        warp_side = self._stream(0, x.device) if (self._two_streams() and self.warp_streams and x.is_cuda) else None

        tap_fn = self.tap_fn if self.tap_fn is not None else (lambda t: t)
        diag = (self.record_streams, self.sync_joins)

        def warp(blk, name, a, e):
            a, e = blk.forward_cl(a, e, cams, f, sel=coins[order[name]], side=warp_side, diag=diag)
            if taps is not None:
                taps[name] = (tap_fn(a), tap_fn(e))
            return a, e

        # ---- the two branches between WarpAttn calls are independent: `pair` runs one segment of each, the panorama's on a
        #      side stream when dual_stream is set (see _run_pair)
        dbg = getattr(self, "debug_taps", None)
        skips, pskips = [x], [px]
        st = {"x": x, "px": px}
        dbx, dbp = {}, {}              # per-branch debug taps, merged after each segment

        def pair(pers_fn, pano_fn):
            self._run_pair(pers_fn, pano_fn, (st["x"], st["px"]))
            if dbg is not None:
                for k in dbx:
                    dbg[k] = (tap_fn(dbx[k]), tap_fn(dbp[k]))
                dbx.clear()
                dbp.clear()

        def down_pers(i):
            def fn():
                db, x = un.down_blocks[i], st["x"]
                for j in range(len(db.resnets)):
                    x = db.resnets[j].forward_cl(x, emb, f)
                    if dbg is not None:
                        dbx[f"res{i}{j}"] = x
                    if db.has_cross_attention:           # DownBlock3D's motion modules are skipped (:292-303)
                        x = db.attentions[j].forward_cl(x, ctx, f)
                        if dbg is not None:
                            dbx[f"attn{i}{j}"] = x
                        if db.motion_modules[j] is not None:
                            x = db.motion_modules[j].forward_cl(x, f)
                        if dbg is not None:
                            dbx[f"mm{i}{j}"] = x
                    skips.append(x)
                if db.downsamplers is not None:
                    x = db.downsamplers[0].forward_cl(x)
                    skips.append(x)
                st["x"] = x
            return fn

        def down_pano(i):
            def fn():
                pdb, px = pu.down_blocks[i], st["px"]
                for j in range(len(pdb.resnets)):
                    px = pdb.resnets[j].forward_cl(px, pemb, f, pano)
                    if dbg is not None:
                        dbp[f"res{i}{j}"] = px
                    if pdb.has_cross_attention:
                        px = pdb.attentions[j].forward_cl(px, pctx, f)
                        if dbg is not None:
                            dbp[f"attn{i}{j}"] = px
                        if pdb.motion_modules[j] is not None:
                            px = pdb.motion_modules[j].forward_cl(px, f)
                        if dbg is not None:
                            dbp[f"mm{i}{j}"] = px
                    pskips.append(px)
                if pdb.downsamplers is not None:
                    px = pdb.downsamplers[0].forward_cl(px, pano)
                    pskips.append(px)
                st["px"] = px
            return fn

        def up_pers(i, first=None):
            def fn():
                x = st["x"]
                if first is not None:
                    x = first(x)
                ub = un.up_blocks[i]
                for j in range(len(ub.resnets)):
                    # skip connections: the ResnetBlock reads (x, skip) in place, torch.cat([x, skip]) is never written
                    x = ub.resnets[j].forward_cl((x, skips.pop()), emb, f)
                    if ub.has_cross_attention:           # UpBlock3D's motion modules are skipped (:426-443)
                        x = ub.attentions[j].forward_cl(x, ctx, f)
                        if ub.motion_modules[j] is not None:
                            x = ub.motion_modules[j].forward_cl(x, f)
                st["x"] = x
            return fn

        def up_pano(i, first=None):
            def fn():
                px = st["px"]
                if first is not None:
                    px = first(px)
                pub = pu.up_blocks[i]
                for j in range(len(pub.resnets)):
                    px = pub.resnets[j].forward_cl((px, pskips.pop()), pemb, f, pano)
                    if pub.has_cross_attention:
                        px = pub.attentions[j].forward_cl(px, pctx, f)
                        if pub.motion_modules[j] is not None:
                            px = pub.motion_modules[j].forward_cl(px, f)
                st["px"] = px
            return fn

        # ---- down (MVGenModel.py:261-326); the last down block has no downsampler: it runs into the mid block (:336-380)
        nd = len(un.down_blocks)
        for i in range(nd):
            if un.down_blocks[i].downsamplers is not None:
                pair(down_pers(i), down_pano(i))
                st["x"], st["px"] = warp(self.cp_blocks_encoder[i], f"enc{i}", st["x"], st["px"])
            else:
                assert i == nd - 1, "only the last down block runs into the mid block"
                dp, dq = down_pers(i), down_pano(i)

                def mid_pers():
                    dp()
                    st["x"] = un.mid_block.forward_cl(st["x"], emb, ctx, f)

                def mid_pano():
                    dq()
                    st["px"] = pu.mid_block.forward_cl(st["px"], pemb, pctx, f, pano)
                pair(mid_pers, mid_pano)
        st["x"], st["px"] = warp(self.cp_blocks_mid, "mid", st["x"], st["px"])
        # ---- up (:395-458): a block's WarpAttn sits in front of its upsampler, so the upsample opens the next segment
        up_x = up_px = None
        for i, (ub, pub) in enumerate(zip(un.up_blocks, pu.up_blocks)):
            pair(up_pers(i, up_x), up_pano(i, up_px))
            up_x = up_px = None
            if ub.upsamplers is not None:
                st["x"], st["px"] = warp(self.cp_blocks_decoder[i], f"dec{i}", st["x"], st["px"])
                up_x = (lambda u: (lambda t: u.forward_cl(t)))(ub.upsamplers[0])
                up_px = (lambda u: (lambda t: u.forward_cl(t, pano)))(pub.upsamplers[0])
        x, px = st["x"], st["px"]
        if up_x is not None:           # (not the case for the SD layout: the last up block has no upsampler)
            x, px = up_x(x), up_px(px)
        # ---- out (:462-479)
        x = un.conv_out_cl(x)
        px = pu.conv_out_cl(px, pano)
        return x, px

    def forward(self, latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras, use_fps_condition,
                use_ip_plus_cross_attention, fps_tensor_pano, fps_tensor_pers, reference_images_clip_feat_pano,
                reference_images_clip_feat_pers, relative_position_tensor, pitchs_tensor):
        if not use_ip_plus_cross_attention:
            raise NotImplementedError("the reference forward cannot run without use_ip_plus_cross_attention "
                                      "(MVGenModel.py:245-246 never binds the encoder states)")
        un, pu = self.unet, self.pano_unet
        dt = un.dtype
        b, m, c, f, h, w = latents.shape
        cams = self._rig(cameras, m)
        # ---- time / fps embeddings (MVGenModel.py:104-133)
        ts = timestep.reshape(-1)[:1]
        fps_pers = fps_tensor_pers.reshape(-1).expand(b * m) if use_fps_condition else None
        fps_pano = fps_tensor_pano.reshape(-1).expand(b) if use_fps_condition else None
        emb = un.time_embed(ts.expand(b * m), fps_pers)
        pemb = pu.time_embed(ts.expand(b), fps_pano)
        # ---- inputs to channels-last, channel-padded for the MFMA conv
        x = latents.to(dt).permute(0, 1, 3, 4, 5, 2).reshape(b * m * f, h, w, c)
        px = pano_latent.to(dt).permute(0, 2, 3, 4, 1).reshape(b * f, *pano_latent.shape[3:], c)
        x = un.conv_in_cl(x.contiguous())
        px = pu.conv_in_cl(px.contiguous(), pano=self.pano_pad)
        # ---- IP-adapter conditioning (MVGenModel.py:155-246)
        feat_pers = reference_images_clip_feat_pers
        ip_pano = pu.ip_tokens_clean(reference_images_clip_feat_pano)
        if feat_pers.stride(1) == 0 or m == 1:       # one feature tensor shared by all views (pipeline :713)
            ip_pers = un.ip_tokens_clean(feat_pers[:, 0]).repeat_interleave(m, dim=0)
        else:
            ip_pers = un.ip_tokens_clean(feat_pers.reshape(b * m, *feat_pers.shape[2:]))
        ip_pano = ip_pano + self._ip_noise(ip_pano) * 0.1
        ip_pers = ip_pers + self._ip_noise(ip_pers) * 0.1
        if relative_position_tensor is not None and pu.use_relative_postions == "WithAdapter":
            ip_pano = ip_pano + pu.relpos_tokens(relative_position_tensor, pitchs_tensor, ip_pano.shape[1])
        pctx = torch.cat([pano_prompt_embd.to(dt), ip_pano], dim=1)
        ctx = torch.cat([prompt_embd.to(dt), ip_pers], dim=1)

        coins = self._coins_dev if self.coins_preloaded else self.draw_coins(x.device)
        dbg = getattr(self, "debug_taps", None)
        if dbg is not None:
            dbg["conv_in"] = (x, px)
            dbg["ctx"] = (ctx, pctx)
            dbg["emb"] = (emb, pemb)
        x, px = self._trunk(x, px, emb, pemb, ctx, pctx, cams, f, coins)
        co = x.shape[-1]
        sample = x.reshape(b, m, f, h, w, co).permute(0, 1, 5, 2, 3, 4)
        pano_sample = px.reshape(b, f, *px.shape[1:3], co).permute(0, 4, 1, 2, 3)
        return sample, pano_sample
