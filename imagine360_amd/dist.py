"""Multi-GPU partitioning of the denoising path over RCCL / xGMI (one process per GPU).

The reference has no distributed code; the path shards along axes on which the
reference's arithmetic is independent:

  * samples (BASELINE config 3): one prompt per GPU, weights replicated, no per-step traffic; the only
    collective is ONE all-gather of the final panorama latents at the latent boundary
    (``gather_latents``), 1 MB per rank at 16x512x1024;
  * frames (BASELINE configs 4/5): every op is per (batch, frame) image -- conv, GroupNorm, spatial and
    cross-view attention -- except the motion modules' attention over the frame axis.  ``FrameShard``
    gives each rank a contiguous chunk of frames and turns frame-sharded tokens into pixel-sharded tokens
    (and back) with one all-to-all each way PER MOTION MODULE (round 5: everything between the module's
    GroupNorm and its residual add is per pixel over frames or per token, so the whole temporal transformer
    runs pixel-sharded -- 2 C per token and module on the wire; ``boundary="attention"`` keeps round 3's
    exchange around every attention, 8 C); xGMI is point-to-point, so an all-to-all of activation slabs
    uses all 7 links at once where a ring would be bound by one.

``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    return world


def gather_latents(latent, group=None):
    """All-gather of per-rank latents at the latent boundary -> [world, *latent.shape] on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return latent.unsqueeze(0)
    out = [torch.empty_like(latent) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, latent.contiguous(), group=group)
    return torch.stack(out)


def shard_samples(items, rank=None, world=None):
    """Round-robin assignment of independent samples (prompts / seeds) to ranks."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(items)[rank::world]


FRAME_AXIS = {"latents": 3, "pano_latent": 2}       # frame axis of the model inputs that are per-frame


def shard_mv_inputs(inputs, shard):
    """Frame-chunk view of the keyword tensors of ``MultiViewBaseModel.forward``: the noisy / masked latents are cut to
    this rank's frames; everything else (text, SAM features of ALL frames, relative positions, pitches, fps) is
    step-invariant conditioning that every rank holds in full -- the IP-adapter's temporal projection and the
    per-frame position tokens are functions of the whole clip (src/models/MVGenModel.py:155-222)."""
    out = dict(inputs)
    for k, ax in FRAME_AXIS.items():
        out[k] = shard.take(inputs[k], ax).contiguous()
    return out


def cfg_half_inputs(inputs, half):
    """BASELINE config 5 splits the two classifier-free-guidance halves over two rank groups: keep the unconditional
    (half 0) or the text-conditioned (half 1) half of every CFG-batched model input.  The halves never interact inside
    the model; only the CFG combine needs both (``exchange_cfg_halves``)."""
    m = inputs["latents"].shape[1]
    out = dict(inputs)
    for k, v in inputs.items():
        if k != "timestep" and torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] in (2, 2 * m):
            out[k] = v.chunk(2)[half]                 # a view: broadcast (stride-0) feature tensors stay shared
    return out


def exchange_cfg_halves(pred, pair_group):
    """Both halves' predictions of the same frames, [uncond, text] along dim 0, on both ranks of ``pair_group``
    (= [rank of half 0, rank of half 1]): one small all-gather per branch and step."""
    parts = [torch.empty_like(pred) for _ in range(dist.get_world_size(pair_group))]      # (2; 1 in the single-GPU RCCL test)
    dist.all_gather(parts, pred.contiguous(), group=pair_group)
    return torch.cat(parts)


def cfg_frame_layout(frames, world=None, rank=None, boundary="module"):
    """Rank layout of the CFG x frames decomposition: ranks [0, world/2) take the unconditional half, the rest the text
    half; inside a half the frames are sharded.  Returns (half index, FrameShard of the half, pair group).  Every rank
    must call this (it creates process groups collectively)."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    if world % 2:
        raise ValueError("the CFG x frames layout needs an even number of ranks")
    half = world // 2
    groups = [dist.new_group(list(range(h * half, (h + 1) * half))) for h in range(2)]
    pairs = [dist.new_group([r, r + half]) for r in range(half)]
    my_half = rank // half
    return my_half, FrameShard(frames, group=groups[my_half], rank=rank % half, world=half, boundary=boundary), pairs[rank % half]


def frame_shard_pair(total_frames, group=None, boundary="module", force_collectives=False):
    """(shard, pano_shard): two FrameShards over the ranks of ``group`` -- the second on a NEW process group of the same ranks
    (a second RCCL communicator) for the panorama UNet (``MultiViewBaseModel.set_frame_shard(shard, pano_shard)``).  Collective:
    every rank of ``group`` has to call it (``dist.new_group``)."""
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
    second = dist.new_group(ranks=ranks)
    world, rank = len(ranks), ranks.index(dist.get_rank())
    return (FrameShard(total_frames, group=group, rank=rank, world=world, boundary=boundary, force_collectives=force_collectives),
            FrameShard(total_frames, group=second, rank=rank, world=world, boundary=boundary, force_collectives=force_collectives))


class FrameShard:
    """Contiguous frame chunks across the ranks of ``group`` (frames % world == 0).  ``boundary``: where the motion
    modules exchange tokens -- "module" (default: one all-to-all behind the module's GroupNorm, one in front of its
    residual add; TemporalTransformer3DModel._forward_pixel_sharded) or "attention" (around every temporal attention;
    VersatileAttention.forward).  ``force_collectives``: a shard of ONE rank normally short-circuits every exchange; with the flag it
    packs, calls the collective and unpacks like any other -- how the single-GPU test pushes the whole exchange path (pack kernel ->
    ``all_to_all_single`` -> temporal kernel on the receive buffer -> return trip, ``all_gather``) through RCCL at world size 1,
    eagerly and captured in a hipGraph (tests/test_dist_gpu.py)."""

    def __init__(self, total_frames, group=None, rank=None, world=None, boundary="module", force_collectives=False):
        if boundary not in ("module", "attention"):
            raise ValueError(f"FrameShard boundary {boundary!r}: 'module' or 'attention'")
        self.boundary = boundary
        self.force_collectives = bool(force_collectives)
        self.group = group
        self.world = dist.get_world_size(group) if world is None else world
        self.rank = dist.get_rank(group) if rank is None else rank
        if total_frames % self.world:
            raise ValueError(f"{total_frames} frames do not split over {self.world} ranks")
        self.total = total_frames
        self.local = total_frames // self.world
        self.f0 = self.rank * self.local

    def take(self, x, dim):
        """Local frame chunk of a tensor whose ``dim`` indexes all frames."""
        return x.narrow(dim, self.f0, self.local)

    def gather_frames(self, x, dim):
        """All-gather the frame chunks back along ``dim`` (latent boundary)."""
        if self.world == 1 and not self.force_collectives:
            return x
        parts = [torch.empty_like(x) for _ in range(self.world)]
        dist.all_gather(parts, x.contiguous(), group=self.group)
        return torch.cat(parts, dim=dim)

    # ---- frame-sharded tokens [B, Fl, P, C]  <->  pixel-sharded tokens of ALL frames, layout [F_total, B, PP, C] ----------
    # One all-to-all each way per motion module (boundary "module") or per motion-module attention ("attention").  The buffers are pre-sized and cached per shape (stable
    # addresses: with RCCL the exchange can be captured in a hipGraph); pack / unpack are ONE kernel each
    # (kernels.shard_pack) and the receive buffer is consumed in place by the temporal-attention kernel through its frame /
    # batch strides (``kernels.temporal_attention(..., frame_major=True)``), which writes the return trip's send buffer.
    def pixels_per_rank(self, pixels):
        return -(-pixels // self.world)

    def _buf(self, tag, shape, like):
        key = (tag, tuple(shape), like.dtype, like.device)
        cache = self.__dict__.setdefault("_buffers", {})
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return cache[key]

    def exchange(self, send, tag):
        """all_to_all of a [W, ...] buffer along its first axis into a cached receive buffer.  RCCL: device to device over
        xGMI; gloo (the CPU tests, and the single-GPU test where two ranks share one device): staged through the host."""
        recv = self._buf(tag, send.shape, send)
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            h_send = send.cpu()
            h_recv = torch.empty_like(h_send)
            dist.all_to_all_single(h_recv, h_send, group=self.group)
            recv.copy_(h_recv)
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def frames_to_pixels(self, x):
        """x [B, Fl, P, C] (this rank's frames) -> [F_total * B * PP, C]: this rank's PP pixels of EVERY frame, rows ordered
        (frame, batch, pixel); pixels past P (last rank) are zero rows."""
        from . import kernels
        b, fl, p, c = x.shape
        w, pp = self.world, self.pixels_per_rank(p)
        if w == 1 and not self.force_collectives:
            return x.permute(1, 0, 2, 3).reshape(fl * b * p, c).contiguous()
        send = kernels.shard_pack(x.contiguous(), self._buf("f2p_send", (w, fl, b, pp, c), x), b, fl, p, w, pp)
        return self.exchange(send, "f2p_recv").reshape(w * fl * b * pp, c)

    def pixel_result_buffer(self, like, batch, pixels, channels):
        """The send buffer of the return trip, [W, Fl, B, PP, C] viewed as rows (frame, batch, pixel): the attention kernel
        writes its result straight into it."""
        pp = self.pixels_per_rank(pixels)
        return self._buf("p2f_send", (self.world, self.local, batch, pp, channels), like).reshape(-1, channels)

    def pixels_to_frames(self, y, batch, pixels):
        """y [F_total * B * PP, C] (rows (frame, batch, pixel), e.g. ``pixel_result_buffer``) -> [B, Fl, P, C] of this rank's frames."""
        from . import kernels
        c = y.shape[-1]
        w, fl, pp = self.world, self.local, self.pixels_per_rank(pixels)
        if w == 1 and not self.force_collectives:
            return y.reshape(fl, batch, pixels, c).permute(1, 0, 2, 3).contiguous()
        recv = self.exchange(y.reshape(w, fl, batch, pp, c), "p2f_recv")
        out = torch.empty((batch, fl, pixels, c), dtype=y.dtype, device=y.device)
        return kernels.shard_pack(recv, out, batch, fl, pixels, w, pp, unpack=True)
