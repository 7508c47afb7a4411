"""Multi-GPU partitioning of the denoising path over RCCL / xGMI (one process per GPU).

The reference has no distributed code; the path shards along axes on which the
reference's arithmetic is independent:

  * samples (BASELINE config 3): one prompt per GPU, weights replicated, no per-step traffic; the only
    collective is ONE all-gather of the final panorama latents at the latent boundary
    (``gather_latents``), 1 MB per rank at 16x512x1024;
  * frames (BASELINE configs 4/5): every op is per (batch, frame) image -- conv, GroupNorm, spatial and
    cross-view attention -- except the motion modules' attention over the frame axis.  ``FrameShard``
    gives each rank a contiguous chunk of frames and turns frame-sharded tokens into pixel-sharded tokens
    (and back) with one all-to-all per temporal attention, Ulysses style; xGMI is point-to-point, so an
    all-to-all of activation slabs uses all 7 links at once where a ring would be bound by one.

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
    parts = [torch.empty_like(pred) for _ in range(2)]
    dist.all_gather(parts, pred.contiguous(), group=pair_group)
    return torch.cat(parts)


def cfg_frame_layout(frames, world=None, rank=None):
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
    return my_half, FrameShard(frames, group=groups[my_half], rank=rank % half, world=half), pairs[rank % half]


class FrameShard:
    """Contiguous frame chunks across the ranks of ``group`` (frames % world == 0)."""

    def __init__(self, total_frames, group=None, rank=None, world=None):
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
        if self.world == 1:
            return x
        parts = [torch.empty_like(x) for _ in range(self.world)]
        dist.all_gather(parts, x.contiguous(), group=self.group)
        return torch.cat(parts, dim=dim)

    # tokens are [B, F_local, P, C] (frame-sharded)  <->  [B, F_total, P / world, C] (pixel-sharded)
    def frames_to_pixels(self, x):
        b, fl, p, c = x.shape
        w = self.world
        if w == 1:
            return x
        pp = -(-p // w)                                   # pixels per rank, last rank zero-padded
        if pp * w != p:
            x = torch.nn.functional.pad(x, (0, 0, 0, pp * w - p))
        send = x.reshape(b, fl, w, pp, c).permute(2, 0, 1, 3, 4).contiguous()          # [w, b, fl, pp, c]
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return recv.permute(1, 0, 2, 3, 4).reshape(b, w * fl, pp, c)                    # frames ordered by source rank

    def pixels_to_frames(self, y, pixels):
        b, f, pp, c = y.shape
        w, fl = self.world, self.local
        if w == 1:
            return y
        send = y.reshape(b, w, fl, pp, c).permute(1, 0, 2, 3, 4).contiguous()           # [w, b, fl, pp, c]
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        out = recv.permute(1, 2, 0, 3, 4).reshape(b, fl, w * pp, c)
        return out[:, :, :pixels]
