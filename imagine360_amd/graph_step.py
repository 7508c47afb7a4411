"""One denoising step captured in a hipGraph (torch.cuda.CUDAGraph on ROCm).

A cfg2 step is ~3000 kernel launches; issued from Python that is 0.4-0.6 s of host time, i.e. the loop is
launch-bound on a slow host.  Everything in the step is static-shaped and sync-free, so it is captured once and
replayed: per step the host only refreshes four tiny device buffers (timestep, (guidance, cx, cv), the 7 WarpAttn
coins drawn from Python's RNG in the reference's order) and calls replay().  The per-step IP-adapter noise is drawn
inside the graph from torch's device generator (graph-safe philox state), like the reference's GPU path.
"""
import random

import torch


class _PinnedUploads:
    """Small host -> device parameter uploads that do not stall the host: a ring of pinned staging slots, one event per
    slot (a copy from pageable memory is a memcpy + stream synchronise in PyTorch, i.e. one full host-device sync per
    upload per step)."""

    def __init__(self, like, slots=8):
        self.slots = [torch.empty(like.shape, dtype=like.dtype, pin_memory=True) for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def upload(self, dst, values):
        k = self.i % len(self.slots)
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy that last used this slot has been consumed
        self.slots[k].copy_(torch.as_tensor(values, dtype=dst.dtype))
        dst.copy_(self.slots[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


class GraphedDenoiseStep:
    def __init__(self, mv, scheduler, inputs, cameras, pano_latent, pers_latent, guidance, use_fps=True, warmup=2, cfg_pair=None):
        """``inputs``: the keyword tensors of MultiViewBaseModel.forward (CFG-batched, resident on the GPU);
        ``pano_latent`` [1,4,F,H,W] / ``pers_latent`` [1,m,4,F,h,w]: initial noisy latents.
        Frame-sharded models (``mv.set_frame_shard``) capture their all-to-alls with the step: the exchange buffers are
        pre-sized and cached (dist.FrameShard), RCCL collectives are stream operations.  ``cfg_pair``: process group of the
        two ranks holding the two CFG halves of the same frames (dist.cfg_frame_layout) -- their predictions are exchanged
        inside the captured step before the CFG combine."""
        self.mv, self.sch, self.inp, self.cams, self.g = mv, scheduler, inputs, cameras, float(guidance)
        self.cfg_pair = cfg_pair
        dev = pano_latent.device
        # private copies: the caller's tensors may alias the model-input buffers the body writes into
        init_pano, init_pers = pano_latent.clone(), pers_latent.clone()
        self.pano_lat = init_pano.clone()
        self.pers_lat = init_pers.clone()
        self.timestep = torch.zeros(1, dtype=torch.int64, device=dev)
        self.coef = torch.zeros(3, dtype=torch.float32, device=dev)
        self.use_fps = use_fps
        self.graph = None
        self._up_t, self._up_c = _PinnedUploads(self.timestep), _PinnedUploads(self.coef)
        # Building the graph must not consume randomness: the warm-up steps draw device noise (the per-step IP-adapter
        # noise) and allocating the coin buffer used to take 7 Python draws, which shifted the streams of the default
        # (graphed) pipeline relative to the eager one and to the reference for the same seeds.
        py_state, cuda_state = random.getstate(), torch.cuda.get_rng_state(dev)
        mv.draw_coins(dev)                       # allocates the device coin buffer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        #  the warm-up issues the step the way the capture will: with the panorama branch on the model's side stream when
        # dual_stream is set, so that stream's allocator pool and GEMM workspaces exist before the capture begins
        was_eager = getattr(mv, "dual_stream_eager", False)
        mv.dual_stream_eager = bool(getattr(mv, "dual_stream", False))
        try:
            with torch.cuda.stream(side):            # eager warm-up off the default stream (caches, allocator)
                self._upload(scheduler._timesteps_host[0], draw=False)
                for _ in range(warmup):
                    self._body()
        finally:
            mv.dual_stream_eager = was_eager
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        mv.coins_preloaded = True
        try:
            # thread-local capture mode: a RCCL watchdog / other host thread touching the runtime must not abort the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._body()
        finally:
            mv.coins_preloaded = False
        # the eager warm-up steps advanced the latents; capture itself executes nothing: start from the given ones
        self.pano_lat.copy_(init_pano)
        self.pers_lat.copy_(init_pers)
        random.setstate(py_state)
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(cuda_state, dev)

    def _body(self):
        inp = self.inp
        inp["pano_latent"][:, :4] = self.pano_lat
        inp["latents"][:, :, :4] = self.pers_lat
        was = self.mv.coins_preloaded
        self.mv.coins_preloaded = True           # coins are uploaded by _upload, never drawn inside the body
        try:
            pred_pers, pred_pano = self.mv(
                latents=inp["latents"], pano_latent=inp["pano_latent"], timestep=self.timestep,
                prompt_embd=inp["prompt_embd"], pano_prompt_embd=inp["pano_prompt_embd"], cameras=self.cams,
                use_fps_condition=self.use_fps, use_ip_plus_cross_attention=True,
                fps_tensor_pano=inp["fps_tensor_pano"], fps_tensor_pers=inp["fps_tensor_pers"],
                reference_images_clip_feat_pano=inp["reference_images_clip_feat_pano"],
                reference_images_clip_feat_pers=inp["reference_images_clip_feat_pers"],
                relative_position_tensor=inp["relative_position_tensor"], pitchs_tensor=inp["pitchs_tensor"])
        finally:
            self.mv.coins_preloaded = was
        if self.cfg_pair is not None:
            from .dist import exchange_cfg_halves
            pred_pano, pred_pers = exchange_cfg_halves(pred_pano, self.cfg_pair), exchange_cfg_halves(pred_pers, self.cfg_pair)
        self.pred_pano, self.pred_pers = pred_pano, pred_pers          # static graph-pool tensors (inspection / tests)
        ldt = self.pano_lat.dtype            # latents may be kept in another 16-bit type than the model (the reference promotes)
        pred_pano, pred_pers = pred_pano.to(ldt), pred_pers.to(ldt)
        new_pano = self.sch.fused_cfg_step(pred_pano[0:1], pred_pano[1:2], self.g, None, self.pano_lat, coef_dev=self.coef)
        new_pers = self.sch.fused_cfg_step(pred_pers[0:1], pred_pers[1:2], self.g, None, self.pers_lat, coef_dev=self.coef)
        self.pano_lat.copy_(new_pano)
        self.pers_lat.copy_(new_pers)

    def _upload(self, t_host, draw=True):
        cx, cv = self.sch.coefficients(t_host)
        self._up_t.upload(self.timestep, [int(t_host)])
        self._up_c.upload(self.coef, [self.g, cx, cv])
        if draw:
            self.mv.draw_coins(self.timestep.device)

    def step(self, t_host):
        """Advance the latents by one DDIM step at (host int) timestep ``t_host``."""
        self._upload(t_host)
        self.graph.replay()
        return self.pano_lat, self.pers_lat
