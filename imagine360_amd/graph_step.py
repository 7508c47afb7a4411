"""One denoising step captured in a hipGraph (torch.cuda.CUDAGraph on ROCm).

A cfg2 step is ~3000 kernel launches; issued from Python that is 0.4-0.6 s of host time, i.e. the loop is
launch-bound on a slow host.  Everything in the step is static-shaped and sync-free, so it is captured once and
replayed: per step the host only refreshes four tiny device buffers (timestep, (guidance, cx, cv), the 7 WarpAttn
coins drawn from Python's RNG in the reference's order) and calls replay().  The per-step IP-adapter noise is drawn
inside the graph from torch's device generator (graph-safe philox state), like the reference's GPU path.
"""
import torch


class GraphedDenoiseStep:
    def __init__(self, mv, scheduler, inputs, cameras, pano_latent, pers_latent, guidance, use_fps=True, warmup=2):
        """``inputs``: the keyword tensors of MultiViewBaseModel.forward (CFG-batched, resident on the GPU);
        ``pano_latent`` [1,4,F,H,W] / ``pers_latent`` [1,m,4,F,h,w]: initial noisy latents."""
        self.mv, self.sch, self.inp, self.cams, self.g = mv, scheduler, inputs, cameras, float(guidance)
        dev = pano_latent.device
        # private copies: the caller's tensors may alias the model-input buffers the body writes into
        init_pano, init_pers = pano_latent.clone(), pers_latent.clone()
        self.pano_lat = init_pano.clone()
        self.pers_lat = init_pers.clone()
        self.timestep = torch.zeros(1, dtype=torch.int64, device=dev)
        self.coef = torch.zeros(3, dtype=torch.float32, device=dev)
        self.use_fps = use_fps
        self.graph = None
        mv.draw_coins(dev)                       # allocates the device coin buffer (consumes 7 Python draws)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # eager warm-up off the default stream (caches, allocator)
            self._upload(scheduler._timesteps_host[0], draw=False)
            for _ in range(warmup):
                self._body()
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
        self.pred_pano, self.pred_pers = pred_pano, pred_pers          # static graph-pool tensors (inspection / tests)
        new_pano = self.sch.fused_cfg_step(pred_pano[0:1], pred_pano[1:2], self.g, None, self.pano_lat, coef_dev=self.coef)
        new_pers = self.sch.fused_cfg_step(pred_pers[0:1], pred_pers[1:2], self.g, None, self.pers_lat, coef_dev=self.coef)
        self.pano_lat.copy_(new_pano)
        self.pers_lat.copy_(new_pers)

    def _upload(self, t_host, draw=True):
        cx, cv = self.sch.coefficients(t_host)
        # fresh pageable host tensors (staged at call time): safe when the host runs steps ahead of the GPU
        self.timestep.copy_(torch.tensor([int(t_host)], dtype=torch.int64))
        self.coef.copy_(torch.tensor([self.g, cx, cv], dtype=torch.float32))
        if draw:
            self.mv.draw_coins(self.timestep.device)

    def step(self, t_host):
        """Advance the latents by one DDIM step at (host int) timestep ``t_host``."""
        self._upload(t_host)
        self.graph.replay()
        return self.pano_lat, self.pers_lat
