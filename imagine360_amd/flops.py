"""Algorithmic FLOPs of one CFG-batched dual-branch denoising step (SURVEY.md section 8d formula):
GEMM = 2 M N K, attention = 4 Nq Nk C; norms, activations, mask building and the hoistable IP-adapter
conditioning are excluded; cross-attention K/V projections are counted per frame like the reference
computes them.  Used by bench.py for the whole-step TFLOP/s figure and to scale the CPU baseline."""


def step_flops(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), views=20, cfg_batch=2,
               block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, ctx_tokens=141, ctx_dim=1024,
               in_ch=9, out_ch=4, breakdown=False):
    F = frames
    boc = block_out_channels
    parts = {}

    def add(name, v):
        parts[name] = parts.get(name, 0.0) + v

    def branch(tag, B, H, W, pad):
        """B = batch of videos of the branch; pad = circular pad of pano resnet convs (0 for pers)."""
        def conv(cin, cout, h, w, k=3):
            add(tag + " conv", 2.0 * cin * cout * k * k * h * w * B * F)

        def resnet(cin, cout, h, w):
            conv(cin, cout, h, w + 2 * pad)          # conv1 runs on the padded width in the pano branch
            conv(cout, cout, h, w + 2 * pad)
            if cin != cout:
                conv(cin, cout, h, w + 2 * pad, k=1)

        def spatial(c, h, w):
            n = h * w
            add(tag + " spatial GEMM", B * F * n * (2.0 * c * c * 2 + 2.0 * c * c * 4 + 2.0 * c * c * 2 + 2.0 * c * 8 * c + 2.0 * 4 * c * c))
            add("cross-attn+KV", B * F * ctx_tokens * 2.0 * ctx_dim * c * 2 + 4.0 * n * ctx_tokens * c * B * F)
            add(tag + " self-attn", 4.0 * n * n * c * B * F)

        def temporal(c, h, w):
            n = h * w
            add(tag + " temporal GEMM", B * n * F * (2.0 * c * c * 2 + 2 * (2.0 * c * c * 4) + 2.0 * c * 8 * c + 2.0 * 4 * c * c))
            add("temporal attn", 2 * 4.0 * F * F * c * B * n)

        h, w = H, W
        conv(in_ch, boc[0], h, w + (2 if pad else 0))
        skips = [boc[0]]
        cin = boc[0]
        nlev = len(boc)
        for i, c in enumerate(boc):
            for j in range(layers_per_block):
                resnet(cin, c, h, w)
                cin = c
                if i < nlev - 1:
                    spatial(c, h, w)
                    temporal(c, h, w)
                skips.append(c)
            if i < nlev - 1:
                conv(c, c, h // 2, w // 2 + (1 if pad else 0) * 2)
                h, w = h // 2, w // 2
                skips.append(c)
        c = boc[-1]
        resnet(c, c, h, w)
        spatial(c, h, w)
        temporal(c, h, w)
        resnet(c, c, h, w)
        rc = list(reversed(boc))
        for i, c in enumerate(rc):
            for j in range(layers_per_block + 1):
                resnet(cin + skips.pop(), c, h, w)
                cin = c
                if i > 0:
                    spatial(c, h, w)
                    temporal(c, h, w)
            if i < nlev - 1:
                h, w = h * 2, w * 2
                conv(c, c, h, w + (2 if pad else 0) * 2)
        conv(boc[0], out_ch, h, w + (2 if pad else 0))

    branch("pers", cfg_batch * views, pers_hw[0], pers_hw[1], 0)
    branch("pano", cfg_batch, pano_hw[0], pano_hw[1], 2)
    # WarpAttn at enc L1..L3, mid L3, dec L3, L2, L1
    lv = lambda k: ((pano_hw[0] >> k) * (pano_hw[1] >> k), views * (pers_hw[0] >> k) * (pers_hw[1] >> k))
    for c, k in ((boc[0], 1), (boc[1], 2), (boc[2], 3), (boc[3], 3), (boc[3], 3), (boc[2], 2), (boc[1], 1)):
        ne, npx = lv(k)
        add("WarpAttn attn", 2 * 4.0 * ne * npx * c * cfg_batch * F)
        add("WarpAttn GEMM", (ne + npx) * cfg_batch * F * (2.0 * c * c * 4 + 2.0 * c * 8 * c + 2.0 * 4 * c * c))
    total = sum(parts.values())
    return (total, parts) if breakdown else total


CONFIGS = {
    "cfg1": dict(frames=8, pano_hw=(32, 64), pers_hw=(16, 16)),
    "cfg2": dict(frames=16, pano_hw=(64, 128), pers_hw=(32, 32)),
    "cfg4": dict(frames=48, pano_hw=(64, 128), pers_hw=(32, 32)),
    "cfg5": dict(frames=16, pano_hw=(128, 256), pers_hw=(64, 64)),
}

if __name__ == "__main__":
    for k, v in CONFIGS.items():
        t, p = step_flops(breakdown=True, **v)
        print(k, f"{t / 1e12:.1f} TF", {a: round(b / 1e12, 1) for a, b in sorted(p.items())})
