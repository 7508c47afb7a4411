/* libim360_kernels.so -- C ABI of the MI355X (gfx950) kernels behind imagine360_amd.
 *
 * The reference (3DTopia/Imagine360) has no FFI: its arithmetic for this path lives in third-party
 * wheels (torch ATen / cuDNN, xformers).  Each entry point below therefore cites the reference call
 * site(s) whose kernel it replaces (paths relative to the reference repository).  Conventions:
 *   - plain device pointers + int64 sizes/strides (in ELEMENTS), no torch types;
 *   - dtype: 0 = bfloat16, 1 = float16 (16-bit storage, fp32 accumulation);
 *   - every function only enqueues work on `stream` (a hipStream_t); no allocation, no sync;
 *   - returns 0, or a negative code with a message available from im360_last_error().
 * Activations are channels-last: images [N, H, W, C] where N = (batch x frame), tokens [B, N, C].
 */
#ifndef IM360_KERNELS_H
#define IM360_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 5 since round 6 (3: packed-bias block maps of im360_attn_fwd; 4: im360_groupnorm_partial_pad / im360_groupnorm_apply_partials added;
 * 5: im360_conv_ksplit_plan / im360_conv_fwd_ksplit added).
 * 2 since round 5: im360_conv_fwd / im360_linear_fwd take a trailing gn_partial pointer and the table rows of
 * im360_linear_ln_fwd include c2 (both introduced in round 4 under version 1).  imagine360_amd/kernels.py refuses a mismatch. */
int im360_abi_version(void);
/* bit 0: ablation build (`make ablate`): the rejected A/B kernel variants behind the attn_dbg / attn_hl / attn_hg / attn_ds knobs exist */
int im360_build_flags(void);
const char* im360_last_error(void);

/* softmax(Q K^T * scale + bias) V, head dim D in {32, 64}; head h of batch b lives at
 * base + b*bs + row*rs + h*D.  bias (optional) is ONE [Nq, Nk] matrix shared by all (b, h).
 * accumulate != 0: out += out_scale * result (second KV set of the IP cross attention).
 * bias_alt / bias_sel (optional): a second [Nq, Nk] bias and a device int32; the kernel uses bias_alt when
 *   *bias_sel != 0 (WarpAttn's normal / antipodal mask choice made on the device, so a step can be graph-replayed).
 * kv_group: K/V batch index = query batch index / kv_group (one context per video, F frames of queries).
 * dtype: 0 bf16, 1 fp16; + 256 (D = 32 only, Nk % 8 == 0): bias and bias_alt are fp16 matrices already multiplied by
 *   log2(e) (im360_attn_pack_bias) and are added to the scores by the matrix pipe instead of the vector ALU.
 * bias_blocks / bias_blocks_alt (optional, packed bias only; ABI version 3): block maps of bias / bias_alt -- uint32 words
 *   [ceil(Nq / 32)][blocks_rs], bit (h % 32) of word (h / 32) of row r clear = every entry of the 32 x 32 block (queries 32 r ..,
 *   keys 32 h ..) of the packed matrix is ZERO; the kernel skips such a block's mask loads and mask MFMAs (identical results: they
 *   would add zeros).  The caller makes the common value of a mask zero by subtracting it from the whole matrix -- softmax is
 *   invariant under a per-row constant (WarpAttn: ~ 80 - 90 % of the blocks are background).
 * Replaces: xformers.ops.memory_efficient_attention / F.scaled_dot_product_attention at
 *   diffusers/models/attention_processor.py:1264, 1351, 641 (spatial self / cross attention),
 *   animatediff/models/attention.py:113-148 (text + IP cross attention),
 *   src/modules/transformer.py:72 (WarpAttn cross-view attention with additive mask). */
int im360_attn_fwd(const void* q, const void* k, const void* v, const void* bias, void* out,
                   int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                   int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                   int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, int64_t bias_rs,
                   int64_t kv_group, float scale, float out_scale, int accumulate, int dtype, void* stream,
                   const void* bias_alt, const void* bias_sel,
                   const void* bias_blocks, const void* bias_blocks_alt, int64_t blocks_rs);

/* out_f16[i] = fp16(clamp(bias[i] * log2(e), -60000, 60000)) for the n elements of a bias matrix of dtype 0 / 1: the packed
 * form accepted by im360_attn_fwd with dtype + 256.  The packed entries must be FINITE (the kernel adds them with an
 * identity-slice MFMA, where 0 * inf would poison the whole score block): this function guarantees it by clamping, so -inf
 * masks are usable (weight exactly 0).  Done once per (resolution, camera rig) -- WarpAttn's masks are cached. */
int im360_attn_pack_bias(const void* bias, void* out_f16, int64_t n, int dtype, void* stream);

/* Two key / value sets for the same queries in one launch (head dim 64, no bias):
 *   out = out_scale * softmax(q k^T scale) v + out_scale2 * softmax(q k2^T scale) v2
 * -- the text tokens and the IP-adapter tokens of the spatial cross attention, which the reference evaluates as two
 * attention calls and an add.  Strides / kv_group as in im360_attn_fwd.  With 65..96 + 33..64 keys, Nq % 32 == 0 and 16-byte
 * aligned output rows (the model's 77 + 64 tokens) the launch keeps BOTH sets resident in LDS per (video, head) and streams the
 * query blocks of all the video's frames past them (knob 11); other shapes take the generic two-pass kernel.
 * Replaces: IPCrossAttention.forward, animatediff/models/attention.py:113-148. */
int im360_attn_fwd2(const void* q, const void* k, const void* v, const void* k2, const void* v2, void* out,
                    int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Nk2, int64_t D,
                    int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs,
                    int64_t k2_bs, int64_t k2_rs, int64_t v2_bs, int64_t v2_rs, int64_t o_bs, int64_t o_rs,
                    int64_t kv_group, float scale, float out_scale, float out_scale2, int dtype, void* stream);

/* Temporal self-attention over F <= 64 frames on token-major activations [B, F, P, heads*d]
 * (q, k, v are three views with common strides, e.g. slices of a fused QKV projection).
 * Replaces: VersatileAttention.forward -> Attention._attention (baddbmm/softmax/bmm) and its
 *   '(b f) d c <-> (b d) f c' rearranges, animatediff/models/motion_module.py:343-429,
 *   diffusers/models/attention_processor.py:562-591. */
int im360_temporal_attn_fwd(const void* q, const void* k, const void* v, void* out,
                            int64_t B, int64_t F, int64_t P, int64_t heads, int64_t d,
                            int64_t qkv_fs, int64_t qkv_ps, int64_t qkv_bs,
                            int64_t o_fs, int64_t o_ps, int64_t o_bs,
                            float scale, int dtype, void* stream);

/* Frame-chunk sharding (one process per GPU, imagine360_amd.dist.FrameShard): frame-sharded tokens [B, Fl, P, C] (16-bit,
 * C % 8 == 0) <-> the buffer of the frame <-> pixel all-to-all, [W][Fl][B][PP][C] (W ranks, PP = ceil(P / W), the tail of
 * the last rank zero-filled); dir 0 packs, 1 unpacks.  The receive side of the exchange is read IN PLACE by
 * im360_temporal_attn_fwd through its frame / batch / pixel strides (frame stride B * PP * C), which also writes the
 * return trip's send buffer directly.  Pre-sized caller-owned buffers: the exchange can be captured in a hipGraph.
 * Replaces: nothing in the reference (single process); it surrounds VersatileAttention, motion_module.py:343-429. */
int im360_shard_pack(const void* src, void* dst, int64_t B, int64_t Fl, int64_t P, int64_t C, int64_t W, int64_t PP,
                     int dir, void* stream);

/* GroupNorm statistics -> per-(image, channel) fp32 scale/shift such that GN(x) = x*scale + shift.
 * pad > 0: statistics of the circularly W-padded tensor (the pano branch pads before norm1,
 * src/models/MVGenModel.py:277-278).  partial: fp32 workspace of N * S * 2 * C floats,
 * S = im360_gn_num_slabs(N, H, W).
 * Replaces: nn.GroupNorm / InflatedGroupNorm statistics, animatediff/models/resnet.py:9-17, 224, 236;
 *   animatediff/models/attention.py:262; motion_module.py:169; unet.py (conv_norm_out); VAE norms. */
int64_t im360_gn_num_slabs(int64_t N, int64_t H, int64_t W);
int im360_groupnorm_stats(const void* x, const void* gamma, const void* beta, void* partial,
                          void* scale, void* shift, int64_t N, int64_t H, int64_t W, int64_t C,
                          int64_t G, int64_t pad, float eps, int dtype, void* stream);

/* y[N, H, W + 2 pad, C] = act(x * scale + shift) with circular W addressing; act 0 = none, 1 = SiLU.
 * Replaces: the normalise + F.silu pass (resnet.py:224-225, 236-243) fused with pad_pano
 *   (src/utils/pano.py:75-95). */
/* Partial sums only: partial fp32 [N][S][2][C], S = im360_gn_num_slabs(N, H, W) (no pad weighting). */
int im360_groupnorm_partial(const void* x, void* partial, int64_t N, int64_t H, int64_t W, int64_t C, int dtype, void* stream);
/* scale / shift [N, C1 + C2] from partial sums of one or two channel ranges: [0, C1) from pa ([N][Sa][2][C1]), [C1, C1 + C2)
 * from pb ([N][Sb][2][C2], null with C2 = 0).  Either may come from im360_groupnorm_partial or from the epilogue of the
 * kernel that produced that tensor (gn_partial of im360_conv_fwd / im360_linear_fwd).  pixels = H * W per image. */
int im360_groupnorm_finalize(const void* pa, int64_t Sa, int64_t C1, const void* pb, int64_t Sb, int64_t C2, const void* gamma,
                             const void* beta, void* scale, void* shift, int64_t N, int64_t G, int64_t pixels, float eps,
                             int dtype, void* stream);
int im360_groupnorm_apply(const void* x, const void* scale, const void* shift, void* y,
                          int64_t N, int64_t H, int64_t W, int64_t C, int64_t pad, int act,
                          int dtype, void* stream);

/* Round 6 (ABI version 4) -- the normalisation straight from PARTIAL SUMS, one launch: no finalize kernel, no scale / shift tensors.
 * im360_groupnorm_partial_pad: im360_groupnorm_partial with the circular-pad weighting of im360_groupnorm_stats (the `pad` wrapped
 *   columns count twice; src/models/MVGenModel.py:277-278).
 * im360_groupnorm_apply_partials: y [N, H, W + 2 pad, C1 + C2] = act(GroupNorm([xa | xb])); the statistics of channels [0, C1)
 *   come from pa ([N][Sa][2][C1]), those of [C1, C1 + C2) from pb ([N][Sb][2][C2]); xb / pb null with C2 = 0.  Every workgroup
 *   rebuilds scale / shift of its image from the partial sums (the reduction order, and the bits, of im360_groupnorm_finalize).
 * Replaces: InflatedGroupNorm / nn.GroupNorm (+ F.silu, + pad_pano, + torch.cat([x, skip])), animatediff/models/resnet.py:9-17,
 *   221-243; animatediff/models/attention.py:206, 262; motion_module.py:128, 169; src/models/MVGenModel.py:407-437. */
int im360_groupnorm_partial_pad(const void* x, void* partial, int64_t N, int64_t H, int64_t W, int64_t C, int64_t pad, int dtype,
                                void* stream);
int im360_groupnorm_apply_partials(const void* xa, const void* xb, const void* pa, int64_t Sa, const void* pb, int64_t Sb,
                                   const void* gamma, const void* beta, void* y, int64_t N, int64_t H, int64_t W, int64_t C1,
                                   int64_t C2, int64_t G, int64_t pad, float eps, int act, int dtype, void* stream);

/* The two GroupNorm passes on the channel concatenation [xa | xb] (xa [N, H, W, C1], xb [N, H, W, C2]) WITHOUT
 * materialising it: gamma / beta / scale / shift / y span C1 + C2 channels, partial holds N * S * 2 * (C1 + C2) floats.
 * Replaces: torch.cat([x, skip], dim=1) -> ResnetBlock3D.norm1 in the decoder, src/models/MVGenModel.py:407, 415, 431,
 *   437 -> animatediff/models/resnet.py:221-225. */
int im360_groupnorm_stats_cat(const void* xa, const void* xb, const void* gamma, const void* beta, void* partial,
                              void* scale, void* shift, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2,
                              int64_t G, int64_t pad, float eps, int dtype, void* stream);
int im360_groupnorm_apply_cat(const void* xa, const void* xb, const void* scale, const void* shift, void* y,
                              int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t pad, int act,
                              int dtype, void* stream);

/* The whole GroupNorm (+ SiLU, + circular W pad; on x or on the never-materialised concatenation [xa | xb], xb may be null
 * with C2 = 0) in ONE launch: per-slab partial sums, a per-image arrival counter, every workgroup reduces its image's
 * partials in the fixed order of the two-kernel path (same bits) and normalises its slab -- whose second read comes out of
 * the L2 / Infinity Cache microseconds after the first: 2 HBM passes instead of 3.  partial: N * S * 2 * (C1 + C2) floats
 * with S = im360_gn_num_slabs(N, H, W); counter: N int32 (zeroed by the call, on the stream).
 * Replaces: InflatedGroupNorm / nn.GroupNorm + F.silu + pad_pano, animatediff/models/resnet.py:9-17, 221-243;
 *   attention.py:262; motion_module.py:169; unet.py conv_norm_out; src/utils/pano.py:75-95. */
int im360_groupnorm_fused(const void* xa, const void* xb, const void* gamma, const void* beta, void* partial, void* counter,
                          void* y, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t G, int64_t pad,
                          float eps, int act, int dtype, void* stream);

/* 3x3 (ntaps = 9) or 1x1 (ntaps = 1) convolution, implicit GEMM on MFMA.  x [N, Hin, Win, Cin],
 * y [N, Hout, Wout, Cout], w_packed from im360_pack_conv_weight.  stride 1|2; up: input is
 * nearest-upsampled x2 on the fly; wrap: circular W addressing; x_off / y_off: column / row offset of the
 * taps (pre-padded input; the VAE downsampler's (0,1,0,1) padding is x_off = y_off = 1).  Epilogue: + bias[Cout] + temb[n / imgs_per_temb][Cout] + res[N, Hout, Wout, Cout].
 * Replaces: InflatedConv3d / nn.Conv2d at animatediff/models/resnet.py:19-27, 84, 128, 183, 205, 218,
 *   227-251; unet.py:134-137, 358; pad_pano/unpad_pano around them (MVGenModel.py:138-143, 276-281,
 *   305-314, 449-456, 474-478); F.interpolate nearest (resnet.py:104); the VAE convs. */
int im360_conv_fwd(const void* x, const void* w_packed, const void* bias, const void* temb,
                   const void* res, void* y,
                   int64_t N, int64_t Hin, int64_t Win, int64_t Cin,
                   int64_t Hout, int64_t Wout, int64_t Cout, int64_t ntaps,
                   int64_t stride, int64_t up, int64_t wrap, int64_t x_off, int64_t y_off,
                   int64_t imgs_per_temb, int dtype, void* stream, void* gn_partial);
/* gn_partial (optional; last argument of im360_conv_fwd / im360_linear_fwd): the epilogue also writes per 256-pixel tile and
 * output channel (sum, sum of squares) of the stored values, fp32 [M / 256][2][Cout] -- GroupNorm partial sums with one slab
 * per tile, consumed by im360_groupnorm_finalize in place of a statistics pass over the tensor
 * (animatediff/models/resnet.py:221-243 is norm -> SiLU -> conv twice).  Only launches for which im360_conv_gn_slabs
 * returns S > 0 (slabs per image) can do it. */
int64_t im360_conv_gn_slabs(int64_t N, int64_t Hout, int64_t Wout, int64_t Cin, int64_t Cout, int64_t ntaps);

/* K-split of a 3x3 convolution launch (ABI version 5): every 256 x 320 output tile is computed by `parts` workgroups, each over a
 * contiguous range of the 64-channel chunks of K; parts 0 .. parts - 2 park their fp32 accumulators in ks_ws, the last one (dispatched
 * behind them) adds them in part order and runs the usual epilogue.  For launches whose tile count is not a whole number of rounds
 * of the chip's 256 CUs (perspective level 2 of cfg2: 640 tiles = 2.5 rounds) or below one round (level 3: 160 tiles).
 * im360_conv_ksplit_plan: parts this launch would run in (1 = none: call im360_conv_fwd; knob conv_ksplit 0 = never).
 * im360_conv_fwd_ksplit: im360_conv_fwd + the scratch: ks_ws >= tiles x (parts - 1) x 327 680 bytes (tiles = ceil(N Hout Wout / 256)
 * x Cout / 320), ks_cnt = `tiles` int32 counters, ZERO on entry and zero again on return.  Deterministic; not bit-identical to the
 * unsplit launch (another order of the fp32 partial sums).  Same call sites as im360_conv_fwd (animatediff/models/resnet.py:19-27). */
int64_t im360_conv_ksplit_plan(int64_t N, int64_t Hout, int64_t Wout, int64_t Cin, int64_t Cout, int64_t ntaps,
                               int64_t up, int64_t wrap, int64_t gn_stats);
int im360_conv_fwd_ksplit(const void* x, const void* w_packed, const void* bias, const void* temb,
                          const void* res, void* y,
                          int64_t N, int64_t Hin, int64_t Win, int64_t Cin,
                          int64_t Hout, int64_t Wout, int64_t Cout, int64_t ntaps,
                          int64_t stride, int64_t up, int64_t wrap, int64_t x_off, int64_t y_off,
                          int64_t imgs_per_temb, int dtype, void* stream, void* gn_partial,
                          void* ks_ws, int64_t ks_ws_bytes, void* ks_cnt, int64_t ks_cnt_n);

/* 1x1 convolution of the channel concatenation [xa | xb] without materialising it: the K loop reads channels [0, C1)
 * from xa and [C1, C1 + C2) from xb (C1, C2 multiples of 64); w_packed [CoutPad][1][C1 + C2]; + bias + res.
 * Replaces: ResnetBlock3D.conv_shortcut on torch.cat([x, skip]), animatediff/models/resnet.py:248-249 after
 *   src/models/MVGenModel.py:407, 415, 431, 437. */
int im360_conv1x1_cat_fwd(const void* xa, const void* xb, const void* w_packed, const void* bias, const void* res,
                          void* y, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t Cout,
                          int dtype, void* stream);

/* Nearest-x2 upsample followed by conv3x3 (pad 1; wrap: circular along W), computed as four 2 x 2 convolutions of the
 * LOW-resolution input, one per output parity: after the upsample output pixel (2y + py, 2x + px) sees only the 2 x 2 source
 * pixels (y + r + py - 1, x + c + px - 1), so the nine taps collapse into four pre-summed ones -- 4 / 9 of the MACs.
 * x [N, Hin, Win, Cin] (Cin % 64 == 0) -> y [N, 2 Hin, 2 Win, Cout]; w4 = four im360_pack_conv_weight outputs (taps = 4,
 * [CoutPad][4][Cin] each) back to back in parity order (0,0) (0,1) (1,0) (1,1).  The pre-summed weights are rounded to
 * 16 bits once more than the reference's (a stated deviation).
 * Replaces: Upsample3D = F.interpolate(scale 2, nearest) + InflatedConv3d (animatediff/models/resnet.py:71-114), incl. the
 *   pano branch's pad_pano -> upsample -> unpad_pano sandwich (MVGenModel.py). */
int im360_conv_up2_fwd(const void* x, const void* w4, const void* bias, void* y, int64_t N, int64_t Hin,
                       int64_t Win, int64_t Cin, int64_t Cout, int64_t wrap, int dtype, void* stream);

/* PyTorch conv weight [Cout, Cin, kh, kw] -> [CoutPad (mult. of 128)][kh*kw][CinPad (mult. of 32)]. */
int im360_pack_conv_weight(const void* w, void* out, int64_t Cout, int64_t Cin, int64_t taps,
                           int64_t CoutPad, int64_t CinPad, int dtype, void* stream);

/* x [rows, W, C] -> y [rows, W + 2 pad, C], circular along W.
 * Replaces: pad_pano on the latent before VAE decode, pipeline_animation_inference_dual.py:813. */
int im360_circular_pad_w(const void* x, void* y, int64_t rows, int64_t W, int64_t C, int64_t pad,
                         int dtype, void* stream);

/* x [N, H, W] (W last, elements of esize = 1 / 2 / 4 / 8 bytes) -> y [N, H + top + bottom, W + left + right], circular in
 * both axes = torch.nn.functional.pad(x, (left, right, top, bottom), mode="circular"); each pad <= the size of its axis.
 * Replaces: the 360-degree close-loop patch of the super-resolution stage -- padding_pano / pad_pano on NCHW video and
 *   latent tensors and the circular pad_to_fit, sr/video_to_video_model.py:16-29, 99, 160-162; src/utils/pano.py:75-95. */
int im360_circular_pad_hw(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t left, int64_t right,
                          int64_t top, int64_t bottom, int64_t esize, void* stream);

/* out = cx * x + cv * (uncond + guidance * (cond - uncond)); cx, cv = DDIM v-prediction coefficients;
 * coef_dev (optional): device float[3] = (guidance, cx, cv) overriding the scalars (hipGraph replay).
 * Replaces: the CFG combine + DDIMScheduler.step elementwise chain,
 *   pipeline_animation_inference_dual.py:791-800; diffusers/schedulers/scheduling_ddim.py:300-350. */
int im360_cfg_ddim_update(const void* uncond, const void* cond, const void* x, void* out, int64_t n,
                          float guidance, float cx, float cv, int dtype, void* stream, const void* coef_dev);

/* y[r] = LayerNorm(x[r] + pre[r % pre_period]) * gamma + beta + post[(r / post_div) % post_mod] on token rows
 * [rows, C]; pre / post are optional [*, C] tables (the WarpAttn spherical PE added before norm1, the motion
 * module's frame PE added after the norm).
 * Replaces: nn.LayerNorm at animatediff/models/attention.py:463-507, motion_module.py:249-256 (+ PE add :349-350),
 *   src/modules/transformer.py:156-165 (+ `query + query_pe`, attn_perspano.py:56,63). */
int im360_layernorm(const void* x, const void* gamma, const void* beta, const void* pre, const void* post,
                    void* y, int64_t rows, int64_t C, int64_t pre_period, int64_t post_div,
                    int64_t post_mod, float eps, int dtype, void* stream);

/* out[rows, I] = h[:, :I] * gelu(h[:, I:]) (exact erf GELU).
 * Replaces: GEGLU, diffusers/models/activations.py:93-125; src/modules/transformer.py:10-16. */
int im360_geglu(const void* h, void* out, int64_t rows, int64_t I, int dtype, void* stream);

/* GEGLU feed-forward input projection in one launch: y[m, j] = (x W_v^T + b_v)[m, j] * gelu((x W_g^T + b_g)[m, j]),
 * x [M, K] token-major, y [M, I].  w_packed / bias_packed hold the 2I rows of the projection with 32-row blocks of
 * the value half and the gate half interleaved (v0, g0, v1, g1, ...; see kernels.pack_geglu), the weight in
 * im360_pack_conv_weight's [rows][1][K] layout.  K % 64 == 0, I % 128 == 0.  The projection is rounded to 16 bits
 * before the activation, like the two-kernel path.
 * Replaces: GEGLU.forward, diffusers/models/activations.py:93-125 (Linear + chunk + F.gelu + mul). */
int im360_linear_geglu(const void* x, const void* w_packed, const void* bias_packed, void* y,
                       int64_t M, int64_t K, int64_t I, int dtype, void* stream);

/* Token-major Linear y[M, N] = x[M, K] w^T + bias (+ res) on the persistent MFMA kernel (N % 320 == 0, K % 32 == 0,
 * w_packed = im360_pack_conv_weight of the [N, K, 1, 1] view).  rowstats (optional): fp32 [M][N / 160][2], per row and
 * 160-column slice (sum, sum of squares) of the STORED 16-bit output -- the LayerNorm statistics of y's rows, taken in
 * the epilogue that writes them, for a consumer that folds the normalisation into its GEMM (the two functions below).
 * Replaces: nn.Linear (+ residual add) in front of nn.LayerNorm, animatediff/models/attention.py:264, 461-508;
 *   motion_module.py:172, 230-258; src/modules/transformer.py:156-165. */
int im360_linear_fwd(const void* x, const void* w_packed, const void* bias, const void* res, void* y, void* rowstats,
                     int64_t M, int64_t K, int64_t N, int dtype, void* stream, void* gn_partial);

/* Linear(LayerNorm(x)) with the normalisation folded into the GEMM: x [M, K] are the RAW rows, rowstats [M][rs_p][2] their
 * (sum, sum of squares) slices from im360_linear_fwd, w_packed = pack(gamma (.) W) and
 *   y[r] = rstd_r * (x[r] w^T - mu_r * c1) + (tab ? tab[(r / tab_div) % tab_mod] : c2)
 * with fp32 vectors c1[n] = sum_k w'[n][k] (of the rounded 16-bit w'), c2 = W beta + bias; tab (optional, fp32
 * [tab_mod][N], tab_div % 256 == 0): c2 + the rows added AFTER the normalisation pushed through the projection (the motion
 * module's frame positional encoding; one table row per 256-row tile; since round 4 the rows INCLUDE c2, so that a tile's
 * constant term is one vector the kernel fetches by LDS-DMA).  The LayerNorm pass over the activations and its output tensor do not exist.  Variance = E[x^2] - mu^2 in
 * fp32 from 16-bit data.  N % 320 == 0, K % 32 == 0.
 * Replaces: nn.LayerNorm -> to_q / fused to_q,k,v, animatediff/models/attention.py:470-488; motion_module.py:236-250
 *   (+ pos_encoder :349-350). */
int im360_linear_ln_fwd(const void* x, const void* w_packed, const void* c1, const void* c2, const void* rowstats,
                        int64_t rs_p, float eps, const void* tab, int64_t tab_div, int64_t tab_mod, void* y,
                        int64_t M, int64_t K, int64_t N, int dtype, void* stream);

/* GEGLU(LayerNorm(x)) in one launch: im360_linear_geglu with the normalisation folded in as above; w_packed, c1, c2 in
 * the interleaved row order of kernels.pack_geglu.
 * Replaces: nn.LayerNorm -> GEGLU, animatediff/models/attention.py:503-506; motion_module.py:255-257;
 *   src/modules/transformer.py:164-165. */
int im360_linear_geglu_ln(const void* x, const void* w_packed, const void* c1, const void* c2, const void* rowstats,
                          int64_t rs_p, float eps, void* y, int64_t M, int64_t K, int64_t I, int dtype, void* stream);

/* y[r, :] = softmax(x[r, :] * scale), fp32 arithmetic on 16-bit rows (cols and the row strides multiples of 8; in place
 * allowed).  With two launches of im360_conv_fwd as GEMMs it forms the single-head d = 512 attention of the VAE.
 * Replaces: AttentionBlock.forward's softmax(attention_scores.float()), diffusers/models/attention.py:336-364. */
int im360_softmax_rows(const void* x, void* y, int64_t rows, int64_t cols, int64_t x_rs, int64_t y_rs,
                       float scale, int dtype, void* stream);

/* out[n][m] = cv2.remap(img[n], map_x[m], map_y[m], INTER_CUBIC, borderMode=BORDER_WRAP): uint8 images [N, H, W, C]
 * (C = 1, 3, 4) on the device, float32 maps [M, h, w] in source pixel units, out uint8 [N, M, h, w, C]; wtab = the
 * 1024 x 16 int16 bicubic weight table (imagine360_amd.preprocess.cubic_weight_table).  One launch warps every frame through
 * every camera's map.  OpenCV's fixed-point arithmetic restated (parity unpinned: OpenCV is not available to check it).
 * Replaces: the cv2.remap calls of Equirec2Perspec.GetPerspective / Perspec2Equirec.GetEquirec,
 *   src/utils/pano_utils/Equirec2Perspec.py:61, Perspec2Equirec.py:65, looped per frame and view by process_equi /
 *   pers2pano_vid / get_anchor_target (inference_dual_p2e.py:113-144, 291-304; animatediff/utils/video_mask.py:158-217). */
int im360_remap_cubic_wrap_u8(const void* img, const void* map_x, const void* map_y, const void* wtab, void* out,
                              int64_t N, int64_t M, int64_t H, int64_t W, int64_t C, int64_t h, int64_t w, void* stream);

/* Largest all-ones rectangle of a HOST uint8 mask [H, W]: rect[4] = (top, left, width, height), scan order and
 * tie-breaking of the reference's Python loops.  Host code (no GPU involved).
 * Replaces: get_maxrec_cord, src/modules/utils.py:39-73. */
int im360_max_rect(const uint8_t* mask, int64_t H, int64_t W, int64_t* rect);

/* A/B switches of the host-side launchers (knob ids: 0 attention query blocks per wave, 1 conv tile policy: 0 never the
 * 256x320 tile, 2 / 3 two-workgroup 128x320 / 128x256 tiles, 2 force 32-channel K steps, 3 scalar temporal attention,
 * 4 conv/GEMM pipeline: 0 two-stage kernel, 1 persistent ring kernel with interleaved asm LDS-DMA requests, 2 / 3 plain ring
 * (builtin / asm LDS-DMA), 4 staggered wave groups, 5 ring kernel for convolutions too; 5 reserved; 6 ablation bits of the
 * ring kernel; 7 halo-patch kernel for the stride-1 3x3 convolutions; 8 taps-innermost K order of the 3x3 convolutions
 * (default 1); 9 packed-rows LayerNorm at 320 channels (default 1); 10 cout groups of the persistent kernel's tile walk:
 * 0 = by weight size, 2 / 4 / 8 forced; 11 text + IP cross attention: 1 key / value sets resident in LDS, 2 the same
 * with 8-byte output stores, 3 (default) the same on twelve-wave workgroups whose waves fetch their query blocks through private LDS
 * rings, 0 the generic two-pass kernel; 12 attention row sums as dot2 of the packed weights + v_permlane32_swap max exchange (sums
 * the ROUNDED weights: results differ in the last bits; default 0); 13 single-buffer / three-waves-per-SIMD attention when there
 * is one key tile (default 1); 14 ablation builds of the d = 64 two-block attention kernel: results are garbage; 15 WarpAttn head groups: the four waves of a
 * workgroup on four (batch, head) pairs over the same query rows, measured slower, default 0; 16 tiles of
 * convolutions with Cout % 128 in (0, 64] on small grids: 2 (default) 128 x 128, 0 / 1 256 x 64 with 32- / 64-channel K steps; 17 d = 64 attention without a bias on four-wave grids: 1 (default)
 * one query block per wave at three waves per SIMD, 0 the round-2 rule (two blocks per wave on large grids)).  Defaults are the measured best; the IM360_* environment variables seed them at load time.  Knobs 7 and 8
 * change the fp32 summation order, 11 (0 against 1 - 3: the scale is folded into K instead of Q) and 12 the 16-bit rounding points, 6 and 14
 * break results on purpose, the others do not change results. */
int im360_tuning_set(int knob, int value);

/* ---- diagnostics: the ONLY process-global state behind this ABI (boundary nit).  A production binding needs neither
 * im360_tuning_set (A/B switches; the defaults are the measured best and no compute entry point requires a knob) nor the two
 * profiling calls below (bench.py's per-class HIP events); every compute entry point above is a pure function of its arguments
 * and the knobs' defaults. ---- */

/* HIP-event profiling of kernel classes (bit k of mask enables class k: 0 attn, 1 temporal, 2 conv,
 * 3 gn_stats, 4 gn_apply, 5 layernorm/geglu/elementwise, 6 conv kernel used as a token-major linear).
 * collect() synchronises on the recorded events. */
void im360_prof_enable(unsigned mask);
int im360_prof_collect(int kind, double* total_ms, long* launches);

#ifdef __cplusplus
}
#endif
#endif
