// Parameter block shared by the attention kernels (attn_fwd.hip, attn_pipe.hip).  Host + device, gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
namespace im360 {

struct AttnParams {
    const void* q; const void* k; const void* v; const void* bias; void* out;
    const void* bias_alt; const int* bias_sel;     // *bias_sel != 0 -> use bias_alt (decided on the device: graph-replay safe)
    int B, H, Nq, Nk;
    int nqt;             // query tiles per (batch, head)
    int kv_group;        // K/V batch index = query batch index / kv_group (context shared by the frames of a video)
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, bias_rs;   // element strides
    float scale_log2;    // logit scale * log2(e)
    float out_scale;     // multiplies the normalised result
    int accumulate;      // out += result instead of out = result
    int bias_packed;     // bias / bias_alt are fp16 matrices pre-multiplied by log2(e) (im360_attn_pack_bias)
    // packed bias only (round 6): one bit per (32-query block, 32-key half), row stride blocks_rs words: 0 = every entry of that 32 x 32
    // block of the packed matrix is zero, the kernel then skips its fragment loads and its two bias MFMAs.  nullptr: no map.
    const uint32_t* bias_blocks = nullptr; const uint32_t* bias_blocks_alt = nullptr; int blocks_rs = 0;
    // optional SECOND key/value set of the same queries (DUAL kernels): out = out_scale * attn(q, k, v) + out_scale2 *
    // attn(q, k2, v2), two independent softmaxes -- the text + IP-adapter cross attention in ONE launch
    const void* k2; const void* v2;
    int Nk2;
    long k2_bs, k2_rs, v2_bs, v2_rs;
    float out_scale2;
    // resident-K/V cross attention (xattn_resident_kernel): query blocks (32 rows) per image, per (K/V batch, head) pair, in total
    int x_nqb, x_bpp;
    long x_total;
};

// Software-pipelined d = 64 self-attention (attn_pipe.hip).  Returns IM360_OK, or 1 when the shape is not one it takes
// (the caller then falls back to attn_fwd_kernel).
int launch_attn_pipe(const AttnParams& p, int dtype, hipStream_t stream);

}  // namespace im360
