// Motion-module temporal self-attention (sequence = frames) for gfx950.
//
// Replaces VersatileAttention's explicit baddbmm -> softmax -> bmm path
// (animatediff/models/motion_module.py:343-429 -> diffusers/models/attention_processor.py:562-591),
// including the '(b f) d c -> (b d) f c' / back layout shuffles (motion_module.py:348, 427): the
// kernel indexes the token-major activations [B, F, P, heads*d] with strides instead of moving them.
//
// HBM-bound (16x16 scores, AI ~ 8 flop/B): no MFMA.  One thread owns QPT query rows of one (batch, pixel,
// head); the threads of a pixel read the same K/V rows, so those loads are L1
// broadcasts and HBM sees q, k, v, o exactly once.  Adjacent lanes are adjacent heads of the same
// token, so a wave's 16-byte loads cover whole contiguous token rows.
#include "common.h"

namespace im360 {

struct TAttnParams {
    const void* q; const void* k; const void* v; void* out;
    long total;             // B * P * F * heads threads
    int F, P, heads, d;
    long q_fs, q_ps, q_bs;  // element strides of (frame, pixel, batch) for q; k, v share them via offsets
    long k_fs, k_ps, k_bs;
    long v_fs, v_ps, v_bs;
    long o_fs, o_ps, o_bs;
    float scale_log2;
};

// QPT query frames per thread: each K/V chunk fetched from L1 is reused for QPT queries, which cuts the
// L1 request count (the limiter of the one-query-per-thread form) by QPT
template <typename T, int FMAX, int QPT>
__global__ __launch_bounds__(256) void temporal_attn_kernel(TAttnParams p) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.total) return;
    const int ngrp = (p.F + QPT - 1) / QPT;
    const int h = (int)(gid % p.heads);
    long r = gid / p.heads;
    const int ig = (int)(r % ngrp);
    r /= ngrp;
    const int px = (int)(r % p.P);
    const int b = (int)(r / p.P);
    const int d = p.d, nch = d >> 3;
    const T* qp = (const T*)p.q + (long)b * p.q_bs + (long)px * p.q_ps + (long)h * d;
    const T* kp = (const T*)p.k + (long)b * p.k_bs + (long)px * p.k_ps + (long)h * d;
    const T* vp = (const T*)p.v + (long)b * p.v_bs + (long)px * p.v_ps + (long)h * d;
    T* op = (T*)p.out + (long)b * p.o_bs + (long)px * p.o_ps + (long)h * d;
    int qi[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) qi[t] = min(ig * QPT + t, p.F - 1);      // clamped duplicates are not stored

    float s[QPT][FMAX];
#pragma unroll
    for (int t = 0; t < QPT; ++t)
#pragma unroll
        for (int j = 0; j < FMAX; ++j) s[t][j] = 0.f;
    for (int c = 0; c < nch; ++c) {
        float qf[QPT][8];
#pragma unroll
        for (int t = 0; t < QPT; ++t) unpack8<T>(*(const uint4*)(qp + (long)qi[t] * p.q_fs + c * 8), qf[t]);
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            // unconditional load of a clamped frame: a predicated load here makes hipcc branch around every load
            // and drain vmcnt per element (dependent L2 round trips); frames >= F are masked to -inf below
            float kf[8];
            unpack8<T>(*(const uint4*)(kp + (long)min(j, p.F - 1) * p.k_fs + c * 8), kf);
#pragma unroll
            for (int t = 0; t < QPT; ++t) {
                float acc = s[t][j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(qf[t][e], kf[e], acc);
                s[t][j] = acc;
            }
        }
    }
    float inv[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            s[t][j] = (j < p.F) ? s[t][j] * p.scale_log2 : -INFINITY;
            m = fmaxf(m, s[t][j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            s[t][j] = __builtin_amdgcn_exp2f(s[t][j] - m);
            l += s[t][j];
        }
        inv[t] = 1.0f / l;
    }
    for (int c = 0; c < nch; ++c) {
        float of[QPT][8];
#pragma unroll
        for (int t = 0; t < QPT; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) of[t][e] = 0.f;
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            float vf[8];                                    // frames >= F carry probability exactly 0
            unpack8<T>(*(const uint4*)(vp + (long)min(j, p.F - 1) * p.v_fs + c * 8), vf);
#pragma unroll
            for (int t = 0; t < QPT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) of[t][e] = fmaf(s[t][j], vf[e], of[t][e]);
        }
#pragma unroll
        for (int t = 0; t < QPT; ++t) {
            if (ig * QPT + t < p.F) {
#pragma unroll
                for (int e = 0; e < 8; ++e) of[t][e] *= inv[t];
                *(uint4*)(op + (long)qi[t] * p.o_fs + c * 8) = pack8<T>(of[t]);
            }
        }
    }
}

template <typename T, int FMAX, int QPT>
static void launch_tattn_v(TAttnParams p, hipStream_t stream) {
    const long ngrp = (p.F + QPT - 1) / QPT;
    p.total = (long)(p.total / p.F) * ngrp;          // B * P * heads * query groups
    const long blocks = (p.total + 255) / 256;
    hipLaunchKernelGGL((temporal_attn_kernel<T, FMAX, QPT>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
}

template <typename T>
static int launch_tattn(const TAttnParams& p, hipStream_t stream) {
    if (p.F <= 16) launch_tattn_v<T, 16, 4>(p, stream);
    else if (p.F <= 32) launch_tattn_v<T, 32, 2>(p, stream);
    else launch_tattn_v<T, 64, 1>(p, stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

}  // namespace im360

extern "C" int im360_temporal_attn_fwd(const void* q, const void* k, const void* v, void* out,
                                       int64_t B, int64_t F, int64_t P, int64_t heads, int64_t d,
                                       int64_t qkv_fs, int64_t qkv_ps, int64_t qkv_bs,
                                       int64_t o_fs, int64_t o_ps, int64_t o_bs,
                                       float scale, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(q && k && v && out, "temporal_attn_fwd: null pointer");
    IM360_CHECK_ARG(B > 0 && F > 0 && P > 0 && heads > 0 && d > 0, "temporal_attn_fwd: empty problem");
    IM360_CHECK_ARG(F <= 64, "temporal_attn_fwd: %ld frames > 64 (reference PE max_len, prompt-dual.yaml:28)", (long)F);
    IM360_CHECK_ARG((d % 8) == 0, "temporal_attn_fwd: head dim %ld must be a multiple of 8", (long)d);
    IM360_CHECK_ARG((qkv_fs % 8) == 0 && (qkv_ps % 8) == 0 && (qkv_bs % 8) == 0 && (o_fs % 8) == 0 &&
                    (o_ps % 8) == 0 && (o_bs % 8) == 0, "temporal_attn_fwd: strides must be multiples of 8 elements");
    IM360_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "temporal_attn_fwd: misaligned base pointer");
    TAttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.F = (int)F; p.P = (int)P; p.heads = (int)heads; p.d = (int)d;
    p.total = B * P * F * heads;
    p.q_fs = p.k_fs = p.v_fs = qkv_fs; p.q_ps = p.k_ps = p.v_ps = qkv_ps; p.q_bs = p.k_bs = p.v_bs = qkv_bs;
    p.o_fs = o_fs; p.o_ps = o_ps; p.o_bs = o_bs;
    p.scale_log2 = scale * 1.4426950408889634f;
    IM360_CHECK_ARG((p.total + 255) / 256 <= 0x7fffffffL, "temporal_attn_fwd: problem too large");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_TEMPORAL, stream);
    if (dtype == 0) return launch_tattn<__bf16>(p, s);
    if (dtype == 1) return launch_tattn<_Float16>(p, s);
    im360_set_error("temporal_attn_fwd: dtype %d unsupported", dtype);
    return IM360_ERR_UNSUPPORTED;
}
