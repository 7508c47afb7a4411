// Motion-module temporal self-attention (sequence = frames) for gfx950.
//
// Replaces VersatileAttention's explicit baddbmm -> softmax -> bmm path
// (animatediff/models/motion_module.py:343-429 -> diffusers/models/attention_processor.py:562-591),
// including the '(b f) d c -> (b d) f c' / back layout shuffles (motion_module.py:348, 427): the
// kernel indexes the token-major activations [B, F, P, heads*d] with strides instead of moving them.
//
// HBM-bound by its data (AI ~ 8 flop/B) -- but only if the arithmetic is cheap: a scalar-FMA version spends ~3300
// VALU instructions per query row unpacking 16-bit K/V and runs VALU-bound at ~1.7 TB/s.  Up to 16 frames (the
// shipped configuration) therefore run on 16x16 MFMA tiles: one workgroup stages the q|k|v rows of one pixel (all
// frames) into LDS with coalesced 16-byte loads, each wave takes (pixel, head) pairs: S^T = K Q^T in
// ceil(d/32) MFMAs, softmax on the 4 scores a lane holds (+2 cross-lane exchanges), O^T = V^T P^T in ceil(d/16)
// MFMAs with V gathered by the transposing LDS read, results written over the head's Q slice in LDS and then
// streamed out as whole token rows.  HBM sees q, k, v, o exactly once.  17 .. 64 frames (frame-sharded 48-frame runs)
// run the same scheme on FT x FT tiles (temporal_attn_mfma_long_kernel); the scalar kernel below remains as the
// fallback for head groups that do not fit 64 KB of LDS and behind the tattn_scalar knob.
#include <stdlib.h>
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

// One workgroup = PPB pixels x F frames x heads threads (one query row per thread).  The K and V rows of a
// pixel (F x C each, contiguous 2C-element spans of the fused QKV rows) are staged ONCE into LDS with
// coalesced 16-byte loads issued back to back (deep memory-level parallelism), then every thread reads its
// head's K/V chunks from LDS (threads of one head read the same address -> broadcast).  HBM sees q, k, v, o
// exactly once; the scores never leave registers.
template <typename T, int FMAX>
__global__ __launch_bounds__(512) void temporal_attn_lds_kernel(TAttnParams p, int ppb, int hpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* kv = (T*)smem;                                   // [ppb][F][k | v][hpb * d]
    const int F = p.F, d = p.d;
    const int G = hpb * d;                              // channels of this block's head group
    const int h0 = blockIdx.y * hpb;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long pix0 = (long)blockIdx.x * ppb;           // flattened (b, pixel)
    const long npix = (long)p.total;                    // B * P
    // ---- stage K and V of the head group: chunk id -> (pixel, frame, k|v, 16-byte chunk)
    const int cpr = G >> 3;
    const int nchunks = ppb * F * 2 * cpr;
    for (int c = tid; c < nchunks; c += nthr) {
        const int ch = c % cpr;
        const int kvsel = (c / cpr) & 1;
        const int f = (c / (2 * cpr)) % F;
        const int pl = c / (2 * cpr * F);
        const long pix = min(pix0 + pl, npix - 1);
        const long b = pix / p.P, px = pix % p.P;
        const T* base = kvsel ? (const T*)p.v + b * p.v_bs + px * p.v_ps + (long)f * p.v_fs
                              : (const T*)p.k + b * p.k_bs + px * p.k_ps + (long)f * p.k_fs;
        *(uint4*)(kv + (long)c * 8) = *(const uint4*)(base + h0 * d + ch * 8);
    }
    __syncthreads();
    // ---- one query row per thread: tid -> (pixel, frame i, head), head fastest
    const int hl = tid % hpb;
    const int i = (tid / hpb) % F;
    const int pl = tid / (hpb * F);
    const long pix = pix0 + pl;
    if (pl >= ppb || pix >= npix) return;
    const long b = pix / p.P, px = pix % p.P;
    const int nch = d >> 3;
    const T* qp = (const T*)p.q + b * p.q_bs + px * p.q_ps + (long)i * p.q_fs + (long)(h0 + hl) * d;
    T* op = (T*)p.out + b * p.o_bs + px * p.o_ps + (long)i * p.o_fs + (long)(h0 + hl) * d;
    const T* kl = kv + (long)pl * F * 2 * G + hl * d;   // + j * 2G (+ G for V)
    float s[FMAX];
#pragma unroll
    for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
    for (int c = 0; c < nch; ++c) {
        float qf[8];
        unpack8<T>(*(const uint4*)(qp + c * 8), qf);
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            float kf[8];
            unpack8<T>(*(const uint4*)(kl + (long)min(j, F - 1) * 2 * G + c * 8), kf);
            float acc = s[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qf[e], kf[e], acc);
            s[j] = acc;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
        s[j] = (j < F) ? s[j] * p.scale_log2 : -INFINITY;
        m = fmaxf(m, s[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
        s[j] = __builtin_amdgcn_exp2f(s[j] - m);
        l += s[j];
    }
    const float inv = 1.0f / l;
    for (int c = 0; c < nch; ++c) {
        float of[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FMAX; ++j) {
            float vf[8];
            unpack8<T>(*(const uint4*)(kl + (long)min(j, F - 1) * 2 * G + G + c * 8), vf);
#pragma unroll
            for (int e = 0; e < 8; ++e) of[e] = fmaf(s[j], vf[e], of[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] *= inv;
        *(uint4*)(op + c * 8) = pack8<T>(of);
    }
}

// ---- F <= 16: one workgroup (4 waves) = one (batch, pixel) x hpb heads
// NT (knob tattn_nt; tools/ab_knob-style A/B in tools/ab_tattn.py): bit 0 = the output rows leave with non-temporal stores, bit 1 = q | k | v
// arrive with non-temporal loads -- both streams are touched exactly once by this kernel.
template <typename T, int NT = 0>
__global__ __launch_bounds__(256) void temporal_attn_mfma_kernel(TAttnParams p, int hpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* rows = (T*)smem;                                 // [16 frames][q_g | k_g | v_g | 8 pad], g = this block's heads
    const int F = p.F, d = p.d;
    const int G = hpb * d;
    const int pitch = 3 * G + 8;                        // +16 bytes: the 16 frame rows start 4 banks apart
    const int h0 = blockIdx.y * hpb;
    const int tid = threadIdx.x;
    const long pix = blockIdx.x;
    const long b = pix / p.P, px = pix % p.P;
    // ---- stage: chunk id -> (frame, q|k|v, 16-byte chunk); frames >= F are zero rows
    const int cps = G >> 3;
    for (int c = tid; c < 16 * 3 * cps; c += 256) {
        const int ch = c % cps, seg = (c / cps) % 3, f = c / (3 * cps);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (f < F) {
            const T* base = seg == 0 ? (const T*)p.q + b * p.q_bs + px * p.q_ps + (long)f * p.q_fs
                          : seg == 1 ? (const T*)p.k + b * p.k_bs + px * p.k_ps + (long)f * p.k_fs
                                     : (const T*)p.v + b * p.v_bs + px * p.v_ps + (long)f * p.v_fs;
            if constexpr ((NT & 2) != 0) {
                const u32x4 r = __builtin_nontemporal_load((const u32x4*)(base + h0 * d + ch * 8));
                v = make_uint4(r.x, r.y, r.z, r.w);
            } else {
                v = *(const uint4*)(base + h0 * d + ch * 8);
            }
        }
        *(uint4*)(rows + f * pitch + seg * G + ch * 8) = v;
    }
    __syncthreads();
    const int lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int nstep = (d + 31) >> 5, ncb = (d + 15) >> 4;
    for (int hl = wid; hl < hpb; hl += 4) {
        // ---- S^T = K Q^T: lane (query i16, g) ends up with the scores of keys 4g .. 4g+3
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < nstep; ++st) {
            const int k0 = st * 32 + 8 * g;
            uint4 kf = make_uint4(0u, 0u, 0u, 0u), qf = kf;
            if (k0 < d) {                                // d % 8 == 0: an 8-channel chunk is all in or all out
                kf = *(const uint4*)(rows + i16 * pitch + G + hl * d + k0);
                qf = *(const uint4*)(rows + i16 * pitch + hl * d + k0);
            }
            s = Mfma16<T>::k32(kf, qf, s);
        }
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = (4 * g + r < F) ? s[r] * p.scale_log2 : -INFINITY;
            m = fmaxf(m, s[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m);
            l += s[r];
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        u32x2 pf;                                        // P^T as the B operand: column = query, k = keys 4g .. 4g+3
        pf.x = pack2<T>(s[0], s[1]);
        pf.y = pack2<T>(s[2], s[3]);
        // ---- O^T = V^T P^T per 16-channel block; lane (query i16, g) gets channels c0 + 4g .. +3
        for (int cb = 0; cb < ncb; ++cb) {
            const int c0 = cb * 16;
            const u32x2 vf = lds_read_tr16(rows + (4 * g + (i16 >> 2)) * pitch + 2 * G + hl * d + c0 + 4 * (i16 & 3));
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = Mfma16<T>::k16(vf, pf, o);
            if (c0 + 4 * g < d) {                        // the head's Q slice is dead after QK^T: park the output there
                uint2 w;
                w.x = pack2<T>(o[0] * inv, o[1] * inv);
                w.y = pack2<T>(o[2] * inv, o[3] * inv);
                *(uint2*)(rows + i16 * pitch + hl * d + c0 + 4 * g) = w;
            }
        }
    }
    __syncthreads();
    // ---- whole token rows out
    for (int c = tid; c < 16 * cps; c += 256) {
        const int ch = c % cps, f = c / cps;
        if (f < F) {
            const uint4 w = *(const uint4*)(rows + f * pitch + ch * 8);
            T* dst = (T*)p.out + b * p.o_bs + px * p.o_ps + (long)f * p.o_fs + h0 * d + ch * 8;
            if constexpr ((NT & 1) != 0) __builtin_nontemporal_store(u32x4{w.x, w.y, w.z, w.w}, (u32x4*)dst);
            else *(uint4*)dst = w;
        }
    }
}


// ---- 17 <= F <= 64 (frame-sharded long clips, BASELINE cfg4): the same scheme on FT x FT tiles of 16 x 16.  One workgroup
//      stages the FT * 16 frame rows (q | k | v of its head group) of one pixel; the (head, query tile) pairs are dealt to
//      the four waves; per pair: S^T against all FT key tiles (FT * ceil(d / 32) MFMAs), softmax over the FT * 4 scores a
//      lane holds (+ 2 exchanges), O^T = sum over key tiles of V^T P^T, parked over the head's Q rows of that query tile.
template <typename T, int FT>
__global__ __launch_bounds__(256) void temporal_attn_mfma_long_kernel(TAttnParams p, int hpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* rows = (T*)smem;                                 // [FT * 16 frames][q_g | k_g | v_g | 8 pad]
    constexpr int NR = FT * 16;
    const int F = p.F, d = p.d;
    const int G = hpb * d;
    const int pitch = 3 * G + 8;
    const int h0 = blockIdx.y * hpb;
    const int tid = threadIdx.x;
    const long pix = blockIdx.x;
    const long b = pix / p.P, px = pix % p.P;
    const int cps = G >> 3;
    for (int c = tid; c < NR * 3 * cps; c += 256) {
        const int ch = c % cps, seg = (c / cps) % 3, f = c / (3 * cps);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (f < F) {
            const T* base = seg == 0 ? (const T*)p.q + b * p.q_bs + px * p.q_ps + (long)f * p.q_fs
                          : seg == 1 ? (const T*)p.k + b * p.k_bs + px * p.k_ps + (long)f * p.k_fs
                                     : (const T*)p.v + b * p.v_bs + px * p.v_ps + (long)f * p.v_fs;
            v = *(const uint4*)(base + h0 * d + ch * 8);
        }
        *(uint4*)(rows + f * pitch + seg * G + ch * 8) = v;
    }
    __syncthreads();
    const int lane = tid & 63, wid = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int nstep = (d + 31) >> 5, ncb = (d + 15) >> 4;
    for (int pair = wid; pair < hpb * FT; pair += 4) {
        const int hl = pair / FT, qi = pair % FT;
        const T* qrow = rows + (qi * 16 + i16) * pitch + hl * d;
        // ---- S^T tiles: lane (query qi * 16 + i16, g) gets the scores of keys kt * 16 + 4g .. + 3
        f32x4 s[FT];
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < nstep; ++st) {
            const int k0 = st * 32 + 8 * g;
            uint4 qf = make_uint4(0u, 0u, 0u, 0u);
            if (k0 < d) qf = *(const uint4*)(qrow + k0);
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                uint4 kf = make_uint4(0u, 0u, 0u, 0u);
                if (k0 < d) kf = *(const uint4*)(rows + (kt * 16 + i16) * pitch + G + hl * d + k0);
                s[kt] = Mfma16<T>::k32(kf, qf, s[kt]);
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < FT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][r] = (kt * 16 + 4 * g + r < F) ? s[kt][r] * p.scale_log2 : -INFINITY;
                m = fmaxf(m, s[kt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
        u32x2 pf[FT];                                    // P^T per key tile as the B operand
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m);
                l += s[kt][r];
            }
            pf[kt].x = pack2<T>(s[kt][0], s[kt][1]);
            pf[kt].y = pack2<T>(s[kt][2], s[kt][3]);
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        // ---- O^T = sum over key tiles of V^T P^T, per 16-channel block
        for (int cb = 0; cb < ncb; ++cb) {
            const int c0 = cb * 16;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                const u32x2 vf = lds_read_tr16(rows + (kt * 16 + 4 * g + (i16 >> 2)) * pitch + 2 * G + hl * d + c0 + 4 * (i16 & 3));
                o = Mfma16<T>::k16(vf, pf[kt], o);
            }
            if (c0 + 4 * g < d) {                        // this query tile's Q rows of the head are dead: park the output there
                uint2 w;
                w.x = pack2<T>(o[0] * inv, o[1] * inv);
                w.y = pack2<T>(o[2] * inv, o[3] * inv);
                *(uint2*)(rows + (qi * 16 + i16) * pitch + hl * d + c0 + 4 * g) = w;
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < NR * cps; c += 256) {
        const int ch = c % cps, f = c / cps;
        if (f < F)
            *(uint4*)((T*)p.out + b * p.o_bs + px * p.o_ps + (long)f * p.o_fs + h0 * d + ch * 8) =
                *(const uint4*)(rows + f * pitch + ch * 8);
    }
}

template <typename T, int FMAX>
static void launch_tattn_v(TAttnParams p, hipStream_t stream) {
    const long npix = p.total / ((long)p.F * p.heads);                 // B * P
    int hpb = p.heads;                                                  // heads per block: keep one pixel's K|V <= 48 KiB
    while (hpb > 1 && (hpb % 2) == 0 && ((long)p.F * 2 * hpb * p.d * sizeof(T) > 48 * 1024 || p.F * hpb > 512)) hpb /= 2;
    const long bytes_per_pix = (long)p.F * 2 * hpb * p.d * sizeof(T);
    const int per_pix_threads = p.F * hpb;
    int ppb = 256 / per_pix_threads;
    if (ppb < 1) ppb = 1;
    while (ppb > 1 && ppb * bytes_per_pix > 48 * 1024) --ppb;
    const int threads = ((ppb * per_pix_threads + 63) / 64) * 64;
    p.total = npix;
    dim3 grid((unsigned)((npix + ppb - 1) / ppb), (unsigned)(p.heads / hpb));
    hipLaunchKernelGGL((temporal_attn_lds_kernel<T, FMAX>), grid, dim3(threads), (size_t)(ppb * bytes_per_pix), stream, p, ppb, hpb);
}

template <typename T>
static int launch_tattn(const TAttnParams& p, hipStream_t stream) {
    const int scalar_env = knob(KNOB_TATTN_SCALAR);   // tuning override
    if (p.F <= 16 && !scalar_env) {
        int hpb = p.heads;                               // heads per workgroup: ~32 KB of LDS rows (3 * hpb * d <= 1248 channels)
        while (hpb > 1 && (hpb % 2) == 0 && hpb * p.d > 416) hpb /= 2;
        const size_t lds = (size_t)16 * (3 * hpb * p.d + 8) * sizeof(T) + 64;
        if (lds <= 64 * 1024) {
            const long npix = p.total / ((long)p.F * p.heads);
            dim3 grid((unsigned)npix, (unsigned)(p.heads / hpb));
            switch (knob(KNOB_TATTN_NT) & 3) {
                case 1: hipLaunchKernelGGL((temporal_attn_mfma_kernel<T, 1>), grid, dim3(256), lds, stream, p, hpb); break;
                case 2: hipLaunchKernelGGL((temporal_attn_mfma_kernel<T, 2>), grid, dim3(256), lds, stream, p, hpb); break;
                case 3: hipLaunchKernelGGL((temporal_attn_mfma_kernel<T, 3>), grid, dim3(256), lds, stream, p, hpb); break;
                default: hipLaunchKernelGGL((temporal_attn_mfma_kernel<T, 0>), grid, dim3(256), lds, stream, p, hpb); break;
            }
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (p.F > 16 && !scalar_env) {
        const int ft = (p.F + 15) / 16, nr = ft * 16;
        int hpb = p.heads;                               // heads per workgroup: nr rows of (3 * hpb * d + 8) elements in <= 64 KB
        while (hpb > 1 && (hpb % 2) == 0 && (size_t)nr * (3 * hpb * p.d + 8) * sizeof(T) + 64 > 64 * 1024) hpb /= 2;
        const size_t lds = (size_t)nr * (3 * hpb * p.d + 8) * sizeof(T) + 64;
        if (lds <= 64 * 1024) {
            const long npix = p.total / ((long)p.F * p.heads);
            dim3 grid((unsigned)npix, (unsigned)(p.heads / hpb));
            if (ft == 2) hipLaunchKernelGGL((temporal_attn_mfma_long_kernel<T, 2>), grid, dim3(256), lds, stream, p, hpb);
            else if (ft == 3) hipLaunchKernelGGL((temporal_attn_mfma_long_kernel<T, 3>), grid, dim3(256), lds, stream, p, hpb);
            else hipLaunchKernelGGL((temporal_attn_mfma_long_kernel<T, 4>), grid, dim3(256), lds, stream, p, hpb);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (p.F <= 16) launch_tattn_v<T, 16>(p, stream);
    else if (p.F <= 32) launch_tattn_v<T, 32>(p, stream);
    else launch_tattn_v<T, 64>(p, stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

}  // namespace im360

extern "C" __attribute__((visibility("default"))) int im360_temporal_attn_fwd(const void* q, const void* k, const void* v, void* out,
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
    IM360_CHECK_ARG(F * 2 * d * 2 <= 48 * 1024, "temporal_attn_fwd: one head's K|V rows (%ld frames x %ld) exceed the LDS budget", (long)F, (long)d);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_TEMPORAL, stream);
    if (dtype == 0) return launch_tattn<__bf16>(p, s);
    if (dtype == 1) return launch_tattn<_Float16>(p, s);
    im360_set_error("temporal_attn_fwd: dtype %d unsupported", dtype);
    return IM360_ERR_UNSUPPORTED;
}
