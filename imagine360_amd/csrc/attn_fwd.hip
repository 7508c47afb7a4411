// Flash-attention forward for gfx950 (CDNA4): softmax(Q K^T * scale + bias) V on 16-bit
// operands with fp32 accumulation, head dim D in {32, 64}.
//
// Replaces the reference's xformers / SDPA call sites:
//   spatial self-attention        diffusers/models/attention_processor.py:1195-1371 (d=64)
//   text + IP cross-attention     animatediff/models/attention.py:65-156 (two KV sets, `accumulate`)
//   cross-view WarpAttn attention src/modules/transformer.py:59-74 (d=32, additive bias shared by
//                                 every (batch, head) -- mask[0] broadcast, transformer.py:68-70)
//
// Design (wave64, v_mfma_f32_32x32x16):
//   * one wave owns 32 query rows; a workgroup of NW waves shares one (batch, head) and streams
//     K/V in 64-key tiles through LDS (register prefetch of tile t+1 overlaps compute of tile t).
//   * S^T = K Q^T ("swapped QK^T"): every lane holds 16 scores of ONE query column, so the row
//     max/sum are in-lane reductions plus a single lane^32 exchange.
//   * O^T += V^T P^T: the exponentiated scores are already in B-operand order; V is stored
//     transposed in LDS ([D][64+4], pair-packed on the way in) and keys are consumed in the
//     C-fragment's own order, so no cross-lane permutes are needed.  The running max / sum /
//     rescale are lane-local because O^T keeps the query in the lane index.
#include "common.h"

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
};

constexpr int KVB = 64;            // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 5.0f;   // log2 units: P stays <= 32 between rescales

template <typename T, int D, int NW, bool HAS_BIAS>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_kernel(AttnParams p) {
    constexpr int NT = NW * 64;
    constexpr int KP = D + 8;          // K tile pitch (elements): 16-B slots rotate by an odd count per row
    constexpr int VP = KVB;            // V^T tile pitch (elements): unpadded, 16-byte groups XOR-swizzled by row
    constexpr int DC = D / 16;         // k-steps of the QK^T contraction
    constexpr int DV = D / 32;         // 32-wide blocks of the output head dim
    constexpr int KCH = KVB * D / 8;   // 16-byte chunks in a K tile
    constexpr int KLD = (KCH + NT - 1) / NT;
    constexpr int VIT = (KVB / 2) * (D / 8);   // (key pair, 8-channel chunk) items of a V tile
    constexpr int VLD = (VIT + NT - 1) / NT;

    // double-buffered K / V^T tiles: one barrier per KV tile (the next tile is written while this one is consumed)
    __shared__ __attribute__((aligned(16))) T k_lds2[2][KVB * KP];
    __shared__ __attribute__((aligned(16))) T vt_lds2[2][D * VP];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    // XCD-aware block order: the dispatcher deals consecutive block ids round-robin to the 8 XCDs (private L2s).
    // Give every XCD one contiguous range of the (batch*head major, q-tile minor) space, so the ~100 workgroups
    // resident on an XCD at any moment share ONE head's K/V (2 MB at 8192 keys) in that XCD's 4 MB L2 instead of
    // each streaming its own from HBM.
    long lb = blockIdx.x;
    {
        const long nb = gridDim.x, qn = nb / 8, rn = nb % 8, xcd = lb % 8, idx = lb / 8;
        lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    }
    const int bh = (int)(lb / p.nqt);
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = (int)(lb % p.nqt) * (32 * NW) + wid * 32;

    const T* qb = (const T*)p.q + (long)b * p.q_bs + (long)h * D;
    const T* kb_ = (const T*)p.k + (long)(b / p.kv_group) * p.k_bs + (long)h * D;
    const T* vb = (const T*)p.v + (long)(b / p.kv_group) * p.v_bs + (long)h * D;
    const T* bias = (const T*)p.bias;
    if (HAS_BIAS && p.bias_sel != nullptr && __builtin_nontemporal_load(p.bias_sel) != 0) bias = (const T*)p.bias_alt;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 dc + 8 hi .. +7]
    int qrow = q0 + col;
    const bool q_valid = qrow < p.Nq;
    if (!q_valid) qrow = p.Nq - 1;
    uint4 qf[DC];
#pragma unroll
    for (int dc = 0; dc < DC; ++dc)
        qf[dc] = *(const uint4*)(qb + (long)qrow * p.q_rs + dc * 16 + hi * 8);

    f32x16 o[DV];
#pragma unroll
    for (int i = 0; i < DV; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (p.Nk + KVB - 1) / KVB;
    uint4 kreg[KLD];
    uint4 vreg[VLD][2];

    auto load_tile = [&](int t) {
        const int kv0 = t * KVB;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NT;
            const int row = (c / (D / 8)) % KVB, c8 = c % (D / 8);
            // unconditional loads (rows clamped into range): a predicated load inside an unrolled loop makes hipcc
            // branch around it and drain vmcnt per load.  Out-of-range keys are masked to -inf below, so their
            // (finite, duplicated) K/V rows never contribute.
            const int rr = min(kv0 + row, p.Nk - 1);
            kreg[i] = *(const uint4*)(kb_ + (long)rr * p.k_rs + (c8 % (D / 8)) * 8);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NT;
            const int kp = c / (D / 8), c8 = c % (D / 8);
            const int r0 = min(kv0 + 2 * (kp % (KVB / 2)), p.Nk - 1), r1 = min(kv0 + 2 * (kp % (KVB / 2)) + 1, p.Nk - 1);
            vreg[i][0] = *(const uint4*)(vb + (long)r0 * p.v_rs + c8 * 8);
            vreg[i][1] = *(const uint4*)(vb + (long)r1 * p.v_rs + c8 * 8);
        }
    };
    auto store_tile = [&](int buf) {
        T* k_lds = k_lds2[buf];
        T* vt_lds = vt_lds2[buf];
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int c = tid + i * NT;
            const int row = c / (D / 8), c8 = c % (D / 8);
            if (c < KCH) *(uint4*)(k_lds + row * KP + c8 * 8) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int c = tid + i * NT;
            const int kp = c / (D / 8), c8 = c % (D / 8);
            if (c < VIT) {
                const uint32_t a[4] = {vreg[i][0].x, vreg[i][0].y, vreg[i][0].z, vreg[i][0].w};
                const uint32_t bq[4] = {vreg[i][1].x, vreg[i][1].y, vreg[i][1].z, vreg[i][1].w};
                uint32_t* dst = (uint32_t*)vt_lds;
                // position of key 2kp inside the row: within each 16-key chunk the 8 keys one lane-half consumes
                // (C-fragment order) are made contiguous, so the PV operand is ONE 16-byte read
                const int k0 = 2 * kp, kk = k0 & 15;
                const int ps = (k0 & ~15) + (((kk >> 2) & 1) << 3) + (kk & 3) + ((kk >> 3) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // channels 2j (low halves) and 2j+1 (high halves) of keys 2kp, 2kp+1
                    const uint32_t lo = (a[j] & 0xffffu) | (bq[j] << 16);
                    const uint32_t hi_ = (a[j] >> 16) | (bq[j] & 0xffff0000u);
                    const int r0 = c8 * 8 + 2 * j, r1 = r0 + 1;
                    dst[(r0 * VP + ((((ps >> 3) ^ ((r0 >> 3) ^ r0)) & 7) << 3) + (ps & 7)) >> 1] = lo;
                    dst[(r1 * VP + ((((ps >> 3) ^ ((r1 >> 3) ^ r1)) & 7) << 3) + (ps & 7)) >> 1] = hi_;
                }
            }
        }
    };

    load_tile(0);
    store_tile(0);
    if (ntiles > 1) load_tile(1);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                 // tile t is visible; every wave is done with tile t-1 (buffer (t+1)&1)
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        if (t + 2 < ntiles) load_tile(t + 2);      // in flight during the whole compute phase below
        const T* k_lds = k_lds2[t & 1];
        const T* vt_lds = vt_lds2[t & 1];
        const int kv0 = t * KVB;

        // ---- S^T = K Q^T : s[kb][r] = score(query col, key kv0 + 32 kb + row(r, hi))
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const uint4 a = *(const uint4*)(k_lds + (kb * 32 + col) * KP + dc * 16 + hi * 8);
                s[kb] = Elem<T>::mfma32(a, qf[dc], s[kb]);
            }
        }
        // ---- two 32-key halves, each with its own online-softmax update.  Issue order matters more than
        //      instruction count here: the second half's QK^T MFMAs (issued above) run in the matrix pipe while the
        //      first half's softmax runs on the VALU, and the first half's PV MFMAs run under the second half's softmax.
        const float sc = HAS_BIAS ? 1.0f : p.scale_log2;            // p = exp2(u * sc - m * sc)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (HAS_BIAS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = min(kv0 + kb * 32 + 8 * g + 4 * hi, p.Nk - 4);     // Nk % 4 == 0 (checked on the host)
                    const uint2 w = *(const uint2*)(bias + (long)qrow * p.bias_rs + key0);
                    s[kb][4 * g + 0] = fmaf(s[kb][4 * g + 0], p.scale_log2, unpack_lo<T>(w.x) * LOG2E);
                    s[kb][4 * g + 1] = fmaf(s[kb][4 * g + 1], p.scale_log2, unpack_hi<T>(w.x) * LOG2E);
                    s[kb][4 * g + 2] = fmaf(s[kb][4 * g + 2], p.scale_log2, unpack_lo<T>(w.y) * LOG2E);
                    s[kb][4 * g + 3] = fmaf(s[kb][4 * g + 3], p.scale_log2, unpack_hi<T>(w.y) * LOG2E);
                }
            }
            if (kv0 + kb * 32 + 32 > p.Nk) {                        // only a partial last half masks keys
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + mfma32_row(r, hi) >= p.Nk) s[kb][r] = -INFINITY;
            }
            float mloc = s[kb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            // deferred rescale: keep the old running max while this half's max exceeds it by less than
            // RESCALE_THR (log2 units), so P <= 2^THR and the O / l rescale pass is skipped for most halves
            if (__any(mloc * sc > m_run * sc + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, mloc);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DV; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            // P = exp2(u * sc - m * sc), packed straight into MFMA B-operand order
            const float msc = m_run * sc;
            float pv[16];
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, -msc));
                lsum += pv[r];
            }
            l_run += lsum;
            const uint4 pf0 = pack8<T>(pv), pf1 = pack8<T>(pv + 8);
            // O^T += V^T P^T for this half
#pragma unroll
            for (int dvb = 0; dvb < DV; ++dvb) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = dvb * 32 + col;
                    const int g = kb * 4 + 2 * c + hi;                 // 16-byte group holding this lane-half's 8 keys
                    const uint4 vf = *(const uint4*)(vt_lds + row * VP + (((g ^ ((row >> 3) ^ row)) & 7) << 3));
                    o[dvb] = Elem<T>::mfma32(vf, c == 0 ? pf0 : pf1, o[dvb]);
                }
            }
        }
    }

    // ---- epilogue: normalise, optional accumulate, store 4 consecutive channels per (lane, group)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = p.out_scale / l_tot;
    if (q_valid) {
        T* ob = (T*)p.out + (long)b * p.o_bs + (long)qrow * p.o_rs + (long)h * D;
#pragma unroll
        for (int dvb = 0; dvb < DV; ++dvb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = o[dvb][4 * g + j] * inv;
                uint2* dst = (uint2*)(ob + dvb * 32 + 8 * g + 4 * hi);
                if (p.accumulate) {
                    const uint2 old = *dst;
                    f[0] += unpack_lo<T>(old.x); f[1] += unpack_hi<T>(old.x);
                    f[2] += unpack_lo<T>(old.y); f[3] += unpack_hi<T>(old.y);
                }
                uint2 w;
                w.x = pack2<T>(f[0], f[1]);
                w.y = pack2<T>(f[2], f[3]);
                *dst = w;
            }
        }
    }
}

template <typename T, int D, bool HAS_BIAS>
static int launch_attn_b(AttnParams p, hipStream_t stream) {
    const int nw = p.Nq <= 32 ? 1 : (p.Nq <= 64 ? 2 : 4);
    p.nqt = (p.Nq + 32 * nw - 1) / (32 * nw);
    const long nblk = (long)p.B * p.H * p.nqt;
    if (nblk > 0x7fffffffL) {
        im360_set_error("attn_fwd: %ld workgroups exceed the grid limit", nblk);
        return IM360_ERR_ARG;
    }
    dim3 grid((unsigned)nblk, 1, 1);
    if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 1, HAS_BIAS>), grid, dim3(64), 0, stream, p);
    else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2, HAS_BIAS>), grid, dim3(128), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, HAS_BIAS>), grid, dim3(256), 0, stream, p);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

template <typename T, int D>
static int launch_attn(const AttnParams& p, hipStream_t stream) {
    return p.bias ? launch_attn_b<T, D, true>(p, stream) : launch_attn_b<T, D, false>(p, stream);
}

}  // namespace im360

extern "C" int im360_attn_fwd(const void* q, const void* k, const void* v, const void* bias, void* out,
                              int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                              int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                              int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, int64_t bias_rs,
                              int64_t kv_group, float scale, float out_scale, int accumulate, int dtype, void* stream,
                              const void* bias_alt, const void* bias_sel) {
    using namespace im360;
    IM360_CHECK_ARG(q && k && v && out, "attn_fwd: null pointer");
    IM360_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attn_fwd: empty problem B=%ld H=%ld Nq=%ld Nk=%ld",
                    (long)B, (long)H, (long)Nq, (long)Nk);
    IM360_CHECK_ARG(kv_group >= 1, "attn_fwd: kv_group must be >= 1");
    IM360_CHECK_ARG(D == 32 || D == 64, "attn_fwd: head dim %ld unsupported (32, 64)", (long)D);
    IM360_CHECK_ARG(B * H <= 0x7fffffffL, "attn_fwd: B*H too large");
    IM360_CHECK_ARG((q_rs % 8) == 0 && (k_rs % 8) == 0 && (v_rs % 8) == 0 && (o_rs % 4) == 0 &&
                    (q_bs % 8) == 0 && (k_bs % 8) == 0 && (v_bs % 8) == 0 && (o_bs % 4) == 0,
                    "attn_fwd: strides must keep 16-byte (q,k,v) / 8-byte (out) alignment");
    IM360_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)out % 8) == 0, "attn_fwd: misaligned base pointer");
    if (bias) {
        IM360_CHECK_ARG((Nk % 4) == 0 && Nk >= 4 && (bias_rs % 4) == 0 && ((uintptr_t)bias % 8) == 0,
                        "attn_fwd: bias needs Nk %% 4 == 0 and 8-byte aligned rows");
    }
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.bias = bias; p.out = out;
    p.bias_alt = bias_alt; p.bias_sel = (const int*)bias_sel;
    IM360_CHECK_ARG(!bias_sel || (bias && bias_alt && ((uintptr_t)bias_alt % 8) == 0), "attn_fwd: bias_sel needs bias and an aligned bias_alt");
    p.B = (int)B; p.H = (int)H; p.Nq = (int)Nq; p.Nk = (int)Nk; p.kv_group = (int)kv_group;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs;
    p.o_bs = o_bs; p.o_rs = o_rs; p.bias_rs = bias_rs;
    p.scale_log2 = scale * LOG2E; p.out_scale = out_scale; p.accumulate = accumulate;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, stream);
    if (dtype == 0) return D == 64 ? launch_attn<__bf16, 64>(p, s) : launch_attn<__bf16, 32>(p, s);
    if (dtype == 1) return D == 64 ? launch_attn<_Float16, 64>(p, s) : launch_attn<_Float16, 32>(p, s);
    im360_set_error("attn_fwd: dtype %d unsupported (0=bf16, 1=f16)", dtype);
    return IM360_ERR_UNSUPPORTED;
}
