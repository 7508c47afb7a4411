// HBM-bound token-wise kernels for gfx950: LayerNorm (with the two positional-encoding adds of the
// reference fused in) and the GEGLU gate.
//
// Replaces:
//   nn.LayerNorm call sites of the transformer blocks (animatediff/models/attention.py:463-507,
//   motion_module.py:249,256; src/modules/transformer.py:160-165) together with
//     pre-add : `query + query_pe`, `context + pe` before norm1 in WarpAttn (transformer.py:156-163,
//               src/modules/attn_perspano.py:56,63)
//     post-add: the temporal PositionalEncoding added to the normalised tokens (motion_module.py:349-350)
//   GEGLU: hidden * gelu(gate) (diffusers/models/activations.py:93-125; src/modules/transformer.py:10-16)
#include "common.h"

namespace im360 {

// one wave per ROWS consecutive token rows; row values stay in registers between the two statistics passes.  ROWS > 1
// for narrow rows: a 320-channel row is one 16-byte load on 40 of the 64 lanes, and one such load per wave in flight
// (32 waves x 640 B = 20 KB per CU) does not cover the HBM latency; all ROWS x MAXCH loads are issued before any use.
template <typename T, int MAXCH, int ROWS>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                         const T* __restrict__ beta, const T* __restrict__ pre,
                                                         const T* __restrict__ post, T* __restrict__ y, long rows, int C,
                                                         long pre_period, long post_div, long post_mod, float eps) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= rows) return;
    const int nch = C >> 3;
    uint4 raw[ROWS][MAXCH], rawp[ROWS][MAXCH];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long row = row0 + r < rows ? row0 + r : rows - 1;           // tail rows re-read the last row, never stored
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                raw[r][k] = *(const uint4*)(x + row * C + ch * 8);
                if (pre) rawp[r][k] = *(const uint4*)(pre + (row % pre_period) * C + ch * 8);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long row = row0 + r;
        float v[MAXCH][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                unpack8<T>(raw[r][k], v[k]);
                if (pre) {
                    float a[8];
                    unpack8<T>(rawp[r][k], a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[k][e] += a[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[k][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[k][e] - mean;
                    sq += d * d;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = rsqrtf(sq / (float)C + eps);
        if (row >= rows) continue;                                          // wave-uniform
        const T* po = post ? post + ((row / post_div) % post_mod) * C : nullptr;
        T* yr = y + row * C;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                float g[8], b[8], o[8];
                unpack8<T>(*(const uint4*)(gamma + ch * 8), g);
                unpack8<T>(*(const uint4*)(beta + ch * 8), b);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * g[e] + b[e];
                if (po) {
                    float a[8];
                    unpack8<T>(*(const uint4*)(po + ch * 8), a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += a[e];
                }
                *(uint4*)(yr + ch * 8) = pack8<T>(o);
            }
        }
    }
}

// Packed rows for C = 320 / 640 (NCH = 40 / 80 sixteen-byte chunks per row): a wave takes 320 consecutive chunks = 8 / 4
// whole rows, five per lane, as ONE contiguous 5 KB stream -- every lane loads (the row-per-wave form uses 40 of 64 lanes
// at C = 320) and five loads per lane are in flight.  A chunk's row is (k * 64 + lane) / NCH, so the row statistics go
// through a 320-float LDS scratch per wave: chunk sums in, NCH / 5 lanes per row add five each and finish with xor
// shuffles, row values out.  Same two-pass (mean, then centred squares) arithmetic as the kernel above.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                                const T* __restrict__ beta, const T* __restrict__ pre,
                                                                const T* __restrict__ post, T* __restrict__ y, long rows,
                                                                long pre_period, long post_div, long post_mod, float eps) {
    constexpr int C = NCH * 8, R = 320 / NCH, G = NCH / 5;          // rows per wave, lanes per row in the reduction
    static_assert(320 % NCH == 0 && NCH % 5 == 0 && 64 % G == 0 && R * G == 64, "packing");
    __shared__ float scratch[4][320 + 2 * R];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* sc = scratch[wv];
    const long row0 = ((long)blockIdx.x * 4 + wv) * R;
    if (row0 >= rows) return;
    const int nrow = rows - row0 < R ? (int)(rows - row0) : R;
    int rk[5], ck[5];                                               // row (inside the wave's group) and chunk column of slot k
    float v[5][8];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = k * 64 + lane;
        rk[k] = i / NCH;
        ck[k] = i % NCH;
        const long row = row0 + (rk[k] < nrow ? rk[k] : nrow - 1);            // tail rows re-read the last valid row, never stored
        unpack8<T>(*(const uint4*)(x + row * C + ck[k] * 8), v[k]);
        if (pre) {
            float a[8];
            unpack8<T>(*(const uint4*)(pre + (row % pre_period) * C + ck[k] * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] += a[e];
        }
    }
    // row reduction of one float per chunk: returns, for each of the lane's five slots, the total of that slot's row
    auto row_totals = [&](const float (&part)[5], float (&tot)[5]) {
#pragma unroll
        for (int k = 0; k < 5; ++k) sc[k * 64 + lane] = part[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int r = lane / G, q = lane % G;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) t += sc[r * NCH + q * 5 + j];
#pragma unroll
        for (int o = 1; o < G; o <<= 1) t += __shfl_xor(t, o);
        if (q == 0) sc[320 + r] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 5; ++k) tot[k] = sc[320 + rk[k]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                              // scratch is reused by the next reduction
    };
    float part[5], mean[5], rstd[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        part[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part[k] += v[k][e];
    }
    row_totals(part, mean);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        mean[k] *= 1.0f / (float)C;
        part[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[k][e] - mean[k];
            part[k] += d * d;
        }
    }
    row_totals(part, rstd);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        rstd[k] = rsqrtf(rstd[k] * (1.0f / (float)C) + eps);
        if (rk[k] >= nrow) continue;
        const long row = row0 + rk[k];
        float g[8], b[8], o[8];
        unpack8<T>(*(const uint4*)(gamma + ck[k] * 8), g);
        unpack8<T>(*(const uint4*)(beta + ck[k] * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean[k]) * rstd[k] * g[e] + b[e];
        if (post) {
            float a[8];
            unpack8<T>(*(const uint4*)(post + ((row / post_div) % post_mod) * C + ck[k] * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += a[e];
        }
        *(uint4*)(y + row * C + ck[k] * 8) = pack8<T>(o);
    }
}

// h [rows, 2*I] = (a | gate) -> out [rows, I] = a * gelu(gate), exact (erf) GELU
template <typename T>
__global__ void geglu_kernel(const T* __restrict__ h, T* __restrict__ out, long rows, int I8) {
    const long total = rows * I8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / I8;
        const int c = (int)(i % I8);
        const uint4* hr = (const uint4*)h + r * 2 * I8;
        float a[8], g[8];
        unpack8<T>(hr[c], a);
        unpack8<T>(hr[I8 + c], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= 0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752f));
        ((uint4*)out)[i] = pack8<T>(a);
    }
}

template <typename T>
static int launch_ln(const void* x, const void* gamma, const void* beta, const void* pre, const void* post, void* y,
                     long rows, int C, long pre_period, long post_div, long post_mod, float eps, hipStream_t s) {
    const int nch = C / 8;
#define IM360_LN(MC, R)                                                                                           \
    hipLaunchKernelGGL((layernorm_kernel<T, MC, R>), dim3((unsigned)((rows + 4 * R - 1) / (4 * R))), dim3(256), 0, \
                       s, (const T*)x, (const T*)gamma, (const T*)beta, (const T*)pre, (const T*)post, (T*)y,     \
                       rows, C, pre_period, post_div, post_mod, eps)
#define IM360_LNP(NCH)                                                                                             \
    hipLaunchKernelGGL((layernorm_packed_kernel<T, NCH>), dim3((unsigned)((rows + 4 * (320 / NCH) - 1) / (4 * (320 / NCH)))), \
                       dim3(256), 0, s, (const T*)x, (const T*)gamma, (const T*)beta, (const T*)pre, (const T*)post, \
                       (T*)y, rows, pre_period, post_div, post_mod, eps)
    // measured (tools/bench_kernels.py ln): C = 320 packed 0.173 vs 0.232 ms at 655 360 rows; C = 640 loses (0.088 vs 0.080)
    if (nch == 40 && knob(KNOB_LN_PACKED)) IM360_LNP(40);                   // (knob ln_packed = 0: A/B against the row-per-wave kernel)
    else if (nch <= 64) IM360_LN(1, 4);
    else if (nch <= 128) IM360_LN(2, 2);
    else if (nch <= 192) IM360_LN(3, 1);
    else IM360_LN(4, 1);
#undef IM360_LN
#undef IM360_LNP
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}


// ---- row softmax with a logit scale: y[r, :] = softmax(x[r, :] * scale), fp32 inside, 16-bit in / out.
//      The VAE's single-head d = 512 AttentionBlock (diffusers/models/attention.py:247-379: baddbmm -> softmax(fp32)
//      -> bmm) runs as GEMM -> this kernel -> GEMM; its score matrix is tokens x tokens of ONE image, read and written
//      once here.  One workgroup (256 threads) per row, three passes over the row held in L2 / registers.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, long cols,
                                                            long x_rs, long y_rs, float scale) {
    __shared__ float red[8];
    const long r = blockIdx.x;
    const T* xr = x + r * x_rs;
    T* yr = y + r * y_rs;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long nch = cols >> 3;
    float m = -INFINITY;
    for (long c = tid; c < nch; c += 256) {
        float f[8];
        unpack8<T>(*(const uint4*)(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, f[e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    const float s2 = scale * 1.4426950408889634f, m2 = m * 1.4426950408889634f;
    float sum = 0.f;
    for (long c = tid; c < nch; c += 256) {
        float f[8];
        unpack8<T>(*(const uint4*)(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f(fmaf(f[e], s2, -m2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (long c = tid; c < nch; c += 256) {
        float f[8];
        unpack8<T>(*(const uint4*)(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_exp2f(fmaf(f[e], s2, -m2)) * inv;
        *(uint4*)(yr + c * 8) = pack8<T>(f);
    }
}

// bias [n] (16-bit) -> fp16 [n] = bias * log2(e): the form attn_fwd's packed-bias kernels feed to the matrix pipe.
// The kernel adds it with an identity-slice MFMA (s += I * bias): a non-finite entry would meet the identity's zeros
// (0 * inf = NaN for the whole 32 x 32 score block), so -inf / finfo.min style masks are clamped to +-60000 -- exp2 of
// that is an exact 0 weight all the same; NaN inputs become the clamp value too.
template <typename T>
__global__ void attn_pack_bias_kernel(const T* __restrict__ b, _Float16* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = to_f32(b[i]) * 1.4426950408889634f;
        out[i] = (_Float16)fminf(fmaxf(v, -60000.f), 60000.f);       // (fmaxf / fminf return the finite operand for NaN)
    }
}

// Frame-sharded tokens [B, Fl, P, C] <-> the send / receive layout of the frame <-> pixel all-to-all of the motion
// modules, [W][Fl][B][PP][C] (W ranks, PP = ceil(P / W) pixels per rank, the last rank's tail zero-filled): 16-byte
// chunks, one per thread, grid-stride.  dir 0: pack (tokens -> exchange buffer), 1: unpack.
__global__ void shard_pack_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int Fl, int P, int C8,
                                  int W, int PP, int dir) {
    const long total = (long)W * Fl * B * PP * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long t = i / C8;
        const int pi = (int)(t % PP); t /= PP;
        const int b = (int)(t % B); t /= B;
        const int j = (int)(t % Fl);
        const int r = (int)(t / Fl);
        const int pix = r * PP + pi;
        const long tok = (((long)b * Fl + j) * P + pix) * C8 + c;
        if (dir == 0) dst[i] = pix < P ? src[tok] : uint4{0u, 0u, 0u, 0u};
        else if (pix < P) dst[tok] = src[i];
    }
}

}  // namespace im360

// tokens [B, Fl, P, C] (16-bit, C % 8 == 0) <-> exchange buffer [W, Fl, B, PP, C] with W * PP >= P (see the kernel).
// Replaces: nothing in the reference (it is single-process); this is the pack / unpack of the frame-chunk sharding's
// all-to-all around VersatileAttention (animatediff/models/motion_module.py:343-429), one launch instead of pad + permute +
// contiguous copies, writing into caller-owned (pre-sized) buffers so the exchange can be captured in a hipGraph.
extern "C" __attribute__((visibility("default"))) int im360_shard_pack(const void* src, void* dst, int64_t B, int64_t Fl, int64_t P, int64_t C, int64_t W,
                                int64_t PP, int dir, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(src && dst, "shard_pack: null pointer");
    IM360_CHECK_ARG(B > 0 && Fl > 0 && P > 0 && C > 0 && (C % 8) == 0 && W > 0 && PP > 0 && W * PP >= P && (W - 1) * PP < P,
                    "shard_pack: bad shape B=%ld Fl=%ld P=%ld C=%ld W=%ld PP=%ld", (long)B, (long)Fl, (long)P, (long)C, (long)W, (long)PP);
    IM360_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "shard_pack: misaligned pointer");
    IM360_CHECK_ARG(dir == 0 || dir == 1, "shard_pack: dir must be 0 (pack) or 1 (unpack)");
    const long total = W * Fl * B * PP * (C / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(shard_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       (int)B, (int)Fl, (int)P, (int)(C / 8), (int)W, (int)PP, dir);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// y[r] = LN(x[r] + pre[r % pre_period]) * gamma + beta + post[(r / post_div) % post_mod]; pre/post optional
extern "C" __attribute__((visibility("default"))) int im360_layernorm(const void* x, const void* gamma, const void* beta, const void* pre, const void* post,
                               void* y, int64_t rows, int64_t C, int64_t pre_period, int64_t post_div,
                               int64_t post_mod, float eps, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
    IM360_CHECK_ARG(rows > 0 && C > 0 && (C % 8) == 0 && C <= 2048, "layernorm: C=%ld must be a multiple of 8, <= 2048", (long)C);
    IM360_CHECK_ARG(!pre || pre_period > 0, "layernorm: pre_period must be positive");
    IM360_CHECK_ARG(!post || (post_div > 0 && post_mod > 0), "layernorm: post_div/post_mod must be positive");
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                    ((uintptr_t)beta % 16) == 0, "layernorm: misaligned pointer");
    IM360_CHECK_ARG((rows + 3) / 4 <= 0x7fffffffL, "layernorm: too many rows");
    ProfScope prof(PROF_MISC, stream);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) return launch_ln<__bf16>(x, gamma, beta, pre, post, y, rows, (int)C, pre_period, post_div, post_mod, eps, s);
    if (dtype == 1) return launch_ln<_Float16>(x, gamma, beta, pre, post, y, rows, (int)C, pre_period, post_div, post_mod, eps, s);
    im360_set_error("layernorm: dtype %d unsupported", dtype);
    return IM360_ERR_UNSUPPORTED;
}

// h [rows, 2*I] -> out [rows, I] = h[:, :I] * gelu(h[:, I:])
extern "C" __attribute__((visibility("default"))) int im360_geglu(const void* h, void* out, int64_t rows, int64_t I, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(h && out, "geglu: null pointer");
    IM360_CHECK_ARG(rows > 0 && I > 0 && (I % 8) == 0, "geglu: I=%ld must be a multiple of 8", (long)I);
    IM360_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0, "geglu: misaligned pointer");
    const long total = rows * (I / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    ProfScope prof(PROF_MISC, stream);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((geglu_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)h, (__bf16*)out, (long)rows, (int)(I / 8));
    else if (dtype == 1)
        hipLaunchKernelGGL((geglu_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, (const _Float16*)h, (_Float16*)out, (long)rows, (int)(I / 8));
    else {
        im360_set_error("geglu: dtype %d unsupported", dtype);
        return IM360_ERR_UNSUPPORTED;
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// y[r, c] = softmax_c(x[r, c] * scale) over `cols` (a multiple of 8) columns of `rows` rows; row strides in elements
// (multiples of 8); fp32 arithmetic.  x == y (in place) is allowed.
extern "C" __attribute__((visibility("default"))) int im360_softmax_rows(const void* x, void* y, int64_t rows, int64_t cols, int64_t x_rs, int64_t y_rs,
                                  float scale, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && y, "softmax_rows: null pointer");
    IM360_CHECK_ARG(rows > 0 && cols > 0 && (cols % 8) == 0 && (x_rs % 8) == 0 && (y_rs % 8) == 0 && x_rs >= cols && y_rs >= cols,
                    "softmax_rows: cols / row strides must be positive multiples of 8");
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "softmax_rows: misaligned pointer");
    IM360_CHECK_ARG(rows <= 0x7fffffffL, "softmax_rows: too many rows");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_MISC, stream);
    if (dtype == 0)
        hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, s, (const __bf16*)x, (__bf16*)y,
                           (long)rows, (long)cols, (long)x_rs, (long)y_rs, scale);
    else if (dtype == 1)
        hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, s, (const _Float16*)x,
                           (_Float16*)y, (long)rows, (long)cols, (long)x_rs, (long)y_rs, scale);
    else {
        im360_set_error("softmax_rows: dtype %d unsupported", dtype);
        return IM360_ERR_UNSUPPORTED;
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// out_f16[i] = fp16(bias[i] * log2(e)), i < n: once per bias matrix (they are cached per resolution)
extern "C" __attribute__((visibility("default"))) int im360_attn_pack_bias(const void* bias, void* out_f16, int64_t n, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(bias && out_f16 && n > 0, "attn_pack_bias: null pointer / empty");
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL((attn_pack_bias_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)bias, (_Float16*)out_f16, (long)n);
    else if (dtype == 1) hipLaunchKernelGGL((attn_pack_bias_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, (const _Float16*)bias, (_Float16*)out_f16, (long)n);
    else {
        im360_set_error("attn_pack_bias: dtype %d unsupported", dtype);
        return IM360_ERR_UNSUPPORTED;
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}
