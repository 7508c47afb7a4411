// GroupNorm (+ SiLU, + equirectangular circular pad) on channels-last 16-bit activations, gfx950.
//
// Replaces InflatedGroupNorm / nn.GroupNorm + SiLU call sites (animatediff/models/resnet.py:9-17,
// 224-225, 236-243; Transformer3DModel.norm attention.py:206,262; TemporalTransformer3DModel.norm
// motion_module.py:128,169; conv_norm_out unet.py:350-356; the VAE norms) and fuses the reference's
// pad_pano (src/utils/pano.py:75-95) into the apply pass.
//
// Parity trap reproduced on purpose: in the pano branch the reference pads BEFORE the ResnetBlock, so
// norm1 statistics are taken over the circularly padded tensor (the `pad` wrapped columns count twice,
// MVGenModel.py:277-278).  `pad` > 0 gives those columns weight 2 in the statistics and makes the
// apply pass write the W + 2*pad wide padded tensor directly.
//
// Three HBM-bound passes: (1) per-(image, slab) per-channel partial sums, (2) finalize per-(image,
// channel) scale/shift in fp64 from the partials, (3) y = act(x * scale + shift).
#include "common.h"

namespace im360 {

// ---- pass 1 -----------------------------------------------------------------------------------
// grid (S, N), 256 threads.  partial[n][s][0][c] = sum w*x, partial[n][s][1][c] = sum w*x^2
// Channel concatenations that are never materialised (the decoder's skip connections): the kernel is launched once per
// source tensor; C = channels of THIS tensor (its pixel stride), Ct = channels of the concatenation (the layout of
// `partial`), coff = where this tensor's channels start in it.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                          int HW, int W, int C, int pad, int S, int Ct, int coff) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int nch = C >> 3;
    const int pps = (HW + S - 1) / S;
    const int pix0 = s * pps, pix1 = min(HW, pix0 + pps);
    const T* xb = x + (long)n * HW * C;
    for (int c0 = 0; c0 < nch; c0 += 256) {
        const int ncp = min(256, nch - c0);
        const int lanes = 256 / ncp;
        const int cc = tid % ncp, pl = tid / ncp;
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (pl < lanes) {
            const T* xc = xb + (c0 + cc) * 8;
            auto add = [&](const uint4& raw, int pix) {
                float f[8];
                unpack8<T>(raw, f);
                float wgt = 1.f;
                if (pad > 0) {
                    const int xx = pix % W;
                    if (xx < pad || xx >= W - pad) wgt = 2.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc[e] += wgt * f[e];
                    acc[8 + e] += wgt * f[e] * f[e];
                }
            };
            // four independent 16-byte loads in flight per thread: one is not enough to cover the HBM latency
            // (8 workgroups x 256 threads x 16 B = 32 KB per CU in flight gave ~3.5 TB/s)
            int pix = pix0 + pl;
            for (; pix + 3 * lanes < pix1; pix += 4 * lanes) {
                const uint4 r0 = *(const uint4*)(xc + (long)pix * C);
                const uint4 r1 = *(const uint4*)(xc + (long)(pix + lanes) * C);
                const uint4 r2 = *(const uint4*)(xc + (long)(pix + 2 * lanes) * C);
                const uint4 r3 = *(const uint4*)(xc + (long)(pix + 3 * lanes) * C);
                add(r0, pix);
                add(r1, pix + lanes);
                add(r2, pix + 2 * lanes);
                add(r3, pix + 3 * lanes);
            }
            for (; pix < pix1; pix += lanes) add(*(const uint4*)(xc + (long)pix * C), pix);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[tid * 16 + e] = acc[e];
        __syncthreads();
        if (pl == 0) {
            for (int l = 1; l < lanes; ++l)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] += red[(l * ncp + cc) * 16 + e];
            float* dst = partial + ((long)(n * S + s) * 2) * Ct + coff + (c0 + cc) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dst[e] = acc[e];
                dst[Ct + e] = acc[8 + e];
            }
        }
    }
}

// ---- pass 2 -----------------------------------------------------------------------------------
// one wave per (image, group): the S x (C / G) partial sums of a group are read ONCE (coalesced over the group's
// channels), reduced in fp64 across the wave, and the wave writes scale / shift of the group's channels.
template <typename T>
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, const T* __restrict__ gamma,
                                                         const T* __restrict__ beta, float* __restrict__ scale,
                                                         float* __restrict__ shift, int N, int C, int G, int S,
                                                         double count, float eps) {
    const int n = blockIdx.x / G, g = blockIdx.x % G, lane = threadIdx.x;
    const int cpg = C / G;
    double sum = 0.0, sq = 0.0;
    const float* base = partial + (long)n * S * 2 * C + g * cpg;
    for (int i = lane; i < S * cpg; i += 64) {
        const int s = i / cpg, j = i % cpg;
        const float* src = base + (long)s * 2 * C + j;
        sum += (double)src[0];
        sq += (double)src[C];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        sq += __shfl_xor(sq, o);
    }
    const double mean = sum / count;
    double var = sq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int j = lane; j < cpg; j += 64) {
        const int c = g * cpg + j;
        const double ga = (double)to_f32(gamma[c]), be = (double)to_f32(beta[c]);
        scale[(long)n * C + c] = (float)(rstd * ga);
        shift[(long)n * C + c] = (float)(be - mean * rstd * ga);
    }
}

// The same from partial sums that need not come from gn_partial_kernel: the channels [0, C1) from `pa` (Sa slabs per image,
// [n][s][2][C1]) and [C1, C1 + C2) from `pb` (Sb slabs, [n][s][2][C2]) -- either buffer may have been written by the epilogue
// of the convolution / linear that produced that tensor (ConvParams::gn_out: one slab per 256-pixel tile).
// scale / shift of every channel of image n from partial sums in two buffers, by one 256-thread workgroup -- the ONE reduction
// order of this file's two-source paths (gn_finalize2_kernel, gn_apply_p_kernel), all in fp64:
//   (A) thread per channel: the channel's sums over the slabs in ascending slab order (coalesced across the threads, four
//       independent loads in flight) -> LDS;
//   (B) eight threads per group: thread k adds the channels k, k + 8, ... of the group in ascending order, then a butterfly over the
//       eight; mean / rstd of the group -> LDS;
//   (C) thread per channel: put(c, rstd * gamma, beta - mean * rstd * gamma).
// `wk`: (2 C + 2 G) doubles of LDS.  Ends with the LDS free again only after the caller's next barrier (put may write LDS that
// overlays `wk`'s channel sums: they are dead after (B)'s barrier).
template <typename T, typename Put>
__device__ __forceinline__ void gn_image_scale_shift(const float* __restrict__ pa, int Sa, int C1, const float* __restrict__ pb, int Sb, int C2,
                                                     const T* __restrict__ gamma, const T* __restrict__ beta, int n, int G, double count, float eps,
                                                     double* wk, Put put) {
    const int tid = threadIdx.x;
    const int C = C1 + C2, cpg = C / G;
    double* const gst = wk + 2 * C;
    for (int c = tid; c < C; c += 256) {
        const bool second = c >= C1;
        const int S = second ? Sb : Sa, Cx = second ? C2 : C1;
        const float* src = (second ? pb + (long)n * Sb * 2 * C2 + (c - C1) : pa + (long)n * Sa * 2 * C1 + c);
        double sum = 0.0, sq = 0.0;
        int sl = 0;
        for (; sl + 4 <= S; sl += 4) {
            float u[4], v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = src[(long)(sl + i) * 2 * Cx];
                v[i] = src[(long)(sl + i) * 2 * Cx + Cx];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sum += (double)u[i];
                sq += (double)v[i];
            }
        }
        for (; sl < S; ++sl) {
            sum += (double)src[(long)sl * 2 * Cx];
            sq += (double)src[(long)sl * 2 * Cx + Cx];
        }
        wk[c] = sum;
        wk[C + c] = sq;
    }
    __syncthreads();
    for (int g = tid >> 3; g < ((G + 31) & ~31); g += 32) {          // (whole waves run the butterfly)
        double sum = 0.0, sq = 0.0;
        if (g < G)
            for (int j = tid & 7; j < cpg; j += 8) {
                sum += wk[g * cpg + j];
                sq += wk[C + g * cpg + j];
            }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            sum += __shfl_xor(sum, o);
            sq += __shfl_xor(sq, o);
        }
        if (g < G && (tid & 7) == 0) {
            const double mean = sum / count;
            double var = sq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            gst[2 * g] = mean;
            gst[2 * g + 1] = 1.0 / sqrt(var + (double)eps);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const double mean = gst[2 * g], rstd = gst[2 * g + 1];
        const double ga = (double)to_f32(gamma[c]), be = (double)to_f32(beta[c]);
        put(c, (float)(rstd * ga), (float)(be - mean * rstd * ga));
    }
}

// grid N, 256 threads, (2 C + 2 G) doubles of dynamic LDS
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize2_kernel(const float* __restrict__ pa, int Sa, int C1, const float* __restrict__ pb, int Sb, int C2,
                                                           const T* __restrict__ gamma, const T* __restrict__ beta, float* __restrict__ scale,
                                                           float* __restrict__ shift, int G, double count, float eps) {
    extern __shared__ double gn_wk[];
    const int C = C1 + C2;
    const int n = blockIdx.x;
    gn_image_scale_shift<T>(pa, Sa, C1, pb, Sb, C2, gamma, beta, n, G, count, eps, gn_wk, [&](int c, float sc, float sh) {
        scale[(long)n * C + c] = sc;
        shift[(long)n * C + c] = sh;
    });
}

// ---- pass 3 -----------------------------------------------------------------------------------
// grid (S, N).  out[n][y][x'][c] = act(x[n][y][(x' - pad) mod W][c] * scale[n][c] + shift[n][c]),
// x' in [0, W + 2 pad)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, T* __restrict__ y,
                                                        int H, int W, int C, int pad, int act, int S, int Ct, int coff) {
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int nch = C >> 3;
    const int Wo = W + 2 * pad;
    const int HWo = H * Wo;
    const int pps = (HWo + S - 1) / S;
    const int pix0 = s * pps, pix1 = min(HWo, pix0 + pps);
    const T* xb = x + (long)n * H * W * C;
    T* yb = y + (long)n * HWo * Ct + coff;         // (C, Ct, coff as in gn_partial_kernel: y and scale / shift span the concatenation)
    for (int c0 = 0; c0 < nch; c0 += 256) {
        const int ncp = min(256, nch - c0);
        const int lanes = 256 / ncp;
        const int cc = tid % ncp, pl = tid / ncp;
        if (pl >= lanes) continue;
        float sc[8], sh[8];
        const float* sp = scale + (long)n * Ct + coff + (c0 + cc) * 8;
        const float* hp = shift + (long)n * Ct + coff + (c0 + cc) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = sp[e];
            sh[e] = hp[e];
        }
        const T* xc = xb + (c0 + cc) * 8;
        T* yc = yb + (c0 + cc) * 8;
        auto src = [&](int pix) {
            int sx = pix % Wo - pad;
            const int sy = pix / Wo;
            if (sx < 0) sx += W;
            else if (sx >= W) sx -= W;
            return *(const uint4*)(xc + ((long)sy * W + sx) * C);
        };
        auto emit = [&](const uint4& raw, int pix) {
            float f[8];
            unpack8<T>(raw, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = f[e] * sc[e] + sh[e];
                f[e] = act ? silu_f(v) : v;
            }
            *(uint4*)(yc + (long)pix * Ct) = pack8<T>(f);
        };
        int pix = pix0 + pl;                       // four loads in flight per thread (see gn_partial_kernel)
        for (; pix + 3 * lanes < pix1; pix += 4 * lanes) {
            const uint4 r0 = src(pix), r1 = src(pix + lanes), r2 = src(pix + 2 * lanes), r3 = src(pix + 3 * lanes);
            emit(r0, pix);
            emit(r1, pix + lanes);
            emit(r2, pix + 2 * lanes);
            emit(r3, pix + 3 * lanes);
        }
        for (; pix < pix1; pix += lanes) emit(src(pix), pix);
    }
}

// ---- passes 2 + 3 in ONE launch (the default since round 6) ---------------------------------------------------------------------
// grid (S, N).  Every workgroup first rebuilds scale / shift of ITS image from the partial sums (a few KB per image out of the L2;
// gn_image_scale_shift: the bits of gn_finalize2_kernel) into LDS, then normalises its slab of the
// output -- of x, or of the never-materialised concatenation [xa | xb] -- as gn_apply_kernel does.  The finalize launch (157 per
// denoising step, each a ~10 us chain of fp64 latencies on N x G single-wave workgroups) and the [N, C] scale / shift round trip
// are gone; the concatenation takes one launch instead of two.  NLD: 16-byte loads in flight per thread.
template <typename T, int NLD, bool NTS, bool NTL = false>
__global__ __launch_bounds__(256) void gn_apply_p_kernel(const T* __restrict__ xa, const T* __restrict__ xb, int C1, int C2,
                                                          const float* __restrict__ pa, int Sa, const float* __restrict__ pb, int Sb,
                                                          const T* __restrict__ gamma, const T* __restrict__ beta, T* __restrict__ y,
                                                          int H, int W, int G, int pad, int act, int S, float eps) {
    extern __shared__ double gn_wk[];                // (2 C + 2 G) doubles; then, over the dead channel sums, [2][C] floats: scale | shift of image n
    float* const sc_sh = (float*)gn_wk;
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int C = C1 + C2;
    gn_image_scale_shift<T>(pa, Sa, C1, pb, Sb, C2, gamma, beta, n, G, (double)H * (double)(W + 2 * pad) * (double)(C / G), eps, gn_wk,
                            [&](int c, float sc, float sh) {
                                sc_sh[c] = sc;
                                sc_sh[C + c] = sh;
                            });
    __syncthreads();
    const int Wo = W + 2 * pad, HWo = H * Wo;
    const int pps = (HWo + S - 1) / S;
    const int pix0 = s * pps, pix1 = min(HWo, pix0 + pps);
    T* yb = y + (long)n * HWo * C;
    for (int src = 0; src < (xb ? 2 : 1); ++src) {
        const T* x = src ? xb : xa;
        const int Cx = src ? C2 : C1, coff = src ? C1 : 0;
        const int nch = Cx >> 3;
        const T* xbase = x + (long)n * H * W * Cx;
        for (int c0 = 0; c0 < nch; c0 += 256) {
            const int ncp = min(256, nch - c0);
            const int lanes = 256 / ncp;
            const int cc = tid % ncp, pl = tid / ncp;
            if (pl >= lanes) continue;
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = sc_sh[coff + (c0 + cc) * 8 + e];
                sh[e] = sc_sh[C + coff + (c0 + cc) * 8 + e];
            }
            const T* xc = xbase + (c0 + cc) * 8;
            T* yc = yb + coff + (c0 + cc) * 8;
            // (PADDED = false: output pixel == source pixel -- no pix / Wo, pix % Wo per 16-byte piece: the two integer divisions were ~ 60 of the
            //  ~ 140 vector-issue slots of a piece with SiLU, which put the kernel's VALU ceiling (~ 8 TB/s) right at the HBM roofline)
            auto srcp = [&](int pix, auto padded_tag) -> uint4 {
                long off;
                if constexpr (decltype(padded_tag)::value) {
                    int sx = pix % Wo - pad;
                    const int sy = pix / Wo;
                    if (sx < 0) sx += W;
                    else if (sx >= W) sx -= W;
                    off = ((long)sy * W + sx) * Cx;
                } else {
                    off = (long)pix * Cx;
                }
                if constexpr (NTL) {
                    const u32x4 r = __builtin_nontemporal_load((const u32x4*)(xc + off));
                    return uint4{r.x, r.y, r.z, r.w};
                } else {
                    return *(const uint4*)(xc + off);
                }
            };
            // (ACT as a tag: with the run-time `act` hipcc branched around the SiLU of EVERY element -- 32 scalar branches per four pieces, each
            //  exp / rcp chain alone behind its branch with s_nop wait states; as straight-line code the eight chains of a piece interleave)
            auto emit = [&](const uint4& raw, int pix, auto act_tag) {
                float f[8];
                unpack8<T>(raw, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = f[e] * sc[e] + sh[e];
                    f[e] = decltype(act_tag)::value ? silu_f(v) : v;
                }
                const uint4 o = pack8<T>(f);
                if constexpr (NTS) __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, (u32x4*)(yc + (long)pix * C));
                else *(uint4*)(yc + (long)pix * C) = o;
            };
            auto run = [&](auto padded_tag, auto act_tag) {
                int pix = pix0 + pl;
                for (; pix + (NLD - 1) * lanes < pix1; pix += NLD * lanes) {
                    uint4 r[NLD];
#pragma unroll
                    for (int i = 0; i < NLD; ++i) r[i] = srcp(pix + i * lanes, padded_tag);
#pragma unroll
                    for (int i = 0; i < NLD; ++i) emit(r[i], pix + i * lanes, act_tag);
                }
                for (; pix < pix1; pix += lanes) emit(srcp(pix, padded_tag), pix, act_tag);
            };
            if (pad == 0) {
                if (act) run(std::false_type{}, std::true_type{});
                else run(std::false_type{}, std::false_type{});
            } else {
                if (act) run(std::true_type{}, std::true_type{});
                else run(std::true_type{}, std::false_type{});
            }
        }
    }
}

// ---- all three passes in ONE launch -------------------------------------------------------------------------------
// Workgroup (image n, slab s): (1) the slab's per-channel partial sums -> partial[n][s] (exactly gn_partial_kernel's), a
// release fence and one atomic increment of counter[n]; (2) wait until all S slabs of the image have arrived, then every
// workgroup reduces the image's partials itself, in gn_finalize_kernel's fixed order and in fp64 (so scale / shift are
// bit-identical to the three-kernel path), into LDS; (3) act(x * scale + shift) on its slab of the output.  The second
// read of the slab comes out of the L2 / Infinity Cache a few microseconds after the first instead of from HBM one
// whole-tensor pass later: 2 HBM passes instead of 3, one launch instead of three.  Liveness: workgroups are dispatched in
// block-id order (image-major), so a waiting workgroup only ever waits for ids that are resident or will be as soon as
// an EARLIER image finishes; nothing waits on a later image.
template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(const T* __restrict__ xa, const T* __restrict__ xb, int C1, int C2,
                                                        float* __restrict__ partial, int* __restrict__ counter,
                                                        const T* __restrict__ gamma, const T* __restrict__ beta,
                                                        T* __restrict__ y, int H, int W, int G, int pad, int act, int S,
                                                        float eps, int cmax) {
    __shared__ float red[256 * 16];
    extern __shared__ float sc_sh[];                 // [2][C]: scale | shift of image n
    const int n = blockIdx.x / S, s = blockIdx.x % S, tid = threadIdx.x;
    const int HW = H * W, C = C1 + C2;
    // ---- (1) partial sums of this slab, both sources
    {
        const int pps = (HW + S - 1) / S;
        const int pix0 = s * pps, pix1 = min(HW, pix0 + pps);
        for (int src = 0; src < (xb ? 2 : 1); ++src) {
            const T* x = src ? xb : xa;
            const int Cx = src ? C2 : C1, coff = src ? C1 : 0;
            const int nch = Cx >> 3;
            const T* xbase = x + (long)n * HW * Cx;
            for (int c0 = 0; c0 < nch; c0 += 256) {
                const int ncp = min(256, nch - c0);
                const int lanes = 256 / ncp;
                const int cc = tid % ncp, pl = tid / ncp;
                float acc[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                if (pl < lanes) {
                    const T* xc = xbase + (c0 + cc) * 8;
                    auto add = [&](const uint4& raw, int pix) {
                        float f[8];
                        unpack8<T>(raw, f);
                        float wgt = 1.f;
                        if (pad > 0) {
                            const int xx = pix % W;
                            if (xx < pad || xx >= W - pad) wgt = 2.f;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            acc[e] += wgt * f[e];
                            acc[8 + e] += wgt * f[e] * f[e];
                        }
                    };
                    int pix = pix0 + pl;
                    for (; pix + 3 * lanes < pix1; pix += 4 * lanes) {
                        const uint4 r0 = *(const uint4*)(xc + (long)pix * Cx);
                        const uint4 r1 = *(const uint4*)(xc + (long)(pix + lanes) * Cx);
                        const uint4 r2 = *(const uint4*)(xc + (long)(pix + 2 * lanes) * Cx);
                        const uint4 r3 = *(const uint4*)(xc + (long)(pix + 3 * lanes) * Cx);
                        add(r0, pix);
                        add(r1, pix + lanes);
                        add(r2, pix + 2 * lanes);
                        add(r3, pix + 3 * lanes);
                    }
                    for (; pix < pix1; pix += lanes) add(*(const uint4*)(xc + (long)pix * Cx), pix);
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; ++e) red[tid * 16 + e] = acc[e];
                __syncthreads();
                if (pl == 0) {
                    for (int l = 1; l < lanes; ++l)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] += red[(l * ncp + cc) * 16 + e];
                    // published WRITE-THROUGH (relaxed agent-scope stores = `sc1`): the consumers are other workgroups, possibly
                    // on another XCD (private L2); a release fence instead would write back the whole L2 -- full of other
                    // workgroups' freshly normalised output -- once per workgroup (measured: 0.91 ms against 0.26 ms)
                    float* dst = partial + ((long)(n * S + s) * 2) * C + coff + (c0 + cc) * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        __hip_atomic_store(dst + e, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + C + e, acc[8 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
    // ---- arrive (every write-through store of this workgroup has been acknowledged: vmcnt(0) in each thread, then the
    //      barrier), wait for the image's other slabs.  No fences: the partials are read back with `sc1` loads below.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    // ---- (2) scale / shift of every channel of image n: one wave per group at a time, gn_finalize_kernel's order
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int cpg = C / G;
        const double count = (double)H * (double)(W + 2 * pad) * (double)cpg;
        for (int g = wave; g < G; g += 4) {
            double sum = 0.0, sq = 0.0;
            const float* base = partial + (long)n * S * 2 * C + g * cpg;
            for (int i = lane; i < S * cpg; i += 64) {
                const int sl = i / cpg, j = i % cpg;
                const float* src = base + (long)sl * 2 * C + j;
                sum += (double)__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sq += (double)__hip_atomic_load(src + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sum += __shfl_xor(sum, o);
                sq += __shfl_xor(sq, o);
            }
            const double mean = sum / count;
            double var = sq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            for (int j = lane; j < cpg; j += 64) {
                const int c = g * cpg + j;
                const double ga = (double)to_f32(gamma[c]), be = (double)to_f32(beta[c]);
                sc_sh[c] = (float)(rstd * ga);
                sc_sh[cmax + c] = (float)(be - mean * rstd * ga);
            }
        }
    }
    __syncthreads();
    // ---- (3) apply on this workgroup's slab of the (W + 2 pad wide) output, both sources
    {
        const int Wo = W + 2 * pad, HWo = H * Wo;
        const int pps = (HWo + S - 1) / S;
        const int pix0 = s * pps, pix1 = min(HWo, pix0 + pps);
        T* yb = y + (long)n * HWo * C;
        for (int src = 0; src < (xb ? 2 : 1); ++src) {
            const T* x = src ? xb : xa;
            const int Cx = src ? C2 : C1, coff = src ? C1 : 0;
            const int nch = Cx >> 3;
            const T* xbase = x + (long)n * HW * Cx;
            for (int c0 = 0; c0 < nch; c0 += 256) {
                const int ncp = min(256, nch - c0);
                const int lanes = 256 / ncp;
                const int cc = tid % ncp, pl = tid / ncp;
                if (pl >= lanes) continue;
                float sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = sc_sh[coff + (c0 + cc) * 8 + e];
                    sh[e] = sc_sh[cmax + coff + (c0 + cc) * 8 + e];
                }
                const T* xc = xbase + (c0 + cc) * 8;
                T* yc = yb + coff + (c0 + cc) * 8;
                auto srcp = [&](int pix) {
                    int sx = pix % Wo - pad;
                    const int sy = pix / Wo;
                    if (sx < 0) sx += W;
                    else if (sx >= W) sx -= W;
                    return *(const uint4*)(xc + ((long)sy * W + sx) * Cx);
                };
                auto emit = [&](const uint4& raw, int pix) {
                    float f[8];
                    unpack8<T>(raw, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = f[e] * sc[e] + sh[e];
                        f[e] = act ? silu_f(v) : v;
                    }
                    *(uint4*)(yc + (long)pix * C) = pack8<T>(f);
                };
                int pix = pix0 + pl;
                for (; pix + 3 * lanes < pix1; pix += 4 * lanes) {
                    const uint4 r0 = srcp(pix), r1 = srcp(pix + lanes), r2 = srcp(pix + 2 * lanes), r3 = srcp(pix + 3 * lanes);
                    emit(r0, pix);
                    emit(r1, pix + lanes);
                    emit(r2, pix + 2 * lanes);
                    emit(r3, pix + 3 * lanes);
                }
                for (; pix < pix1; pix += lanes) emit(srcp(pix), pix);
            }
        }
    }
}

// ---- circular pad along W of a channels-last tensor (latent boundary of the VAE decode,
//      pipeline_animation_inference_dual.py:349-357, 813-815) ----------------------------------------
template <typename T>
__global__ void circular_pad_w_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int W, int C8, int pad) {
    const int Wo = W + 2 * pad;
    const long total = rows * Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        const long t = i / C8;
        int sx = (int)(t % Wo) - pad;
        const long row = t / Wo;
        sx %= W;
        if (sx < 0) sx += W;
        ((uint4*)y)[i] = ((const uint4*)x)[(row * W + sx) * C8 + c];
    }
}

// ---- circular pad of the last two axes of a W-last tensor (torch.nn.functional.pad(x, (l, r, t, b), "circular")):
//      the 360-degree close-loop patch of the super-resolution stage (sr/video_to_video_model.py:16-29, 99, 160-162 via
//      src/utils/pano.py:75-101) and any NCHW pad_pano.  Works on units of U bytes (the host picks the widest U in
//      {16, 8, 4, 2} that divides the row, both W pads and the pointers), one unit per thread, rows x units grid-stride.
template <typename U>
__global__ void circular_pad_hw_kernel(const U* __restrict__ x, U* __restrict__ y, long N, int H, int W, int left,
                                       int right, int top, int bottom) {
    const int Wo = W + left + right, Ho = H + top + bottom;
    const long total = N * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        const long t = i / Wo;
        const int yo = (int)(t % Ho);
        const long n = t / Ho;
        int sx = xo - left, sy = yo - top;
        if (sx < 0) sx += W;
        else if (sx >= W) sx -= W;
        if (sy < 0) sy += H;
        else if (sy >= H) sy -= H;
        y[i] = x[(n * H + sy) * W + sx];
    }
}

// ---- CFG combine + DDIM v-prediction update (eta = 0), one elementwise pass
//      (pipeline_animation_inference_dual.py:791-800; diffusers/schedulers/scheduling_ddim.py:300-350):
//      v = u + g (c - u);  x_prev = cx * x + cv * v   with cx, cv precomputed on the host in fp64
template <typename T>
__global__ void cfg_ddim_kernel(const T* __restrict__ uncond, const T* __restrict__ cond, const T* __restrict__ x,
                                T* __restrict__ out, long n8, float g, float cx, float cv, const float* __restrict__ coef) {
    if (coef != nullptr) {          // (guidance, cx, cv) read on the device: the launch can be replayed from a hipGraph
        g = coef[0];
        cx = coef[1];
        cv = coef[2];
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float u[8], c[8], s[8];
        unpack8<T>(((const uint4*)uncond)[i], u);
        unpack8<T>(((const uint4*)cond)[i], c);
        unpack8<T>(((const uint4*)x)[i], s);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = cx * s[e] + cv * (u[e] + g * (c[e] - u[e]));
        ((uint4*)out)[i] = pack8<T>(s);
    }
}

static inline int pick_slabs(long N, long HW) {
    long s = (2048 + N - 1) / N;
    const long maxs = HW / 64 > 0 ? HW / 64 : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace im360

extern "C" __attribute__((visibility("default"))) int64_t im360_gn_num_slabs(int64_t N, int64_t H, int64_t W) { return im360::pick_slabs(N, H * W); }

namespace im360 {
template <typename T>
static void launch_gn_stats(const void* xa, const void* xb, int64_t C1, int64_t C2, const void* gamma, const void* beta, void* partial,
                            void* scale, void* shift, int64_t N, int64_t H, int64_t W, int64_t G, int64_t pad, float eps, hipStream_t s) {
    const int S = pick_slabs(N, H * W);
    const int Ct = (int)(C1 + C2);
    const double count = (double)H * (double)(W + 2 * pad) * (double)(Ct / G);
    hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(S, (unsigned)N), dim3(256), 0, s, (const T*)xa, (float*)partial,
                       (int)(H * W), (int)W, (int)C1, (int)pad, S, Ct, 0);
    if (xb)
        hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(S, (unsigned)N), dim3(256), 0, s, (const T*)xb, (float*)partial,
                           (int)(H * W), (int)W, (int)C2, (int)pad, S, Ct, (int)C1);
    hipLaunchKernelGGL((gn_finalize_kernel<T>), dim3((unsigned)(N * G)), dim3(64), 0, s, (const float*)partial, (const T*)gamma,
                       (const T*)beta, (float*)scale, (float*)shift, (int)N, Ct, (int)G, S, count, eps);
}
template <typename T>
static void launch_gn_apply(const void* xa, const void* xb, int64_t C1, int64_t C2, const void* scale, const void* shift, void* y,
                            int64_t N, int64_t H, int64_t W, int64_t pad, int act, hipStream_t s) {
    const int S = pick_slabs(N, H * (W + 2 * pad));
    const int Ct = (int)(C1 + C2);
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(S, (unsigned)N), dim3(256), 0, s, (const T*)xa, (const float*)scale,
                       (const float*)shift, (T*)y, (int)H, (int)W, (int)C1, (int)pad, act, S, Ct, 0);
    if (xb)
        hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(S, (unsigned)N), dim3(256), 0, s, (const T*)xb, (const float*)scale,
                           (const float*)shift, (T*)y, (int)H, (int)W, (int)C2, (int)pad, act, S, Ct, (int)C1);
}
}  // namespace im360

// x [N,H,W,C]; partial: fp32 workspace of N * S * 2 * C floats with S = im360_gn_num_slabs(N,H,W);
// scale, shift: fp32 [N, C] outputs.
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_stats(const void* x, const void* gamma, const void* beta, void* partial,
                                     void* scale, void* shift, int64_t N, int64_t H, int64_t W, int64_t C,
                                     int64_t G, int64_t pad, float eps, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && gamma && beta && partial && scale && shift, "groupnorm_stats: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && G > 0, "groupnorm_stats: empty problem");
    IM360_CHECK_ARG((C % 8) == 0 && (C % G) == 0, "groupnorm_stats: C=%ld must be a multiple of 8 and of G=%ld", (long)C, (long)G);
    IM360_CHECK_ARG(pad >= 0 && 2 * pad <= W, "groupnorm_stats: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0, "groupnorm_stats: misaligned x");
    IM360_CHECK_ARG(N <= 65535, "groupnorm_stats: N=%ld exceeds grid.y", (long)N);
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_stats: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_STATS, stream);
    if (dtype == 0) launch_gn_stats<__bf16>(x, nullptr, C, 0, gamma, beta, partial, scale, shift, N, H, W, G, pad, eps, (hipStream_t)stream);
    else launch_gn_stats<_Float16>(x, nullptr, C, 0, gamma, beta, partial, scale, shift, N, H, W, G, pad, eps, (hipStream_t)stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// Per-slab partial sums only (no finalize, no pad weighting): partial fp32 [N][S][2][C], S = im360_gn_num_slabs(N, H, W).
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_partial(const void* x, void* partial, int64_t N, int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && partial, "groupnorm_partial: null pointer");
    IM360_CHECK_ARG(N > 0 && N <= 65535 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "groupnorm_partial: bad shape");
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0, "groupnorm_partial: misaligned x");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_partial: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_STATS, stream);
    const int S = pick_slabs(N, H * W);
    if (dtype == 0) hipLaunchKernelGGL((gn_partial_kernel<__bf16>), dim3(S, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (float*)partial, (int)(H * W), (int)W, (int)C, 0, S, (int)C, 0);
    else hipLaunchKernelGGL((gn_partial_kernel<_Float16>), dim3(S, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (float*)partial, (int)(H * W), (int)W, (int)C, 0, S, (int)C, 0);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// scale / shift [N, C1 + C2] from partial sums: channels [0, C1) from pa ([N][Sa][2][C1]), [C1, C1 + C2) from pb ([N][Sb][2][C2];
// pb may be null with C2 = 0).  The partial sums come from im360_groupnorm_partial or from the epilogue of the kernel that
// produced the tensor (im360_conv_fwd / im360_linear_fwd, gn_partial).  count = H * W pixels per image.
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_finalize(const void* pa, int64_t Sa, int64_t C1, const void* pb, int64_t Sb, int64_t C2, const void* gamma,
                                        const void* beta, void* scale, void* shift, int64_t N, int64_t G, int64_t pixels, float eps,
                                        int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(pa && gamma && beta && scale && shift && (pb || C2 == 0), "groupnorm_finalize: null pointer");
    IM360_CHECK_ARG(N > 0 && G > 0 && C1 > 0 && C2 >= 0 && Sa > 0 && (C2 == 0 || Sb > 0) && pixels > 0 && ((C1 + C2) % G) == 0, "groupnorm_finalize: bad shape");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_finalize: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_STATS, stream);
    const double count = (double)pixels * (double)((C1 + C2) / G);
    const size_t dyn = (size_t)(2 * (C1 + C2) + 2 * G) * sizeof(double);
    IM360_CHECK_ARG(dyn <= 64 * 1024, "groupnorm_finalize: C=%ld too wide for the LDS work area", (long)(C1 + C2));
    if (dtype == 0) hipLaunchKernelGGL((gn_finalize2_kernel<__bf16>), dim3((unsigned)N), dim3(256), dyn, (hipStream_t)stream, (const float*)pa, (int)Sa, (int)C1, (const float*)pb, (int)Sb, (int)C2,
                                       (const __bf16*)gamma, (const __bf16*)beta, (float*)scale, (float*)shift, (int)G, count, eps);
    else hipLaunchKernelGGL((gn_finalize2_kernel<_Float16>), dim3((unsigned)N), dim3(256), dyn, (hipStream_t)stream, (const float*)pa, (int)Sa, (int)C1, (const float*)pb, (int)Sb, (int)C2,
                            (const _Float16*)gamma, (const _Float16*)beta, (float*)scale, (float*)shift, (int)G, count, eps);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// GroupNorm statistics of the channel concatenation [xa | xb] without materialising it (xa [N,H,W,C1], xb [N,H,W,C2];
// gamma / beta / scale / shift span C1 + C2 channels; partial: N * S * 2 * (C1 + C2) floats).  The decoder's ResnetBlocks
// normalise torch.cat([x, skip]) (src/models/MVGenModel.py:407-437 -> animatediff/models/resnet.py:221-225).
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_stats_cat(const void* xa, const void* xb, const void* gamma, const void* beta, void* partial,
                                         void* scale, void* shift, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2,
                                         int64_t G, int64_t pad, float eps, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(xa && xb && gamma && beta && partial && scale && shift, "groupnorm_stats_cat: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C1 > 0 && C2 > 0 && G > 0, "groupnorm_stats_cat: empty problem");
    IM360_CHECK_ARG((C1 % 8) == 0 && (C2 % 8) == 0 && ((C1 + C2) % G) == 0, "groupnorm_stats_cat: C1=%ld, C2=%ld must be multiples of 8 and C1 + C2 of G=%ld", (long)C1, (long)C2, (long)G);
    IM360_CHECK_ARG(pad >= 0 && 2 * pad <= W, "groupnorm_stats_cat: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0, "groupnorm_stats_cat: misaligned input");
    IM360_CHECK_ARG(N <= 65535, "groupnorm_stats_cat: N=%ld exceeds grid.y", (long)N);
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_stats_cat: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_STATS, stream);
    if (dtype == 0) launch_gn_stats<__bf16>(xa, xb, C1, C2, gamma, beta, partial, scale, shift, N, H, W, G, pad, eps, (hipStream_t)stream);
    else launch_gn_stats<_Float16>(xa, xb, C1, C2, gamma, beta, partial, scale, shift, N, H, W, G, pad, eps, (hipStream_t)stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// y [N, H, W + 2 pad, C] = act(x * scale + shift) with circular W addressing; act: 0 none, 1 SiLU
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_apply(const void* x, const void* scale, const void* shift, void* y,
                                     int64_t N, int64_t H, int64_t W, int64_t C, int64_t pad, int act,
                                     int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && scale && shift && y, "groupnorm_apply: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "groupnorm_apply: bad shape");
    IM360_CHECK_ARG(pad >= 0 && pad <= W, "groupnorm_apply: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "groupnorm_apply: misaligned pointer");
    IM360_CHECK_ARG(N <= 65535, "groupnorm_apply: N=%ld exceeds grid.y", (long)N);
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_apply: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_APPLY, stream);
    if (dtype == 0) launch_gn_apply<__bf16>(x, nullptr, C, 0, scale, shift, y, N, H, W, pad, act, (hipStream_t)stream);
    else launch_gn_apply<_Float16>(x, nullptr, C, 0, scale, shift, y, N, H, W, pad, act, (hipStream_t)stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// y [N, H, W + 2 pad, C1 + C2] = act([xa | xb] * scale + shift): the normalised concatenation is written directly, the
// concatenation itself never exists (see im360_groupnorm_stats_cat)
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_apply_cat(const void* xa, const void* xb, const void* scale, const void* shift, void* y,
                                         int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t pad, int act,
                                         int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(xa && xb && scale && shift && y, "groupnorm_apply_cat: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C1 > 0 && C2 > 0 && (C1 % 8) == 0 && (C2 % 8) == 0, "groupnorm_apply_cat: bad shape");
    IM360_CHECK_ARG(pad >= 0 && pad <= W, "groupnorm_apply_cat: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0 && ((uintptr_t)y % 16) == 0, "groupnorm_apply_cat: misaligned pointer");
    IM360_CHECK_ARG(N <= 65535, "groupnorm_apply_cat: N=%ld exceeds grid.y", (long)N);
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_apply_cat: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_APPLY, stream);
    if (dtype == 0) launch_gn_apply<__bf16>(xa, xb, C1, C2, scale, shift, y, N, H, W, pad, act, (hipStream_t)stream);
    else launch_gn_apply<_Float16>(xa, xb, C1, C2, scale, shift, y, N, H, W, pad, act, (hipStream_t)stream);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// Per-slab partial sums with the circular-pad weighting of im360_groupnorm_stats (the `pad` wrapped columns count twice):
// partial fp32 [N][S][2][C], S = im360_gn_num_slabs(N, H, W).  pad = 0: im360_groupnorm_partial.
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_partial_pad(const void* x, void* partial, int64_t N, int64_t H, int64_t W, int64_t C, int64_t pad,
                                           int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && partial, "groupnorm_partial_pad: null pointer");
    IM360_CHECK_ARG(N > 0 && N <= 65535 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "groupnorm_partial_pad: bad shape");
    IM360_CHECK_ARG(pad >= 0 && 2 * pad <= W, "groupnorm_partial_pad: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0, "groupnorm_partial_pad: misaligned x");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_partial_pad: dtype %d unsupported", dtype);
    ProfScope prof(PROF_GN_STATS, stream);
    const int S = pick_slabs(N, H * W);
    if (dtype == 0) hipLaunchKernelGGL((gn_partial_kernel<__bf16>), dim3(S, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (float*)partial, (int)(H * W), (int)W, (int)C, (int)pad, S, (int)C, 0);
    else hipLaunchKernelGGL((gn_partial_kernel<_Float16>), dim3(S, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (float*)partial, (int)(H * W), (int)W, (int)C, (int)pad, S, (int)C, 0);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// y [N, H, W + 2 pad, C1 + C2] = act(GroupNorm([xa | xb])) straight from PARTIAL SUMS: channels [0, C1) of the statistics from pa
// ([N][Sa][2][C1]), [C1, C1 + C2) from pb ([N][Sb][2][C2]; xb / pb may be null with C2 = 0) -- written by im360_groupnorm_partial(_pad)
// or by the epilogue of the kernel that produced the tensor (im360_conv_fwd / im360_linear_fwd, gn_partial).  One launch: no
// finalize kernel, no scale / shift tensors (gn_apply_p_kernel).  Same bits as im360_groupnorm_finalize + im360_groupnorm_apply(_cat).
// Replaces: InflatedGroupNorm / nn.GroupNorm (+ F.silu, + pad_pano), animatediff/models/resnet.py:9-17, 221-243; attention.py:206, 262;
// motion_module.py:128, 169.
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_apply_partials(const void* xa, const void* xb, const void* pa, int64_t Sa, const void* pb, int64_t Sb,
                                              const void* gamma, const void* beta, void* y, int64_t N, int64_t H, int64_t W, int64_t C1,
                                              int64_t C2, int64_t G, int64_t pad, float eps, int act, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(xa && pa && gamma && beta && y, "groupnorm_apply_partials: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0 && G > 0 && Sa > 0 && (xb != nullptr) == (C2 > 0) && (pb != nullptr) == (C2 > 0) && (C2 == 0 || Sb > 0),
                    "groupnorm_apply_partials: bad problem");
    IM360_CHECK_ARG((C1 % 8) == 0 && (C2 % 8) == 0 && ((C1 + C2) % G) == 0, "groupnorm_apply_partials: C1=%ld, C2=%ld must be multiples of 8 and C1 + C2 of G=%ld", (long)C1, (long)C2, (long)G);
    IM360_CHECK_ARG(pad >= 0 && 2 * pad <= W, "groupnorm_apply_partials: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0 && ((uintptr_t)y % 16) == 0, "groupnorm_apply_partials: misaligned pointer");
    IM360_CHECK_ARG(N <= 65535, "groupnorm_apply_partials: N=%ld exceeds grid.y", (long)N);
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_apply_partials: dtype %d unsupported", dtype);
    const int C = (int)(C1 + C2);
    const size_t dyn = (size_t)(2 * C + 2 * G) * sizeof(double);
    IM360_CHECK_ARG(dyn <= 64 * 1024, "groupnorm_apply_partials: C=%d too wide for the LDS work area", C);
    ProfScope prof(PROF_GN_APPLY, stream);
    // slabs per image: every workgroup repeats the image's reduction in front of its slab, so fewer, larger slabs than the statistics
    // kernel's (knob gn_wgs = target number of workgroups; 0 = pick_slabs' 2048)
    int S = pick_slabs(N, H * (W + 2 * pad));
    if (const int tgt = knob(KNOB_GN_WGS); tgt > 0) {
        long s2 = (tgt + N - 1) / N;
        if (s2 < 1) s2 = 1;
        if (s2 < S) S = (int)s2;
    }
    // knob gn_apply: 0 = plain stores, 2 (default) = non-temporal stores (tools/ab_gn.py: - 2 ... - 22 % on the step's shapes, the
    // output is hundreds of MB that the next kernel streams once), 1 / 3 = the same with eight loads in flight (ties), 6 = non-temporal loads too
    const int v = knob(KNOB_GN_APPLY);
    hipStream_t s = (hipStream_t)stream;
#define IM360_GN_AP(T, NLD, NTS, NTL) hipLaunchKernelGGL((gn_apply_p_kernel<T, NLD, NTS, NTL>), dim3(S, (unsigned)N), dim3(256), dyn, s, (const T*)xa, (const T*)xb, (int)C1, (int)C2, \
        (const float*)pa, (int)Sa, (const float*)pb, (int)Sb, (const T*)gamma, (const T*)beta, (T*)y, (int)H, (int)W, (int)G, (int)pad, act, S, eps)
    if (dtype == 0) {
        switch (v & 7) {
            case 0: IM360_GN_AP(__bf16, 4, false, false); break;
            case 1: IM360_GN_AP(__bf16, 8, false, false); break;
            case 3: IM360_GN_AP(__bf16, 8, true, false); break;
            case 6: IM360_GN_AP(__bf16, 4, true, true); break;
            default: IM360_GN_AP(__bf16, 4, true, false); break;
        }
    } else {
        switch (v & 7) {
            case 0: IM360_GN_AP(_Float16, 4, false, false); break;
            case 1: IM360_GN_AP(_Float16, 8, false, false); break;
            case 3: IM360_GN_AP(_Float16, 8, true, false); break;
            case 6: IM360_GN_AP(_Float16, 4, true, true); break;
            default: IM360_GN_AP(_Float16, 4, true, false); break;
        }
    }
#undef IM360_GN_AP
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// GroupNorm (+ SiLU, + circular W pad, on x or on the never-materialised concatenation [xa | xb]) in ONE launch: statistics,
// their reduction and the normalisation pass of gn_fused_kernel.  partial: fp32 workspace of N * S * 2 * (C1 + C2) floats,
// S = im360_gn_num_slabs(N, H, W); counter: N int32 (zeroed here, on the stream).  xb may be null (C2 = 0).  Same bits as
// im360_groupnorm_stats + im360_groupnorm_apply.
// Replaces: InflatedGroupNorm / nn.GroupNorm (+ F.silu, + pad_pano) as the two functions above; one pass over HBM less.
extern "C" __attribute__((visibility("default"))) int im360_groupnorm_fused(const void* xa, const void* xb, const void* gamma, const void* beta, void* partial,
                                     void* counter, void* y, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2,
                                     int64_t G, int64_t pad, float eps, int act, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(xa && gamma && beta && partial && counter && y, "groupnorm_fused: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0 && G > 0 && (xb != nullptr) == (C2 > 0), "groupnorm_fused: bad problem");
    IM360_CHECK_ARG((C1 % 8) == 0 && (C2 % 8) == 0 && ((C1 + C2) % G) == 0, "groupnorm_fused: C1=%ld, C2=%ld must be multiples of 8 and C1 + C2 of G=%ld", (long)C1, (long)C2, (long)G);
    IM360_CHECK_ARG(pad >= 0 && 2 * pad <= W, "groupnorm_fused: pad %ld out of range", (long)pad);
    IM360_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0 && ((uintptr_t)y % 16) == 0, "groupnorm_fused: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "groupnorm_fused: dtype %d unsupported", dtype);
    const int S = pick_slabs(N, H * W);
    IM360_CHECK_ARG(N * S <= 0x7fffffffL, "groupnorm_fused: grid too large");
    const int C = (int)(C1 + C2);
    const size_t dyn = (size_t)2 * C * sizeof(float);
    IM360_CHECK_ARG(dyn <= 48 * 1024, "groupnorm_fused: C=%d too wide for the LDS scale / shift table", C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_GN_APPLY, stream);
    if (hipMemsetAsync(counter, 0, (size_t)N * sizeof(int), s) != hipSuccess) {
        im360_set_error("groupnorm_fused: hipMemsetAsync failed");
        return IM360_ERR_LAUNCH;
    }
    if (dtype == 0)
        hipLaunchKernelGGL((gn_fused_kernel<__bf16>), dim3((unsigned)(N * S)), dim3(256), dyn, s, (const __bf16*)xa, (const __bf16*)xb, (int)C1, (int)C2,
                           (float*)partial, (int*)counter, (const __bf16*)gamma, (const __bf16*)beta, (__bf16*)y, (int)H, (int)W, (int)G,
                           (int)pad, act, S, eps, C);
    else
        hipLaunchKernelGGL((gn_fused_kernel<_Float16>), dim3((unsigned)(N * S)), dim3(256), dyn, s, (const _Float16*)xa, (const _Float16*)xb, (int)C1, (int)C2,
                           (float*)partial, (int*)counter, (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)y, (int)H, (int)W, (int)G,
                           (int)pad, act, S, eps, C);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// x [rows, W, C] -> y [rows, W + 2 pad, C], circular along W (16-bit elements, C % 8 == 0)
extern "C" __attribute__((visibility("default"))) int im360_circular_pad_w(const void* x, void* y, int64_t rows, int64_t W, int64_t C, int64_t pad,
                                    int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && y, "circular_pad_w: null pointer");
    IM360_CHECK_ARG(rows > 0 && W > 0 && C > 0 && (C % 8) == 0 && pad >= 0, "circular_pad_w: bad shape");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "circular_pad_w: dtype %d unsupported", dtype);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "circular_pad_w: misaligned pointer");
    const long total = rows * (W + 2 * pad) * (C / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    // a byte copy of 16-bit elements: one instantiation serves both dtypes
    hipLaunchKernelGGL((circular_pad_w_kernel<uint16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, (long)rows, (int)W, (int)(C / 8), (int)pad);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// x [N, H, W] -> y [N, H + top + bottom, W + left + right], circular in both axes; elements of `esize` bytes (1, 2, 4, 8).
// Each pad must not exceed the size of its axis (torch's rule for mode="circular").
extern "C" __attribute__((visibility("default"))) int im360_circular_pad_hw(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t left, int64_t right,
                                     int64_t top, int64_t bottom, int64_t esize, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && y, "circular_pad_hw: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0, "circular_pad_hw: empty tensor");
    IM360_CHECK_ARG(esize == 1 || esize == 2 || esize == 4 || esize == 8, "circular_pad_hw: element size %ld unsupported", (long)esize);
    IM360_CHECK_ARG(left >= 0 && right >= 0 && top >= 0 && bottom >= 0 && left <= W && right <= W && top <= H && bottom <= H,
                    "circular_pad_hw: pads (%ld, %ld, %ld, %ld) must lie in [0, size of the axis]", (long)left, (long)right, (long)top, (long)bottom);
    IM360_CHECK_ARG((W + left + right) * esize <= 0x7fffffffL && H + top + bottom <= 0x7fffffffL, "circular_pad_hw: row too long");
    const long wb = W * esize, lb = left * esize, rb = right * esize;
    const uintptr_t align = (uintptr_t)x | (uintptr_t)y | (uintptr_t)wb | (uintptr_t)lb | (uintptr_t)rb;
    const int u = (align % 16) == 0 ? 16 : (align % 8) == 0 ? 8 : (align % 4) == 0 ? 4 : (align % 2) == 0 ? 2 : 1;
    const long units = N * (H + top + bottom) * ((wb + lb + rb) / u);
    const unsigned blocks = (unsigned)((units + 255) / 256 > 16384 ? 16384 : (units + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define IM360_PAD(U)                                                                                             \
    hipLaunchKernelGGL((circular_pad_hw_kernel<U>), dim3(blocks), dim3(256), 0, s, (const U*)x, (U*)y, (long)N,   \
                       (int)H, (int)(wb / u), (int)(lb / u), (int)(rb / u), (int)top, (int)bottom)
    if (u == 16) IM360_PAD(uint4);
    else if (u == 8) IM360_PAD(uint2);
    else if (u == 4) IM360_PAD(uint32_t);
    else if (u == 2) IM360_PAD(uint16_t);
    else IM360_PAD(uint8_t);
#undef IM360_PAD
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// out = cx * x + cv * (uncond + g (cond - uncond)), n elements (n % 8 == 0), all same dtype
extern "C" __attribute__((visibility("default"))) int im360_cfg_ddim_update(const void* uncond, const void* cond, const void* x, void* out, int64_t n,
                                     float guidance, float cx, float cv, int dtype, void* stream, const void* coef_dev) {
    using namespace im360;
    IM360_CHECK_ARG(uncond && cond && x && out, "cfg_ddim_update: null pointer");
    IM360_CHECK_ARG(n > 0 && (n % 8) == 0, "cfg_ddim_update: n=%ld must be a positive multiple of 8", (long)n);
    IM360_CHECK_ARG(((uintptr_t)uncond % 16) == 0 && ((uintptr_t)cond % 16) == 0 && ((uintptr_t)x % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "cfg_ddim_update: misaligned pointer");
    const long n8 = n / 8;
    const unsigned blocks = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((cfg_ddim_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)uncond,
                           (const __bf16*)cond, (const __bf16*)x, (__bf16*)out, n8, guidance, cx, cv, (const float*)coef_dev);
    else if (dtype == 1)
        hipLaunchKernelGGL((cfg_ddim_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, (const _Float16*)uncond,
                           (const _Float16*)cond, (const _Float16*)x, (_Float16*)out, n8, guidance, cx, cv, (const float*)coef_dev);
    else {
        im360_set_error("cfg_ddim_update: dtype %d unsupported", dtype);
        return IM360_ERR_UNSUPPORTED;
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}
