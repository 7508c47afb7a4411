// Implicit-GEMM 3x3 / 1x1 convolution on channels-last 16-bit activations for gfx950 (CDNA4 MFMA).
//
// Replaces the reference's per-frame nn.Conv2d call sites (InflatedConv3d is nn.Conv2d on
// '(b f) c h w', animatediff/models/resnet.py:19-27): ResnetBlock3D conv1/conv2/conv_shortcut
// (resnet.py:183-218), Downsample3D (stride 2, resnet.py:117-140), Upsample3D (nearest x2 folded into
// the input index, resnet.py:71-114), conv_in / conv_out (unet.py:134-137, 358) and the VAE convs.
// The equirectangular circular pad that the reference materialises around every pano conv
// (src/utils/pano.py:75-101, MVGenModel.py:138-143 ...) is folded into the addressing:
//   wrap  = 1 : the conv grid is circular along W (pad -> conv -> unpad with unpad >= 1)
//   x_off > 0 : output column j reads the (already W+4 wide) input at j + x_off (ResnetBlock conv2,
//               which must see the padded conv1 output because GroupNorm-2 statistics include it)
// Epilogue fuses bias, the time-embedding add (resnet.py:231-234) and the residual/shortcut add
// (resnet.py:248-251).
//
// The same kernel serves the large token-major nn.Linear layers as a 1x1 conv over a [M, 1, 1, K] view (bias and
// residual in the epilogue; a GEGLU epilogue variant: im360_linear_geglu).
//
// GEMM view: D^T[cout, pixel] = W[cout, (tap, cin)] * X^T[(tap, cin), pixel]; tiles of WM x WN waves with TM x TN
// v_mfma_f32_32x32x16 blocks each (see the template below), K-step = BK channels of one tap.  Both operands go HBM/L2 -> LDS by LDS-DMA
// (global_load_lds, 16 B per lane, no VGPR staging): the taps' shifted / wrapped / upsampled pixel
// addresses are per-lane SOURCE addresses, out-of-image taps read a 16-byte zero chunk, and the XOR
// swizzle that keeps ds_read_b128 fragment reads conflict-free is applied on the source side (the DMA
// destination is lane-linear).  Double-buffered LDS, one barrier per K-step: step s+1 streams in under
// the MFMAs of step s.
#include <string.h>

#include "common.h"

namespace im360 {

struct ConvParams {
    const void* x; const void* w; const void* bias; const void* temb; const void* res; void* y;
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int ntaps;          // 9 (3x3) or 1 (1x1)
    int stride, up, wrap, x_off, y_off;
    int imgs_per_temb;
    long M;             // N * Hout * Wout
    int tiles_n;        // ceil(Cout / 128)
    long nblocks;
    int halo_r, halo_seg, halo_pw, halo_p;      // halo kernel: output rows per tile, rows per image segment, patch width / pixels
    int up2_py, up2_px; // UP2 kernels: output parity (row, column) of this launch
    int dbg;            // ablation switches of the ring kernel (tools/ab_ring.py --ablate): 1 no LDS-DMA in the K loop, 2 no MFMA, 4 no fragment reads, 8 no epilogue
    // 1x1 convolution of a channel concatenation that is never materialised: input channels [0, Cin1) come from x
    // (pixel stride Cin1), channels [Cin1, Cin) from x2 (pixel stride Cin - Cin1); x2 == nullptr: one source
    const void* x2; int Cin1;
    // token-major linears only.  rs_out: the epilogue also writes, per output row and per 160-column wave slice, (sum,
    // sum of squares) of the stored 16-bit values: fp32 [M][Cout / 160][2] -- the LayerNorm statistics of the rows for a
    // consumer GEMM with the normalisation folded in (EPI 3 / 4), which reads them through rs_in ([M][rs_p][2]):
    //   y = rstd_r * (x W'^T - mu_r * c1) + (ln_tab ? ln_tab[(r / tab_div) % tab_mod] : c2)      W' = gamma (.) W; table rows INCLUDE c2
    float* rs_out; const float* rs_in; int rs_p; float ln_eps, ln_invc;
    const float* ln_c1; const float* ln_c2; const float* ln_tab; int tab_div, tab_mod;
    // persistent tile walk of the ring kernel: the cout tiles are split into `ngroups` groups, each walked by 8 / ngroups
    // XCDs over all pixel tiles (keeps one group's weights resident in those XCDs' L2s)
    int ngroups;
    // GroupNorm statistics from the epilogue (GNS kernels, 256 x 320 tiles): per pixel tile and output channel (sum, sum of
    // squares) of the values the tile stores, fp32 [M / 256][2][Cout] -- the layout of groupnorm.hip's per-slab partial sums
    // when an image is a whole number of tiles (H W % 256 == 0: slab s of image n = tile n * (H W / 256) + s), so the consumer's
    // GroupNorm skips its own statistics pass over the tensor (animatediff/models/resnet.py:221-243: norm -> act -> conv, twice)
    float* gn_out;
    // knob nt (default 1; tools/ab_step.py, profiles/r06_step_knobs_ab.log: - 1.5 ms per cfg2 step): the epilogue's output rows leave with
    // non-temporal stores -- tensors of hundreds of MB that should not push the operands out of the L2.  (The same on the attention
    // outputs, which the out-projection reads back at once, cost + 6.7 ms; on the LayerNorm / GEGLU element-wise kernels nothing.)
    int nt_store;
    // K-split of conv_igemm_kernel's 3 x 3 convolutions (taps innermost; round 6): `ksplit` workgroups share one output tile, each over a
    // contiguous range of the 64-channel chunks.  Parts 0 .. ksplit - 2 park their fp32 accumulators in ks_ws ([tile][part][160 * 512]
    // floats, lane-linear) and count themselves into ks_cnt[tile]; the LAST part -- dispatched behind all the others: block ids are part-major --
    // adds them in part order and runs the epilogue.  For launches whose tile count is not a whole number of rounds of 256 CUs (or below one).
    int ksplit;
    float* ks_ws;
    int* ks_cnt;
};

// ConvParams with the optional members cleared
static inline ConvParams conv_params_zero() {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.ngroups = 1;
    p.nt_store = knob(KNOB_NT) & 1;
    return p;
}


// 16 bytes of zeros that out-of-image taps are pointed at (LDS-DMA loads cannot zero-fill by themselves)
__device__ uint4 g_zero_chunk[1];

// keep a value alive without code (ablation switches of the ring kernel; a __device__ helper because the host pass of
// hipcc silently drops a __global__ body whose inline asm it cannot type for x86)
__device__ __forceinline__ void keep_alive(u32x4 v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void keep_alive(f32x16 v) { asm volatile("" ::"v"(v)); }

// ---- epilogue shared by the tile kernels.  `lds` is a region of at least (waves * 32 * row bytes) that no wave reads
//      as operand tiles any more; the caller has passed a workgroup barrier since the last operand read.
// UP2: the tile's rows are pixels of the LOW-resolution grid [N, Hout, Wout]; row (n, y, x) is stored at pixel
// (n, 2 y + up2_py, 2 x + up2_px) of the [N, 2 Hout, 2 Wout, Cout] output (sub-pixel form of nearest-x2 + conv3x3).
// cvec (EPI 3 / 4): the tile's fp32 column vectors staged in LDS by the caller, [0, BN) = c1, [BN, 2 BN) = c2, or the tile's table row (which includes c2).
// (row, 16-byte piece) of `lane` in store round `it` of a 32-row block whose rows hold PIECES pieces; false: the lane idles in this
// round (GNS only: every lane keeps ONE piece through all rounds, RPR = 64 / PIECES whole rows per round)
template <int PIECES, bool GNS>
__device__ __forceinline__ bool epi_round_rc(int it, int lane, int& row, int& pc) {
    if constexpr (GNS) {
        constexpr int RPR = 64 / PIECES;
        row = it * RPR + lane / PIECES;
        pc = lane % PIECES;
        const bool act = lane < RPR * PIECES && row < 32;
        row = row < 32 ? row : 31;
        return act;
    } else {
        const int f = it * 64 + lane;
        row = f / PIECES;
        pc = f % PIECES;
        return true;
    }
}
template <int PIECES, bool GNS>
constexpr int epi_rounds() { return GNS ? (32 + 64 / PIECES - 1) / (64 / PIECES) : (32 * PIECES) / 64; }
// EPI 3 (LayerNorm folded into the projection): mean / rstd of this lane's row in each of the wave's TM blocks from the producer's
// per-slice (sum, sum of squares).  All slices of both rows are requested before the first is added (2 / 4 / 8 slices unrolled: a
// run-time loop waits for each load in turn); the persistent kernel calls this right behind its K loop, BEFORE the next tile's
// first LDS-DMA requests: hipcc's wait for these loads is vmcnt(0) -- it does not see the asm requests -- and behind them it would
// also wait for a stage that was requested a moment ago.
template <int TM>
__device__ __forceinline__ void epi_ln_row_stats(const ConvParams& p, long m0, int wm, int col, float (&mus)[TM], float (&rstds)[TM]) {
    float ss[TM], qq[TM];
    const float* rows[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const long mr = m0 + wm * (TM * 32) + b * 32 + col;
        rows[b] = p.rs_in + (mr < p.M ? mr : p.M - 1) * p.rs_p * 2;
        ss[b] = qq[b] = 0.f;
    }
    auto sum_n = [&](auto nc) {
        constexpr int NS = decltype(nc)::value;
        f32x2 v[TM][NS];
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < NS; ++j) v[b][j] = *(const f32x2*)(rows[b] + j * 2);
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                ss[b] += v[b][j].x;
                qq[b] += v[b][j].y;
            }
    };
    if (p.rs_p == 2) sum_n(std::integral_constant<int, 2>{});
    else if (p.rs_p == 4) sum_n(std::integral_constant<int, 4>{});
    else if (p.rs_p == 8) sum_n(std::integral_constant<int, 8>{});
    else {
#pragma unroll
        for (int b = 0; b < TM; ++b)
            for (int j = 0; j < p.rs_p; ++j) {
                const f32x2 v = *(const f32x2*)(rows[b] + j * 2);
                ss[b] += v.x;
                qq[b] += v.y;
            }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        mus[b] = ss[b] * p.ln_invc;
        rstds[b] = __builtin_amdgcn_rsqf(fmaxf(qq[b] * p.ln_invc - mus[b] * mus[b], 0.f) + p.ln_eps);
    }
}

// RESM: 0 = residual known at run time only (its loads are unconditional: an absent one reads the zero chunk); 1 = residual
// present; 2 = no residual: the row-major pass moves whole 16-byte pieces LDS -> memory without unpacking them.  Round 4
// (tools/patches/ring_cycle_stamps.patch: cycle stamps of one workgroup): the plain epilogue of a 256 x 320 tile takes 14 000
// cycles and is VECTOR-bound (~ 6 instructions per element, more than half of them the unpack / add / repack of the row-major
// pass), 28 000 with a residual whose two loads per block each expose an HBM latency.  RESM 2 drops the row-major pass's
// arithmetic; requesting the residual pieces of BOTH blocks ahead (RESM 1's first form) put 90 - 110 registers into scratch next
// to the 160 accumulators; requesting block 0's from the kernel right behind its last MFMA (no scratch once the lane id was kept from
// being hoisted: a scratch reload is a VMEM load whose wait also waits for the pieces in front of it) measured 0.318 vs 0.306 ms on the
// level-0 out-projection + residual -- that GEMM moves 1.26 GB in 0.31 ms, it is at the HBM rate, not waiting for a latency.  Both
// dropped: RESM 1 requests per block, half before and half behind its register -> LDS pass.
// Same arithmetic in all three: identical values (RESM 2 keeps the sign of a zero that x + 0 would clear).
// ZACC (the four-wave tile, conv3x3_g4.hip): the accumulators live in AGPRs (asm constraint "a") and the next tile accumulates into
// them from its first MFMA; once a pixel block's values have left the registers its TN accumulators are cleared by one MFMA each with
// zero operands and C = 0 (the matrix pipe is idle during the epilogue; 256 v_accvgpr_write would cost 1000 issue cycles per tile).
template <typename T, bool VG = false> __device__ __forceinline__ void zero_acc_mfma(f32x16& d) {       // VG: this accumulator lives in VGPRs (the 256 x 320 tile's fifth cout block)
    const u32x4 z = {0u, 0u, 0u, 0u};
    if constexpr (VG) {
        if constexpr (std::is_same<T, __bf16>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0\n\ts_nop 11" : "=v"(d) : "v"(z));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0\n\ts_nop 11" : "=v"(d) : "v"(z));
        return;
    }
    // (s_nop 11: the wait states an 8-pass MFMA result needs before anything but an accumulating MFMA may touch it -- hipcc does not know
    //  what the statement is and may copy / spill the output right behind it; s_nop 1 in front: VALU write of z -> MFMA operand)
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0\n\ts_nop 11" : "=a"(d) : "v"(z));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0\n\ts_nop 11" : "=a"(d) : "v"(z));
}
template <typename T, int NT, int TM, int TN, int EPI, bool COUT8 = false, bool UP2 = false, bool GNS = false, int WN_ = 2, int RESM = 0, bool ZACC = false>
__device__ __forceinline__ void tile_epilogue(const ConvParams& p, f32x16 (&acc)[TN][TM], char* lds, long m0, int n0,
                                              int wm, int wn, int wid_s, int lane, const float* cvec = nullptr, int bn = 0,
                                              const float* ln_pre = nullptr) {
    const int col = lane & 31, hi = lane >> 5;
    const T* bias = (const T*)p.bias;
    const T* temb = (const T*)p.temb;
    const T* res = (const T*)p.res;
    T* yg = (T*)p.y;
    const int nw0 = n0 + wn * (TN * 32);          // first cout of this wave
    // LDS transpose of one 32-pixel block of this wave: rows of ROWB bytes (TN or TN/2 blocks of 32 couts), unpadded
    // and 16-byte aligned; the 16-byte piece index is XORed with row bits and the two 8-byte halves of a piece are
    // swapped on odd row octets, which makes the fragment-side 8-byte writes of 16 consecutive pixels hit 16
    // different bank pairs (checked exhaustively for 320- and 128-byte rows).  The row-major side reads whole pieces.
    // (256-byte rows -- the four-wave tile's plain epilogue -- take the 128-byte pattern: a row is then a whole number of the
    //  stores' 128-byte bank windows, and the XOR stays inside a group of eight pieces)
    // (64-byte rows -- the GEGLU epilogue of the 256 x 128 tile, TN = 2: four pieces per row, two rows per bank window)
    auto piece_xor = [](int row, int rowb) { return (rowb == 320 || rowb == 64) ? ((row >> 1) & 3) : (row & 7); };
    if constexpr (EPI == 1 || EPI == 4) {
        // GEGLU epilogue (token-major linear only): the packed weight rows alternate 32 value rows / 32 gate rows of the
        // same output channels, so accumulators (2i, 2i+1) hold value and gate of one channel in the same lane and
        // register: out = (value + b) * gelu(gate + b), half as many columns as the GEMM is wide.
        static_assert(TN % 2 == 0, "value / gate blocks come in pairs");
        constexpr int ROWB = (TN / 2) * 64;
        constexpr int PIECES = ROWB / 16;
        static_assert(ROWB == 128 || ROWB == 64, "swizzle pattern");
        const int I = p.Cout / 2;
        const int ow0 = nw0 / 2;                  // first output channel of this wave
        char* wlds = lds + wid_s * (32 * ROWB);
        const int fr = piece_xor(col, ROWB), br = (col >> 3) & 1;
        const T* gb = bias ? bias : (const T*)g_zero_chunk;      // unconditional bias loads (see the plain epilogue)
        const int gbmul = bias ? 1 : 0;
        // the biases of this lane's channels, unpacked once for all TM pixel blocks (the K-loop operand registers are free)
        f32x2 bv[EPI == 1 ? TN / 2 : 1][4][2], bt[EPI == 1 ? TN / 2 : 1][4][2];
        if constexpr (EPI == 1) {
#pragma unroll
        for (int i = 0; i < TN / 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rv = nw0 + (2 * i) * 32 + 8 * g + 4 * hi, rg = rv + 32;          // packed rows (bias index)
                const uint2 wv = *(const uint2*)(gb + rv * gbmul), wg = *(const uint2*)(gb + rg * gbmul);
                bv[i][g][0] = f32x2{unpack_lo<T>(wv.x), unpack_hi<T>(wv.x)};
                bv[i][g][1] = f32x2{unpack_lo<T>(wv.y), unpack_hi<T>(wv.y)};
                bt[i][g][0] = f32x2{unpack_lo<T>(wg.x), unpack_hi<T>(wg.x)};
                bt[i][g][1] = f32x2{unpack_lo<T>(wg.y), unpack_hi<T>(wg.y)};
            }
        }
        // EPI 4: LayerNorm folded into the projection -- the accumulators hold x W'^T of the RAW rows (W' = gamma (.) W);
        // with the row's mean / rstd from the producer's statistics, value = rstd * (acc - mu * c1) + c2 (fp32 vectors in
        // packed row order: c1 = row sums of W', c2 = W beta + bias).  The statistics of this lane's row in every block are
        // requested up front (one memory latency for the tile instead of one per block).
        float ln_mu[TM], ln_rs[TM];
        if constexpr (EPI == 4) {
            if (ln_pre) {                 // (the persistent kernel computed them behind its K loop, see epi_ln_row_stats)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    ln_mu[b] = ln_pre[b];
                    ln_rs[b] = ln_pre[TM + b];
                }
            } else {
                epi_ln_row_stats<TM>(p, m0, wm, col, ln_mu, ln_rs);
            }
        }
        static_for<TM>([&](auto bcc) {
            constexpr int b = decltype(bcc)::value;
            const long mb = m0 + wm * (TM * 32) + b * 32;
            f32x2 MU = {0.f, 0.f}, RS = {1.f, 1.f};
            if constexpr (EPI == 4) {
                MU = f32x2{ln_mu[b], ln_mu[b]};
                RS = f32x2{ln_rs[b], ln_rs[b]};
            }
            static_for<TN / 2>([&](auto icc) {
                constexpr int i = decltype(icc)::value;
                const f32x16 av = acc[2 * i][b], ag = acc[2 * i + 1][b];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // (the unfused path rounds the projection to 16 bits before the activation; this one does not --
                    //  one rounding less on the way to the fp32 oracle)
                    f32x2 v01 = {av[4 * g], av[4 * g + 1]}, v23 = {av[4 * g + 2], av[4 * g + 3]};
                    f32x2 t01 = {ag[4 * g], ag[4 * g + 1]}, t23 = {ag[4 * g + 2], ag[4 * g + 3]};
                    if constexpr (EPI == 4) {
                        const int rv = nw0 - n0 + (2 * i) * 32 + 8 * g + 4 * hi, rg = rv + 32;      // packed rows, relative to the tile
                        const f32x4 k1v = *(const f32x4*)(cvec + rv), k2v = *(const f32x4*)(cvec + bn + rv);
                        const f32x4 k1g = *(const f32x4*)(cvec + rg), k2g = *(const f32x4*)(cvec + bn + rg);
                        v01 = (v01 - MU * k1v.xy) * RS + k2v.xy;
                        v23 = (v23 - MU * k1v.zw) * RS + k2v.zw;
                        t01 = (t01 - MU * k1g.xy) * RS + k2g.xy;
                        t23 = (t23 - MU * k1g.zw) * RS + k2g.zw;
                    } else {
                    v01 += bv[i][g][0];
                    v23 += bv[i][g][1];
                    t01 += bt[i][g][0];
                    t23 += bt[i][g][1];
                    }
                    v01 *= gelu_poly_pk(t01);
                    v23 *= gelu_poly_pk(t23);
                    uint2 o;
                    o.x = pack2<T>(v01.x, v01.y);
                    o.y = pack2<T>(v23.x, v23.y);
                    *(uint2*)(wlds + col * ROWB + (((i * 4 + g) ^ fr) << 4) + ((hi ^ br) << 3)) = o;
                }
            });
            if constexpr (ZACC) static_for<TN>([&](auto ac) { zero_acc_mfma<T, (TN == 5 && decltype(ac)::value == 4)>(acc[decltype(ac)::value][b]); });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (32 * PIECES + 63) / 64; ++it) {
                const int f = it * 64 + lane;
                const int row = f / PIECES, pc = f % PIECES;
                const long mr = mb + row;
                const int co = ow0 + pc * 8;
                if (row < 32 && mr < p.M && co < I) {
                    uint4 o = *(const uint4*)(wlds + row * ROWB + ((pc ^ piece_xor(row, ROWB)) << 4));
                    if ((row >> 3) & 1) { const uint32_t t0 = o.x, t1 = o.y; o.x = o.z; o.y = o.w; o.z = t0; o.w = t1; }
                    if (p.nt_store) __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, (u32x4*)(yg + mr * I + co));
                    else *(uint4*)(yg + mr * I + co) = o;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        });
        return;
    } else {
    // ---- epilogue.  Lane (pixel = col, hi) holds 4 consecutive couts per register group, i.e. stored directly every
    //      lane would write 8 bytes at a pixel-row stride (32 rows x 16 B per instruction).  Instead each wave
    //      transposes its tile through LDS (free after the K loop), 32 pixels at a time: bias / temb are added in
    //      registers, the 16-bit rows are written to LDS, then read back row-major so consecutive lanes store (and
    //      fetch the residual from) consecutive 16-byte pieces of one output row -- whole 128-byte lines.
    if (COUT8 || (p.Cout & 7) == 0) {
        constexpr int ROWB = TN * 64;
        constexpr int PIECES = ROWB / 16;
        static_assert(ROWB == 128 || ROWB == 256 || ROWB == 320, "swizzle pattern");
        static_assert((32 * PIECES) % 64 == 0, "store rounds cover the block exactly");
        static_assert(!GNS || (!UP2 && EPI != 3), "GroupNorm statistics: plain conv / linear epilogues");
        // GNS: every lane keeps ONE 16-byte piece (8 channels) through all store rounds -- RPR whole rows per round on
        // RPR * PIECES lanes (60 of 64 at 320-byte rows: 11 rounds instead of 10) -- so that it can add up its channels'
        // (sum, sum of squares) over the rows it stores in 16 registers
        constexpr int RPR = 64 / PIECES;
        float gsum[GNS ? 8 : 1], gsq[GNS ? 8 : 1];
        if constexpr (GNS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gsum[e] = gsq[e] = 0.f;
        }
        char* wlds = lds + wid_s * (32 * ROWB);
        const int fr = piece_xor(col, ROWB), br = (col >> 3) & 1;
        // Every global load of the epilogue is UNCONDITIONAL: an absent operand (no bias / temb / residual) reads the
        // 16-byte zero chunk with a zero index multiplier, out-of-range rows / couts are clamped into the tensor (their
        // results are never stored).  A load behind a per-lane or per-operand branch makes hipcc wait vmcnt(0) at every
        // use -- and since stores count in vmcnt too, that serialised the ten store rounds of a block behind each other.
        // Addresses are a wave-uniform 64-bit base (block row 0) + a 32-bit per-lane byte offset: ten 64-bit load and ten
        // 64-bit store addresses per block would otherwise be kept live (and spilled) next to the accumulators.
        const char* zsrc = (const char*)g_zero_chunk;
        const char* bsrc = bias ? (const char*)bias : zsrc;
        const char* tsrc = temb ? (const char*)temb : zsrc;
        const uint32_t bmul = bias ? 2u : 0u, tmul = temb ? 2u : 0u, rmul = res ? 2u : 0u;      // bytes per element, or 0
        const int cmax4 = p.Cout - 4, cmax8 = p.Cout - 8;
        const long hw = (long)p.Hout * p.Wout;
        // EPI 3: LayerNorm folded into the projection (see ConvParams): mean / rstd of this lane's row in every block from the
        // producer's per-slice (sum, sum of squares), all requested up front (one memory latency per tile, not one per block)
        float ln_mus[EPI == 3 ? TM : 1], ln_rstds[EPI == 3 ? TM : 1];
        if constexpr (EPI == 3) {
            if (ln_pre) {                 // (the persistent kernel computed them behind its K loop: [mu_0 .. mu_TM-1, rstd_0 ..])
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    ln_mus[b] = ln_pre[b];
                    ln_rstds[b] = ln_pre[TM + b];
                }
            } else {
                epi_ln_row_stats<TM>(p, m0, wm, col, ln_mus, ln_rstds);
            }
        }
        constexpr int NIT = epi_rounds<PIECES, GNS>(), NIT1 = NIT / 2;
        u32x4 rvs[1][NIT];
        auto round_rc = [&](int it, int& row, int& pc) { return epi_round_rc<PIECES, GNS>(it, lane, row, pc); };
        auto piece_off_r = [&](int it, int rows, bool& ok) {      // element offset of this lane's piece in round `it` (clamped into the block)
            int row, pc;
            const bool act = round_rc(it, row, pc);
            const int co = nw0 + pc * 8;
            ok = act && row < rows && co < p.Cout;
            return (uint32_t)((row < rows ? row : rows - 1) * p.Cout + (co < cmax8 ? co : cmax8));
        };
        static_for<TM>([&](auto bcc) {
            constexpr int b = decltype(bcc)::value;
            const long mb = m0 + wm * (TM * 32) + b * 32;           // first pixel of the block (wave-uniform)
            // (ZACC = the four-wave tile: M % 256 == 0, every block is whole -- and no control flow splits the accumulators' live ranges)
            if constexpr (!ZACC) {
                if (mb >= p.M) return;
            }
            const int rows = ZACC ? 32 : (p.M - mb < 32 ? (int)(p.M - mb) : 32);  // valid rows of the block
            u32x4 (&rv)[NIT] = rvs[0];
            const char* rbase = res ? (const char*)(res + mb * p.Cout) : zsrc;
            char* ybase = (char*)(yg + mb * p.Cout);
            const long m = mb + (col < rows ? col : rows - 1);
            const uint32_t toff = (uint32_t)((m / hw) / p.imgs_per_temb) * (uint32_t)p.Cout;   // temb row of this lane's pixel
            float ln_mu = 0.f, ln_rstd = 1.f;
            if constexpr (EPI == 3) {
                ln_mu = ln_mus[b];
                ln_rstd = ln_rstds[b];
            }
            // residual pieces of this block: the first half is requested before the register -> LDS pass, the second
            // right after it (the accumulators it frees make room), so the HBM latency overlaps the shuffle work
            auto piece_off = [&](int it, bool& ok) { return piece_off_r(it, rows, ok); };
            auto load_res = [&](int it0, int it1) {
                if constexpr (RESM != 2) {
#pragma unroll
                    for (int it = it0; it < it1; ++it) {
                        bool ok;
                        rv[it] = *(const u32x4*)(rbase + piece_off(it, ok) * (RESM == 1 ? 2u : rmul));
                    }
                }
            };
            load_res(0, NIT1);
            __builtin_amdgcn_sched_barrier(0);    // (hipcc would hoist every load of the block up here and spill)
            static_for<TN>([&](auto acc_c) {
                constexpr int a = decltype(acc_c)::value;
                const f32x16 at = acc[a][b];
                if constexpr (EPI == 3) {
                    // the column vectors come from LDS (staged once per tile by the kernel: from global memory the forty
                    // dependent load rounds of a block cost more than the LayerNorm pass this epilogue replaces)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cr = nw0 - n0 + a * 32 + 8 * g + 4 * hi;
                        const f32x4 k1 = *(const f32x4*)(cvec + cr), k2 = *(const f32x4*)(cvec + bn + cr);
                        float f[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = (at[4 * g + j] - ln_mu * k1[j]) * ln_rstd + k2[j];
                        uint2 o;
                        o.x = pack2<T>(f[0], f[1]);
                        o.y = pack2<T>(f[2], f[3]);
                        *(uint2*)(wlds + col * ROWB + (((a * 4 + g) ^ fr) << 4) + ((hi ^ br) << 3)) = o;
                    }
                } else {
                uint2 wb[4], wt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = nw0 + a * 32 + 8 * g + 4 * hi;
                    const uint32_t cc = (uint32_t)(co < cmax4 ? co : cmax4);
                    if constexpr (EPI == 2 || EPI == 5) {
                        // (the persistent kernel staged the tile's bias slice in LDS; other callers pass no cvec)
                        // (COUT8 = the persistent kernel: it staged the tile's bias slice in LDS -- a compile-time fact, a run-time test of the
                        //  pointer costs a scalar branch per load, 40 per block)
                        if constexpr (COUT8) wb[g] = *(const uint2*)((const char*)cvec + (co - n0) * 2);
                        else wb[g] = *(const uint2*)(bsrc + cc * bmul);
                    } else {
                        wb[g] = *(const uint2*)(bsrc + cc * bmul);
                    }
                    if constexpr (EPI == 0) wt[g] = *(const uint2*)(tsrc + (toff + cc) * tmul);      // (token-major linears have no temb)
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float f[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = at[4 * g + j];
                    f[0] += unpack_lo<T>(wb[g].x); f[1] += unpack_hi<T>(wb[g].x); f[2] += unpack_lo<T>(wb[g].y); f[3] += unpack_hi<T>(wb[g].y);
                    if constexpr (EPI == 0) {
                        f[0] += unpack_lo<T>(wt[g].x); f[1] += unpack_hi<T>(wt[g].x); f[2] += unpack_lo<T>(wt[g].y); f[3] += unpack_hi<T>(wt[g].y);
                    }
                    uint2 o;
                    o.x = pack2<T>(f[0], f[1]);
                    o.y = pack2<T>(f[2], f[3]);
                    *(uint2*)(wlds + col * ROWB + (((a * 4 + g) ^ fr) << 4) + ((hi ^ br) << 3)) = o;
                }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (ZACC) static_for<TN>([&](auto ac) { zero_acc_mfma<T, (TN == 5 && decltype(ac)::value == 4)>(acc[decltype(ac)::value][b]); });
            load_res(NIT1, NIT);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int row, pc;
                const bool act = round_rc(it, row, pc);
                uint4 o = *(const uint4*)(wlds + row * ROWB + ((pc ^ piece_xor(row, ROWB)) << 4));
                if ((row >> 3) & 1) { const uint32_t t0 = o.x, t1 = o.y; o.x = o.z; o.y = o.w; o.z = t0; o.w = t1; }
                if constexpr (RESM != 2) {
                const u32x4 w = rv[it];
                float fv[8] = {unpack_lo<T>(o.x) + unpack_lo<T>(w.x), unpack_hi<T>(o.x) + unpack_hi<T>(w.x), unpack_lo<T>(o.y) + unpack_lo<T>(w.y), unpack_hi<T>(o.y) + unpack_hi<T>(w.y),
                               unpack_lo<T>(o.z) + unpack_lo<T>(w.z), unpack_hi<T>(o.z) + unpack_hi<T>(w.z), unpack_lo<T>(o.w) + unpack_lo<T>(w.w), unpack_hi<T>(o.w) + unpack_hi<T>(w.w)};
                o.x = pack2<T>(fv[0], fv[1]);
                o.y = pack2<T>(fv[2], fv[3]);
                o.z = pack2<T>(fv[4], fv[5]);
                o.w = pack2<T>(fv[6], fv[7]);
                }
                bool ok;
                const uint32_t off = piece_off(it, ok);
                if constexpr (GNS) {
                    // statistics of the values as STORED (rounded to 16 bits: what a statistics pass over the tensor would read)
                    const float wgt = ok ? 1.f : 0.f;
                    const float sv[8] = {unpack_lo<T>(o.x), unpack_hi<T>(o.x), unpack_lo<T>(o.y), unpack_hi<T>(o.y), unpack_lo<T>(o.z), unpack_hi<T>(o.z), unpack_lo<T>(o.w), unpack_hi<T>(o.w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = wgt * sv[e];
                        gsum[e] += v;
                        gsq[e] = fmaf(v, sv[e], gsq[e]);
                    }
                }
                if (EPI == 5 && (!GNS || act)) {
                    // row statistics of the STORED values (what a LayerNorm of this tensor would read): this piece's
                    // (sum, sum of squares) parked in its own staging slot, collected per row after the store rounds
                    float ss = 0.f, qq = 0.f;
                    ss = dot2_acc<T>(o.x, Elem<T>::ones2, ss); qq = dot2_acc<T>(o.x, o.x, qq);
                    ss = dot2_acc<T>(o.y, Elem<T>::ones2, ss); qq = dot2_acc<T>(o.y, o.y, qq);
                    ss = dot2_acc<T>(o.z, Elem<T>::ones2, ss); qq = dot2_acc<T>(o.z, o.z, qq);
                    ss = dot2_acc<T>(o.w, Elem<T>::ones2, ss); qq = dot2_acc<T>(o.w, o.w, qq);
                    *(f32x2*)(wlds + row * ROWB + ((pc ^ piece_xor(row, ROWB)) << 4)) = f32x2{ss, qq};
                }
                if constexpr (UP2) {
                    const int co = nw0 + pc * 8;
                    const long mrow = mb + (row < rows ? row : rows - 1);
                    const long nimg = mrow / hw;
                    const int rem = (int)(mrow - nimg * hw);
                    const int yy = rem / p.Wout, xx = rem - yy * p.Wout;
                    const long orow = (nimg * (2L * p.Hout) + 2 * yy + p.up2_py) * (2L * p.Wout) + 2 * xx + p.up2_px;
                    if (ok) *(uint4*)((char*)yg + (orow * p.Cout + (co < cmax8 ? co : cmax8)) * 2) = o;
                } else if (ok) {
                    if (p.nt_store) __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, (u32x4*)(ybase + off * 2u));
                    else *(uint4*)(ybase + off * 2u) = o;
                }
            }
            if constexpr (EPI == 5) {
                // two lanes per row add up the row's pieces in a fixed order (deterministic, unlike atomics) and write the
                // wave slice's (sum, sum of squares) of the row: rs_out[row][Cout / (32 TN)][2]
                static_assert(PIECES % 2 == 0, "two lanes per row");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int row = lane >> 1, half = lane & 1;
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int j = 0; j < PIECES / 2; ++j) {
                    const int pc = half * (PIECES / 2) + j;
                    const f32x2 v = *(const f32x2*)(wlds + row * ROWB + ((pc ^ piece_xor(row, ROWB)) << 4));
                    ss += v.x;
                    qq += v.y;
                }
                ss += __shfl_xor(ss, 1);
                qq += __shfl_xor(qq, 1);
                const int slices = p.Cout / (TN * 32);
                if (half == 0 && row < rows)
                    *(f32x2*)(p.rs_out + ((mb + row) * slices + nw0 / (TN * 32)) * 2) = f32x2{ss, qq};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();      // the next block overwrites the staging rows
        });
        if constexpr (GNS) {
            // this wave's staging rows are free: park the lane's 16 sums there ([lane][16] floats), then -- after a workgroup
            // barrier -- thread c of the tile adds up channel c over the waves that stored its rows (same wn, every wm) and
            // the RPR row-lanes of each, in a fixed order (deterministic), and writes the tile's (sum, sum of squares)
            float* st = (float*)wlds;
            if (lane < RPR * PIECES) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    st[lane * 16 + e] = gsum[e];
                    st[lane * 16 + 8 + e] = gsq[e];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            constexpr int WM_ = NT / 64 / WN_, BNT = WN_ * TN * 32;
            const int c = wid_s * 64 + lane;
            if (c < BNT && n0 + c < p.Cout) {
                const int wn_c = c / (TN * 32), chunk = (c % (TN * 32)) / 8, e = c % 8;
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int m = 0; m < WM_; ++m)
#pragma unroll
                    for (int r = 0; r < RPR; ++r) {
                        const float* src = (const float*)(lds + (m * WN_ + wn_c) * (32 * ROWB)) + (r * PIECES + chunk) * 16;
                        ss += src[e];
                        qq += src[8 + e];
                    }
                constexpr int BM_ = WM_ * TM * 32;
                float* dst = p.gn_out + (m0 / BM_) * 2 * (long)p.Cout + n0 + c;
                dst[0] = ss;
                dst[p.Cout] = qq;
            }
        }
        return;
    }
    // Cout not a multiple of 8 (conv_out's 4 channels, odd test shapes): stores straight from the fragments
    if constexpr (!COUT8) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const long m = m0 + wm * (TM * 32) + b * 32 + col;
        if (m >= p.M) continue;
        const long img = m / ((long)p.Hout * p.Wout);
        const T* trow = temb ? temb + (img / p.imgs_per_temb) * p.Cout : nullptr;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = nw0 + a * 32 + 8 * g + 4 * hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (co + j < p.Cout) {
                        float vv = acc[a][b][4 * g + j];
                        if (bias) vv += to_f32(bias[co + j]);
                        if (trow) vv += to_f32(trow[co + j]);
                        if (res) vv += to_f32(res[m * p.Cout + co + j]);
                        yg[m * p.Cout + co + j] = from_f32<T>(vv);
                    }
                }
            }
        }
    }
    }
    }
}

// WM x WN waves, each owning TM x TN blocks of 32 pixels x 32 couts:
//   (2, 2, 2, 2) = 128 pixels x 128 couts, 4 waves     -- small problems, odd Cout
//   (4, 1, 2, 2) = 256 pixels x 64 couts, 4 waves      -- Cout that leaves a half-empty last 128-tile
//   (4, 2, 2, 5) = 256 pixels x 320 couts, 8 waves     -- the UNet's Cout = 320 / 640 / 1280 at large pixel counts.
// EPI: 0 = conv epilogue (bias, temb, residual); 1 = GEGLU; 2 = same as 0, instantiated under its own symbol for the
// token-major Linear use so that profiles tell the two apart.
// The 128-wide tiles move 1 byte of operands into LDS per 64 flops and saturate the CU's global->LDS path at ~30 % of
// the MFMA peak (same throughput at 2 or 3 resident workgroups); the 256 x 320 tile halves the bytes per flop
// (142 flop/B) and runs as one 8-wave workgroup per CU (144 KB of LDS for the two stages).
// CM ("chunk-major", 3x3 without wrap / upsample only): K runs over (64-channel chunk, tap) instead of (tap, chunk).  The nine
// shifted reads of one chunk then follow each other, so eight of them hit the XCD's L2 (the tap-major order re-streams the
// pixel tile from the fabric for every tap: fetch / input = 8.9 - 14, profiles/r02_hbm_traffic.json).  The fp32 summation
// order differs from the tap-major kernels.
// ABL: ablation build (knob conv_dbg, tools/ab_ring.py --ablate): bit 1 of p.dbg skips the LDS-DMA of the K loop, bit 2 the MFMAs.
// UP2 (4 taps): one output parity of nearest-x2-upsample + conv3x3 as a 2 x 2 convolution of the low-resolution input with
// pre-summed weights (after the upsample every output parity sees only 2 x 2 distinct source pixels): tap (r, c) reads
// source pixel (y + r + py - 1, x + c + px - 1); 4 / 9 of the MACs of the upsampled form.
template <typename T, int BK, int WM, int WN, int TM, int TN, int EPI = 0, bool CM = false, bool ABL = false, bool UP2 = false, bool GNS = false, bool STAG = false, int RESM = 0>
// (waves per SIMD given as min AND max: with the minimum alone hipcc aimed the 256 x 64 tile at three waves per SIMD -- 168 registers --
//  and spilled 688 bytes inside the K loop once ConvParams grew in round 3: 1.58 ms instead of 0.25 ms for a cfg1-sized 320 -> 320
//  convolution; its LDS footprint allows two workgroups per CU anyway)
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(TM * TN * 16 > 200 ? 1 : 2, TM * TN * 16 > 200 ? 1 : 2))) void conv_igemm_kernel(ConvParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
    constexpr int ROWB = BK * 2;                // bytes per tile row
    constexpr int RPB = 256 / ROWB;             // tile rows per 256-byte LDS bank row
    static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "staging pattern");
    constexpr int LDA = (BM * CPR) / NT;       // LDS-DMA loads per thread per K-step: activations
    constexpr int LDB = (BN * CPR) / NT;       //                                         weights
    constexpr int KC = BK / 16;
    constexpr int TILE_A = BM * ROWB, TILE_B = BN * ROWB;     // bytes of the operand tiles
    constexpr int STAGE = TILE_A + TILE_B;
    // [buffer][A | B]: unpadded row-major tiles written by global_load_lds (lane-linear destination), the
    // 16-byte chunk position XOR-swizzled with the row so MFMA fragment reads (16 rows, one K chunk) hit 16
    // different slots of the 256-byte bank row
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    // XCD-aware tile order: consecutive logical tiles (same pixel tile, neighbouring cout tiles) share an L2
    long bid = blockIdx.x;
    // K-split (ConvParams::ksplit): block ids are PART-major, so every tile's owner (its last part) is dispatched behind the parts it waits for
    // (compiled into the 256 x 320 tile's 3 x 3 kernels: chunk-major and -- the panorama's wrap-addressed convolutions -- tap-major K order)
    constexpr bool KSOK = WM == 4 && WN == 2 && TM == 2 && TN == 5 && EPI == 0 && !UP2 && !ABL && !STAG;
    const int ks_S = (KSOK && p.ksplit > 1) ? p.ksplit : 1;
    int ks_part = 0;
    if (KSOK && ks_S > 1) {
        ks_part = (int)(bid / p.nblocks);
        bid -= (long)ks_part * p.nblocks;
    }
    {
        const long nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long tile_m = bid / p.tiles_n;
    const int tile_n = (int)(bid % p.tiles_n);
    const long m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int Hc = p.up ? 2 * p.Hin : p.Hin, Wc = p.up ? 2 * p.Win : p.Win;
    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w;
    const T* zero = (const T*)g_zero_chunk;

    // per-thread staging slots: chunk c = i * NT + tid -> tile row c / CPR = tid / CPR + i * (NT / CPR), LDS position
    // c % CPR = tid % CPR.  NT / CPR is a multiple of 32, so the swizzled data chunk is the same for every i; the pixel
    // coordinates are kept packed (y | x << 16, n | valid << 31) to leave registers for the MFMA operands.
    constexpr int RPI = NT / CPR;               // tile rows between a thread's consecutive chunks
    static_assert(RPI % 32 == 0, "staging rows");
    const int srow = tid / CPR;
    const int pd8 = ((tid % CPR) ^ ((srow / RPB) & (CPR - 1))) * 8;        // data chunk (elements) stored at this position
    uint32_t pyx[LDA], pnv[LDA];
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
        const long m = m0 + srow + i * RPI;
        const bool valid = m < p.M;
        const long mm = valid ? m : 0;
        const long t = mm / p.Wout;
        pyx[i] = (uint32_t)(t % p.Hout) | ((uint32_t)(mm % p.Wout) << 16);
        pnv[i] = (uint32_t)(t / p.Hout) | (valid ? 0x80000000u : 0u);
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ksteps_per_tap = p.Cin / BK;
    int nsteps = p.ntaps * ksteps_per_tap;
    int ks_c0 = 0;                             // chunk-major K order (nine taps per chunk): first 64-channel chunk of this part
    int ks_s0 = 0;                             // tap-major K order (Cin / 64 steps per tap): first step of this part
    if (KSOK && ks_S > 1) {
        if constexpr (CM) {
            ks_c0 = (int)((long)ks_part * ksteps_per_tap / ks_S);
            const int c1 = (int)((long)(ks_part + 1) * ksteps_per_tap / ks_S);
            nsteps = (c1 - ks_c0) * p.ntaps;
        } else {
            ks_s0 = (int)((long)ks_part * nsteps / ks_S);
            nsteps = (int)((long)(ks_part + 1) * nsteps / ks_S) - ks_s0;
        }
    }

    // Producer state of the LDS-DMA stream.  Per K-step only pointer bumps remain: the tap geometry (shift,
    // wrap, upsample, bounds -> source pixel or the zero chunk) is evaluated once per tap, i.e. every Cin / BK
    // steps; the packed weights [cout][tap][cin] are contiguous across taps, so their pointers just keep advancing.
    const T* aptr[LDA];
    uint32_t amask = 0;                         // bit i: slot i reads real pixels (advances with K), else the zero chunk
    // weights: the thread's LDB rows are RPI rows apart -> one pointer + a uniform stride
    const T* bptr = wg + (long)(n0 + srow) * p.ntaps * p.Cin + pd8;
    const long bstride = (long)RPI * p.ntaps * p.Cin;
    int tap_p = 0, kk_p = 0;
    // two sources (p.x2, 1x1 only): `second` = the channels past Cin1, read from x2 with its own pixel stride
    const int ksteps_src1 = p.x2 ? p.Cin1 / BK : -1;
    auto set_tap = [&](int tap, bool second = false) {
        const int dy = UP2 ? tap / 2 + p.up2_py : (p.ntaps == 9 ? tap / 3 : 1), dx = UP2 ? tap % 2 + p.up2_px : (p.ntaps == 9 ? tap % 3 : 1);
        const T* src = second ? (const T*)p.x2 : xg;
        const int cs = p.x2 ? (second ? p.Cin - p.Cin1 : p.Cin1) : p.Cin;       // channels per pixel of the source tensor
#pragma unroll
        for (int i = 0; i < LDA; ++i) {
            int gy = (int)(pyx[i] & 0xffffu) * p.stride + dy - 1 + p.y_off;
            int gx = (int)(pyx[i] >> 16) * p.stride + dx - 1 + p.x_off;
            bool ok = (pnv[i] >> 31) && gy >= 0 && gy < Hc;
            if (p.wrap) {
                gx = gx < 0 ? gx + Wc : (gx >= Wc ? gx - Wc : gx);      // |shift| <= 2 < Wc: one conditional wrap
            } else {
                ok = ok && gx >= 0 && gx < Wc;
            }
            const int sy = gy >> p.up, sx = gx >> p.up;
            const long off = ok ? (((long)(pnv[i] & 0x7fffffffu) * p.Hin + sy) * p.Win + sx) * cs + pd8 : 0;
            aptr[i] = ok ? src + off : zero;
            amask = ok ? (amask | (1u << i)) : (amask & ~(1u << i));
        }
    };
    // chunk-major state: the centre pixel's pointer per slot, nine validity bits per slot, and wave-uniform element offsets
    // of the current tap (relative to the centre) and of the current channel chunk
    uint32_t vmask[LDA];
    long tapdelta = 0, kofs = 0;
    if constexpr (CM) {
#pragma unroll
        for (int i = 0; i < LDA; ++i) {
            const int cy = (int)(pyx[i] & 0xffffu) * p.stride + p.y_off, cx = (int)(pyx[i] >> 16) * p.stride + p.x_off;
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int gy = cy + t / 3 - 1, gx = cx + t % 3 - 1;
                m |= (gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) ? (1u << t) : 0u;
            }
            const bool valid = (pnv[i] >> 31) != 0;
            vmask[i] = valid ? m : 0u;
            aptr[i] = valid ? xg + (((long)(pnv[i] & 0x7fffffffu) * p.Hin + cy) * p.Win + cx) * p.Cin + pd8 : xg;
        }
        tapdelta = -(long)(p.Win + 1) * p.Cin;          // tap 0 = (dy, dx) = (-1, -1)
        kofs = (long)ks_c0 * BK;
    } else {
        // (a K part of the tap-major order starts inside tap ks_s0 / steps-per-tap: that tap's pixel pointers, advanced by the channel chunks
        //  in front of it; the packed weights [cout][tap][cin] are contiguous across taps: one offset)
        tap_p = ks_s0 / ksteps_per_tap;
        kk_p = ks_s0 % ksteps_per_tap;
        set_tap(tap_p);
        if (ks_s0) {
#pragma unroll
            for (int i = 0; i < LDA; ++i) aptr[i] += ((amask >> i) & 1u) ? kk_p * BK : 0;
            bptr += (long)ks_s0 * BK;
        }
    }
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);          // wave-uniform: LDS-DMA destinations stay in SGPRs

    const uint32_t lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
    // one LDS-DMA piece (1 KiB per wave) of the next K step: pieces 0 .. LDA-1 are the pixel rows, LDA .. LDA+LDB-1 the weights;
    // the producer state only moves in stage_advance(), so the pieces of a step can be issued anywhere inside the step
    auto stage_piece = [&](int buf, auto jc) {
        constexpr int j = decltype(jc)::value;
        char* abase = lds + buf * STAGE + wid_s * 1024;
        char* bbase = abase + TILE_A;
        if constexpr (j < LDA) {
            const T* src;
            if constexpr (CM) src = ((vmask[j] >> tap_p) & 1u) ? aptr[j] + (tapdelta + kofs) : zero;
            else src = aptr[j];
            if constexpr (STAG) lds_dma16_asm(src, lds_u32 + (uint32_t)(buf * STAGE + wid_s * 1024 + j * (NT * 16)));
            else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(abase + j * (NT * 16)), 16, 0, 0);
        } else {
            constexpr int i = j - LDA;
            const T* src = bptr + i * bstride;
            if constexpr (CM) src += (long)tap_p * p.Cin + kofs;
            if constexpr (STAG) lds_dma16_asm(src, lds_u32 + (uint32_t)(buf * STAGE + wid_s * 1024 + TILE_A + i * (NT * 16)));
            else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(bbase + i * (NT * 16)), 16, 0, 0);
        }
    };
    auto stage_advance = [&]() {
        if constexpr (CM) {
            ++tap_p;
            tapdelta += p.Cin;
            if (tap_p == 3 || tap_p == 6) tapdelta += (long)(p.Win - 3) * p.Cin;
            if (tap_p == 9) {
                tap_p = 0;
                tapdelta = -(long)(p.Win + 1) * p.Cin;
                kofs += BK;
            }
        } else {
#pragma unroll
            for (int i = 0; i < LDA; ++i) aptr[i] += ((amask >> i) & 1u) ? BK : 0;
            bptr += BK;
            if (++kk_p == ksteps_per_tap) {
                kk_p = 0;
                if (++tap_p < p.ntaps) set_tap(tap_p);
            } else if (kk_p == ksteps_src1) {
                set_tap(tap_p, true);           // the concatenation's second tensor takes over (weights just keep advancing)
            }
        }
    };
    auto stage = [&](int buf) {
        static_for<LDA + LDB>([&](auto jc) { stage_piece(buf, jc); });
        stage_advance();
    };
    // One wave per SIMD (the 4-wave 192 x 320 tile): nobody else fills the matrix pipe while this wave issues its 16 pieces
    // back to back, so they are spread over the step's MFMAs instead, PPC pieces per 16-channel chunk.
    constexpr bool ILV = NT == 256 && TM * TN * 16 > 200;
    constexpr int PPC = (LDA + LDB + KC - 1) / KC;

    // fragment byte offsets inside a tile: row R, K chunk d -> R * ROWB + ((d ^ swz(R)) * 16).  Every row a lane reads
    // is its base row + a multiple of 32, and swz only looks at row bits below 32, so the swizzled chunk offset is one
    // register per K chunk (shared by both operands) and the 32-row blocks are immediate offsets.
    const int swz = (col / RPB) & (CPR - 1);
    int koff[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) koff[kc] = ((kc * 2 + hi) ^ swz) * 16;
    const int wrow = (wn * (TN * 32) + col) * ROWB, xrow = (wm * (TM * 32) + col) * ROWB;

    stage(0);
    if constexpr (STAG) {
        // The staggered loop of conv_ring_kernel's MODE 3 (see there for the interval arithmetic) on this kernel's producer: each
        // 64-channel step as two (R, C) interval pairs -- R fetches the 14 fragments of two 16-channel chunks, C runs their 20 MFMAs
        // from registers --, the wave groups 0-3 / 4-7 one barrier interval apart so that every SIMD has one wave in C while its
        // other wave reads LDS; step s + 1 is requested during the two intervals after the last read of step s - 1 and waited for
        // (vmcnt(0): nothing younger in flight) in front of the barrier that ends step s.
        static_assert(!STAG || (NT == 512 && BK == 64), "staggered loop: the 8-wave tile on 64-channel steps");
        constexpr int NP = LDA + LDB, PA = (NP + 1) / 2;
        const int grp = wid_s >> 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (grp) {
            if (nsteps > 1) static_for<PA>([&](auto ic) { stage_piece(1, ic); });
            asm volatile("s_barrier" ::: "memory");
        }
        const char* bt0 = lds + TILE_A;
        for (int s = 0; s < nsteps; ++s) {
            const char* at = lds + (s & 1) * STAGE;
            const char* bt = bt0 + (s & 1) * STAGE;
            const bool req1 = s + 1 < nsteps, req2 = s + 2 < nsteps;
            const int rs1 = (s + 1) & 1, rs2 = s & 1;
            u32x4 xf[2][TM], wf[2][TN];
            if (req1) {
                if (grp) {
                    static_for<NP - PA>([&](auto ic) { stage_piece(rs1, std::integral_constant<int, PA + decltype(ic)::value>{}); });
                    stage_advance();
                } else {
                    static_for<PA>([&](auto ic) { stage_piece(rs1, ic); });
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(at + xrow + koff[kc] + b * (32 * ROWB));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(bt + wrow + koff[kc] + a * (32 * ROWB));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            static_for<2 * TN * TM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                if constexpr (m % 4 == 3 && m / 4 < NP - PA) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (req1 && !grp) stage_piece(rs1, std::integral_constant<int, PA + m / 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (req1 && !grp) stage_advance();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(at + xrow + koff[2 + kc] + b * (32 * ROWB));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(bt + wrow + koff[2 + kc] + a * (32 * ROWB));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            static_for<2 * TN * TM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                if constexpr (m % 4 == 3 && m / 4 < PA) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (req2 && grp) stage_piece(rs2, std::integral_constant<int, m / 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        if (!grp) asm volatile("s_barrier" ::: "memory");
    } else
    for (int s = 0; s < nsteps; ++s) {
        __syncthreads();              // step s landed (vmcnt(0) precedes the barrier); buffer (s+1)&1 is free again
        const bool more = s + 1 < nsteps && !(ABL && (p.dbg & 1));
        if (!ILV && more) stage((s + 1) & 1);
        const char* at = lds + (s & 1) * STAGE;
        const char* bt = at + TILE_A;
        // software-pipelined fragment reads: the pixel fragments and the first TN - LATE weight fragments of K chunk
        // kc+1 are fetched while the MFMAs of chunk kc run (two register sets); the last LATE weight fragments of a
        // chunk are fetched at its start and consumed by its last MFMAs -- the most prefetch 256 registers allow
        constexpr int LATE = TN >= 4 ? 2 : 0;
        u32x4 wf[2][TN - LATE], xf[2][TM], wl[LATE ? LATE : 1];
        auto fetch = [&](auto kcc) {
            constexpr int kc = decltype(kcc)::value;
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[kc & 1][b] = *(const u32x4*)(at + xrow + koff[kc] + b * (32 * ROWB));
#pragma unroll
            for (int a = 0; a < TN - LATE; ++a) wf[kc & 1][a] = *(const u32x4*)(bt + wrow + koff[kc] + a * (32 * ROWB));
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<KC>([&](auto kcc) {
            constexpr int kc = decltype(kcc)::value;
#pragma unroll
            for (int a = 0; a < LATE; ++a) wl[a] = *(const u32x4*)(bt + wrow + koff[kc] + (TN - LATE + a) * (32 * ROWB));
            if constexpr (kc + 1 < KC) fetch(std::integral_constant<int, kc + 1>{});
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this chunk's MFMAs (hipcc sinks it otherwise)
            if (ABL && (p.dbg & 2)) {
#pragma unroll
                for (int a = 0; a < TN - LATE; ++a) keep_alive(wf[kc & 1][a]);
#pragma unroll
                for (int b = 0; b < TM; ++b) keep_alive(xf[kc & 1][b]);
#pragma unroll
                for (int a = 0; a < LATE; ++a) keep_alive(wl[a]);
            } else if constexpr (ILV) {
                // MFMAs of this chunk with the chunk's share of the next step's pieces pinned between them
                constexpr int NM = TN * TM, GAP = NM / PPC;
                static_for<NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, a = m / TM, b = m % TM;
                    acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, a < TN - LATE ? wf[kc & 1][a < TN - LATE ? a : 0] : wl[a >= TN - LATE ? a - (TN - LATE) : 0]),
                                                __builtin_bit_cast(uint4, xf[kc & 1][b]), acc[a][b]);
                    if constexpr ((m + 1) % GAP == 0 && (m + 1) / GAP <= PPC) {
                        constexpr int j = kc * PPC + (m + 1) / GAP - 1;
                        if constexpr (j < LDA + LDB) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) stage_piece((s + 1) & 1, std::integral_constant<int, j>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
            } else
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, a < TN - LATE ? wf[kc & 1][a < TN - LATE ? a : 0] : wl[a >= TN - LATE ? a - (TN - LATE) : 0]),
                                                __builtin_bit_cast(uint4, xf[kc & 1][b]), acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (ILV && more) stage_advance();
    }

    static_assert((NT / 64) * 32 * ((EPI == 1 || EPI == 4) ? (TN / 2) * 64 : TN * 64) <= 2 * STAGE, "epilogue staging exceeds the K-loop LDS");
    __syncthreads();                              // every wave is done reading the operand tiles
    if constexpr (KSOK) {
        if (ks_S > 1) {
            // partial sums of one part: [a][b][r] registers x NT lanes, lane-linear (every store / load instruction one contiguous 2 KB).
            // Agent-scope relaxed atomics = write-through stores and L2-bypassing loads (the parts of a tile may sit on different XCDs);
            // "all of this workgroup's stores are acknowledged" (vmcnt(0) + barrier) orders them in front of the counter increment.
            constexpr int NACC = TN * TM * 16;
            float* const wst = p.ks_ws + (long)bid * (ks_S - 1) * ((long)NACC * NT);
            if (ks_part != ks_S - 1) {
                float* dst = wst + (long)ks_part * ((long)NACC * NT) + tid;
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __hip_atomic_store(dst + ((a * TM + b) * 16 + r) * NT, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(p.ks_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (tid == 0) {
                while (__hip_atomic_load(p.ks_cnt + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ks_S - 1) __builtin_amdgcn_s_sleep(8);
                __hip_atomic_store(p.ks_cnt + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (self-cleaning: the next launch / graph replay finds zeros)
            }
            __syncthreads();
            for (int q = 0; q < ks_S - 1; ++q) {          // fixed order: own (last) K range + part 0 + part 1 ...
                const float* src = wst + (long)q * ((long)NACC * NT) + tid;
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[a][b][r] += __hip_atomic_load(src + ((a * TM + b) * 16 + r) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    tile_epilogue<T, NT, TM, TN, EPI, false, UP2, GNS, WN, RESM>(p, acc, lds, m0, n0, wid_s / WN, wid_s % WN, wid_s, lane);
}

// ---- persistent, ring-pipelined variant of the 8-wave tile (256 pixels x 64 TN couts) ------------------------------
// The kernel above keeps two LDS stages of 64 channels and drains the LDS-DMA queue at every step (one barrier per step,
// vmcnt(0) in front of it): with one workgroup per CU a step lasts as long as its stage takes to arrive (2.7 us measured
// against 1.07 us of MFMA work), and the epilogue of a tile overlaps nothing.  Here
//   * the K loop runs in PHASES of 32 channels over a ring of 4 LDS slots; phase q is requested three phases ahead, the
//     wait in front of phase q's barrier is a COUNTED vmcnt that leaves the two younger phases in flight, so a stage has
//     1.5 steps of MFMA time to arrive instead of one and the queue never drains inside a tile;
//   * workgroups are persistent (one per CU) and walk the tiles in an XCD-aware order; before the epilogue of a tile the
//     first two phases of the NEXT tile are requested into ring slots 0 / 1 (the epilogue's LDS transpose lives in
//     slots 2 / 3), so the operand stream keeps flowing while accumulators are converted and stored.
// Accumulation order over K is the same as in the kernel above (16 channels per MFMA, ascending), so results are
// bit-identical.
template <typename T, int TN, int EPI, bool LINEAR, bool ASM_DMA, int MODE, bool GNS = false, int RESM = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_ring_kernel(ConvParams p) {
    constexpr int NT = 512, WN = 2, TM = 2;
    constexpr bool STAG64 = (MODE & 15) == 3;            // 64-channel stages in two buffers, staggered wave groups (below)
    constexpr bool PLAIN64 = (MODE & 15) == 4;           // 64-channel stages in two buffers, conv_igemm_kernel's loop (one barrier per stage) in this persistent shell
    constexpr bool BK64 = STAG64 || PLAIN64;
    constexpr int ABL = (MODE >> 4) & 7;                 // ablation builds of the loop (tools/ab_stag.py --ablate): 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads
    constexpr bool CMK = !LINEAR && (MODE & 128) != 0;   // 3 x 3 convolution with the taps innermost (conv_igemm_kernel's CM producer: same K order, same bits)
    constexpr int BM = 256, BN = WN * TN * 32, BK = BK64 ? 64 : 32;
    constexpr int CPR = BK / 8, ROWB = BK * 2, RPB = 256 / ROWB;      // 4 chunks per 64-byte row, 4 rows per bank row (8 / 2 at 64 channels)
    constexpr int KC = BK / 16;
    constexpr int NSLOT = BK64 ? 2 : 4;
    constexpr int TILE_A = BM * ROWB, TILE_B = BN * ROWB, SLOT = TILE_A + TILE_B;
    constexpr int LDA = (BM * CPR) / NT;                  // 2 LDS-DMA loads per thread per phase: activations
    constexpr int LDB = (BN * CPR) / NT;                  // 2 full rounds of weights ...
    constexpr bool B_TAIL = (BN * CPR) % NT != 0;         // ... + half a round (waves 0..3) when BN = 320
    static_assert((BM * CPR) % NT == 0 && ((BN * CPR) % NT == 0 || (BN * CPR) % NT == NT / 2), "staging pattern");
    constexpr int EPI_ROWB = (EPI == 1 || EPI == 4) ? (TN / 2) * 64 : TN * 64;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EPI_ROWB;
    constexpr int EPI_OFF = BK64 ? SLOT : 2 * SLOT;          // the epilogue's staging starts behind the slots the next tile is prefetched into
    constexpr int RING_BYTES = NSLOT * SLOT > EPI_OFF + EPI_BYTES ? NSLOT * SLOT : EPI_OFF + EPI_BYTES;
    constexpr bool LNF = EPI == 3 || EPI == 4;                 // LayerNorm-folded epilogues: the tile's fp32 column vectors c1 | c2 live in LDS
    constexpr bool BIAS_LDS = EPI == 2 || EPI == 5;            // token-major Linears: the tile's bias slice lives in LDS (T-typed, BN entries)
    constexpr int CVB = LNF ? 2 * BN * 4 : (BIAS_LDS ? BN * 2 : 0);       // bytes of one tile's column vectors
    // (staggered loop: TWO sets -- the next tile's are loaded before and written behind this tile's epilogue, see the tile boundary)
    constexpr int LDS_BYTES = RING_BYTES + (BK64 ? 2 : 1) * CVB;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    float* const cvec0 = (float*)(lds + RING_BYTES);
    float* cvec = cvec0;
    int cpar = 0;                 // (staggered loop) which set of column vectors this tile reads
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const int Hc = p.up ? 2 * p.Hin : p.Hin, Wc = p.up ? 2 * p.Win : p.Win;
    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w;
    const T* zero = (const T*)g_zero_chunk;

    // persistent tile walk: block b sits on XCD b % 8 (observed dispatch order; only speed depends on it).  Round `it`
    // covers gridDim.x consecutive logical tiles, XCD x takes the x-th eighth of them: the tiles in flight on one XCD are
    // neighbours (same pixel rows / neighbouring cout tiles), the chip as a whole works on one contiguous range.
    // Cout groups (p.ngroups = 1, 2, 4 or 8, dividing tiles_n): group g's cout tiles are walked by XCDs g * XG .. g * XG + XG - 1
    // only, over all pixel tiles -- with few large weight tiles (the 640 -> 5120 GEGLU projection: 20 tiles of 327 KB) one
    // group's weights then stay in those XCDs' L2s instead of being re-fetched by every XCD in every round.
    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / p.ngroups, tn_g = p.tiles_n / p.ngroups;
    const int grp = (blockIdx.x % 8) / xg_n;
    const long ntiles = p.nblocks / p.ngroups;              // tiles of one group, index j = pixel tile * tn_g + local cout tile
    const long tile_first = (long)((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const long tile_step = (long)xg_n * per_xcd;
    // (tile indices fit 32 bits -- the launcher refuses more -- and a 64-bit scalar division is ~ 100 instructions, paid three
    //  times per tile boundary)
    auto tile_m0 = [&](long j) { return (long)((uint32_t)j / (uint32_t)tn_g) * BM; };
    auto tile_n0 = [&](long j) { return (grp * tn_g + (int)((uint32_t)j % (uint32_t)tn_g)) * BN; };

    constexpr int RPI = NT / CPR;               // 128 tile rows between a thread's consecutive chunks
    const int srow = tid / CPR;
    const int pd8 = ((tid % CPR) ^ ((srow / RPB) & (CPR - 1))) * 8;        // data chunk (elements) stored at this LDS position
    const int ksteps_per_tap = p.Cin / BK;
    const int nph = p.ntaps * ksteps_per_tap;

    // ---- producer state (LDS-DMA stream of ONE tile; re-initialised for the next tile as soon as this tile's last
    //      phase has been requested)
    uint32_t pyx[LDA], pnv[LDA];
    const T* aptr[LDA];
    uint32_t amask = 0;
    const T* bptr = wg;
    const long bstride = (long)RPI * p.ntaps * p.Cin;
    int tap_p = 0, kk_p = 0;
    // CMK: the centre pixel's pointer per slot (aptr), nine validity bits per slot, wave-uniform element offsets of the current tap
    // (relative to the centre) and of the current channel chunk -- see conv_igemm_kernel
    uint32_t vmask[CMK ? LDA : 1];
    int tapdelta = 0, kofs = 0;         // (32-bit and pinned to SGPRs by readfirstlane at every update: as 64-bit values hipcc kept them
                                        //  in VGPR pairs, spilled one and reloaded it -- a VMEM load -- inside the K loop)
    auto set_tap = [&](int tap) {
        const int dy = p.ntaps == 9 ? tap / 3 : 1, dx = p.ntaps == 9 ? tap % 3 : 1;
#pragma unroll
        for (int i = 0; i < LDA; ++i) {
            int gy = (int)(pyx[i] & 0xffffu) * p.stride + dy - 1 + p.y_off;
            int gx = (int)(pyx[i] >> 16) * p.stride + dx - 1 + p.x_off;
            bool ok = (pnv[i] >> 31) && gy >= 0 && gy < Hc;
            if (p.wrap) {
                gx = gx < 0 ? gx + Wc : (gx >= Wc ? gx - Wc : gx);
            } else {
                ok = ok && gx >= 0 && gx < Wc;
            }
            const int sy = gy >> p.up, sx = gx >> p.up;
            const long off = ok ? (((long)(pnv[i] & 0x7fffffffu) * p.Hin + sy) * p.Win + sx) * p.Cin + pd8 : 0;
            aptr[i] = ok ? xg + off : zero;
            amask = ok ? (amask | (1u << i)) : (amask & ~(1u << i));
        }
    };
    auto init_tile = [&](long tile) {
        const long m0 = tile_m0(tile);
        const int n0 = tile_n0(tile);
        bptr = wg + (long)(n0 + srow) * p.ntaps * p.Cin + pd8;
        tap_p = 0;
        kk_p = 0;
        if constexpr (LINEAR) {
#pragma unroll
            for (int i = 0; i < LDA; ++i) {
                const long m = m0 + srow + i * RPI;
                const bool ok = m < p.M;
                aptr[i] = ok ? xg + m * p.Cin + pd8 : zero;
                amask = ok ? (amask | (1u << i)) : (amask & ~(1u << i));
            }
        } else if constexpr (CMK) {
#pragma unroll
            for (int i = 0; i < LDA; ++i) {
                const long m = m0 + srow + i * RPI;
                const bool valid = m < p.M;
                const uint32_t mm = valid ? (uint32_t)m : 0u;          // (pixel indices fit 32 bits: host-checked)
                const uint32_t t = mm / (uint32_t)p.Wout;
                const int cy = (int)(t % (uint32_t)p.Hout) * p.stride + p.y_off, cx = (int)(mm % (uint32_t)p.Wout) * p.stride + p.x_off;
                uint32_t vm = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int gy = cy + tp / 3 - 1, gx = cx + tp % 3 - 1;
                    vm |= (gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) ? (1u << tp) : 0u;
                }
                vmask[i] = valid ? vm : 0u;
                aptr[i] = valid ? xg + (((long)(t / (uint32_t)p.Hout) * p.Hin + cy) * p.Win + cx) * p.Cin + pd8 : xg;
            }
            tapdelta = __builtin_amdgcn_readfirstlane(-(p.Win + 1) * p.Cin);          // tap 0 = (dy, dx) = (-1, -1)
            kofs = 0;
        } else {
#pragma unroll
            for (int i = 0; i < LDA; ++i) {
                const long m = m0 + srow + i * RPI;
                const bool valid = m < p.M;
                const long mm = valid ? m : 0;
                const long t = mm / p.Wout;
                pyx[i] = (uint32_t)(t % p.Hout) | ((uint32_t)(mm % p.Wout) << 16);
                pnv[i] = (uint32_t)(t / p.Hout) | (valid ? 0x80000000u : 0u);
            }
            set_tap(0);
        }
    };
    const uint32_t lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
    auto dma = [&](const T* src, int off) {       // off: byte offset inside the LDS array (wave-uniform)
        if constexpr (ASM_DMA) {
            lds_dma16_asm(src, lds_u32 + (uint32_t)off);
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
        }
    };
    // one phase = LDA + LDB (+1 for waves 0..3 when B_TAIL) LDS-DMA instructions per wave ("pieces"), then the pointer bumps
    constexpr int NPIECE = LDA + LDB + (B_TAIL ? 1 : 0);
    auto issue_piece = [&](int slot, auto ic) {
        constexpr int i = decltype(ic)::value;
        const int abase = slot * SLOT + wid_s * 1024;
        const int bbase = abase + TILE_A;
        if constexpr (i < LDA) {
            if constexpr (CMK) {
                dma(((vmask[i] >> tap_p) & 1u) ? aptr[i] + (long)(tapdelta + kofs) : zero, abase + i * (NT * 16));
            } else {
                dma(aptr[i], abase + i * (NT * 16));
                aptr[i] += ((amask >> i) & 1u) ? BK : 0;
            }
        } else if constexpr (i < LDA + LDB) {
            if constexpr (CMK) dma(bptr + (i - LDA) * bstride + (long)(tap_p * p.Cin + kofs), bbase + (i - LDA) * (NT * 16));
            else
            dma(bptr + (i - LDA) * bstride, bbase + (i - LDA) * (NT * 16));
        } else {
            // rows 256 .. 319 of the weight tile: one more KiB for each of the first four waves
            if (wid_s < NT / 128) dma(bptr + LDB * bstride, bbase + LDB * (NT * 16));
        }
    };
    auto issue_advance = [&]() {
        if constexpr (CMK) {
            int tp = tap_p + 1, td = tapdelta + p.Cin, ko = kofs;
            if (tp == 3 || tp == 6) td += (p.Win - 3) * p.Cin;
            if (tp == 9) {
                tp = 0;
                td = -(p.Win + 1) * p.Cin;
                ko += BK;
            }
            tap_p = __builtin_amdgcn_readfirstlane(tp);
            tapdelta = __builtin_amdgcn_readfirstlane(td);
            kofs = __builtin_amdgcn_readfirstlane(ko);
            return;
        }
        bptr += BK;
        if constexpr (!LINEAR) {
            if (++kk_p == ksteps_per_tap) {
                kk_p = 0;
                if (++tap_p < p.ntaps) set_tap(tap_p);
            }
        }
    };
    auto issue = [&](int slot) {
        static_for<NPIECE>([&](auto ic) { issue_piece(slot, ic); });
        issue_advance();
    };
    // counted wait: at most `k` (0, 1, 2) of this wave's most recently requested phases may still be in flight.  The
    // count is an immediate, the per-phase number of LDS-DMA instructions differs between the wave halves (B_TAIL).
    auto wait_inflight = [&](int k) {
        constexpr int N0 = LDA + LDB;
        if (B_TAIL && wid_s < NT / 128) {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (N0 + 1)) : "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0 + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N0) : "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    // fragment byte offsets inside a slot (see the kernel above): row R, K chunk d -> R * ROWB + ((d ^ swz(R)) * 16)
    const int swz = (col / RPB) & (CPR - 1);
    int koff[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) koff[kc] = ((kc * 2 + hi) ^ swz) * 16;
    const int wrow = (wn * (TN * 32) + col) * ROWB + TILE_A, xrow = (wm * (TM * 32) + col) * ROWB;

    // (staggered loop) the column vectors of the tile at (m0n, n0n) -> set `set`, by LDS-DMA: no register carries them across the
    // epilogue and hipcc's wait bookkeeping never sees the load (a visible one made it put vmcnt(0) in front of every fragment read
    // of the K loop).  Thread t supplies bytes [16 t, 16 t + 16) of the set: LNF c1 | c2 (fp32; with a table, its row for this tile,
    // which includes c2: host-folded), BIAS_LDS the T-typed bias slice (zeros without a bias).
    auto cv_fill = [&](long m0n, int n0n, int set) {
        constexpr int NL = LNF ? 2 * (BN / 4) : BN / 8;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_u32 + (uint32_t)(RING_BYTES + set * CVB + wid_s * 1024));
        if (tid < NL) {
            const void* src;
            if constexpr (LNF) {
                const int v = tid / (BN / 4), idx = (tid % (BN / 4)) * 4;
                const float* c2 = p.ln_tab ? p.ln_tab + ((m0n / p.tab_div) % p.tab_mod) * (long)p.Cout : p.ln_c2;
                src = (v ? c2 : p.ln_c1) + n0n + idx;
            } else {
                src = p.bias ? (const void*)((const T*)p.bias + n0n + tid * 8) : (const void*)zero;
            }
            lds_dma16_asm(src, dst);
        }
    };
    long tile = tile_first;
    if (tile >= ntiles) return;
    init_tile(tile);
    issue(0);
    if (!BK64 && nph > 1) issue(1);
    if constexpr (BK64 && (LNF || BIAS_LDS)) cv_fill(tile_m0(tile), tile_n0(tile), 0);       // the first tile's column vectors (set 0)
    bool prev_full = false;       // (staggered loop) the previous tile of this workgroup stored all of its rows
    for (;;) {
        const long m0 = tile_m0(tile);
        const int n0 = tile_n0(tile);
        // phase 2 goes into a slot the previous tile's epilogue used: every wave has to be out of it
        if constexpr (!BK64) asm volatile("s_barrier" ::: "memory");
        if constexpr (BK64) {
            cvec = cvec0 + cpar * (CVB / 4);
        } else
        if constexpr (LNF) {
            // this tile's column vectors -> LDS (read by the epilogue, many barriers from here): threads 0 .. BN/4-1 take c1,
            // the next BN/4 take c2 + the tile's table row (rows of a 256-row tile share one: tab_div % 256 == 0, host-checked)
            if (tid < BN / 2) {
                const int v = tid / (BN / 4), idx = (tid % (BN / 4)) * 4;
                const float* c2 = p.ln_tab ? p.ln_tab + ((m0 / p.tab_div) % p.tab_mod) * (long)p.Cout : p.ln_c2;      // (table rows include c2)
                *(f32x4*)(cvec + v * BN + idx) = *(const f32x4*)((v ? c2 : p.ln_c1) + n0 + idx);
            }
        }
        if constexpr (BIAS_LDS && !BK64) {
            // the tile's bias slice -> LDS: the epilogue's five rounds per 32-row block each started with a dependent global
            // load (an L2 round trip the ten-phase K loop of these GEMMs cannot hide)
            if (tid < BN / 8) {
                const uint4 v = p.bias ? *(const uint4*)((const T*)p.bias + n0 + tid * 8) : uint4{0u, 0u, 0u, 0u};
                *(uint4*)((char*)cvec + tid * 16) = v;
            }
        }
        if (!BK64 && nph > 2) issue(2);

        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        if constexpr (MODE == 1) {
            // Two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD) run the phases ONE BARRIER INTERVAL APART:
            // a phase is split into a read interval R (request phase + 3, fetch all 14 fragments of this phase, counted
            // wait for the next phase's LDS-DMA) and a compute interval C (20 MFMAs from registers) with a barrier after
            // each; while one group is in C the other is in R, so every SIMD always has one wave feeding the matrix pipe
            // while the other hides its LDS latency and its barrier skew.  Hazards across the groups: the leading group's
            // interval j + 1 runs beside the trailing group's interval j, and R(p + 1) / C(p) never touch the same LDS (C is
            // register-only; R(p) requests into the slot last read in R(p - 1), one barrier earlier for both groups).
            const int grp = wid_s >> 2;
            wait_inflight(nph > 2 ? 1 : 0);               // phase 0 (and whatever the previous epilogue left in the queue)
            asm volatile("s_barrier" ::: "memory");
            if (grp) asm volatile("s_barrier" ::: "memory");          // the trailing group skips one interval
            for (int ph = 0; ph < nph; ++ph) {
                if (ph + 3 < nph) issue((ph + 3) & (NSLOT - 1));
                const char* at = lds + (ph & (NSLOT - 1)) * SLOT;
                u32x4 xf[KC][TM], wf[KC][TN];
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(at + xrow + koff[kc] + b * (32 * ROWB));
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(at + wrow + koff[kc] + a * (32 * ROWB));
                }
                // phase ph + 1 has to be in LDS (for every wave) one barrier before anybody reads it
                if (ph + 1 < nph) {
                    const int last = ph + 3 < nph - 1 ? ph + 3 : nph - 1;
                    wait_inflight(last - (ph + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int a = 0; a < TN; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b)
                            acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
            }
            if (!grp) asm volatile("s_barrier" ::: "memory");         // the leading group waits for the trailing one: aligned again
        } else if constexpr (STAG64) {
            // Round 4 (tools/dma_seg_probe.hip, profiles/r04_dma_seg_probe.txt): an LDS-DMA instruction that covers 64-byte
            // row segments -- the 32-channel phases of the ring above -- gets 26 B/clk/CU out of the L2, one that covers whole
            // 128-byte lines 52: the 36.9 KB of a 32-channel phase take longer to come out of the L2 (1400 clk) than its MFMAs
            // take (1280), whatever the schedule.  Here a STAGE is 64 channels (whole lines), two stage buffers; its four
            // 16-channel chunks run as two (R, C) interval pairs like MODE 1's: R fetches the 14 fragments of two chunks, C runs
            // their 20 MFMAs from registers, a barrier after each, and the two wave groups (waves 0-3 / 4-7, one of each per
            // SIMD) run one interval apart, so every SIMD has one wave in C while the other reads LDS.  In barrier intervals
            // t (leading group: R(st,0) at 4 st, C(st,0) at 4 st + 1, R(st,1) at + 2, C(st,1) at + 3; trailing group one later):
            //   * stage st + 1 goes into the buffer stage st - 1 was read from, last by the trailing group's R(st-1,1) at
            //     4 st - 1: both groups request it during t = 4 st and 4 st + 1 (the leading one pieces 0-4 in R(st,0) and 5-8
            //     between the MFMAs of C(st,0); the trailing one 0-4 between the MFMAs of C(st-1,1) and 5-8 in R(st,0)), so the
            //     CU's request path sees half a stage per interval;
            //   * every wave waits for its pieces (vmcnt(0): nothing younger is in flight) in front of the barrier that ends
            //     t = 4 st + 3 (leading: end of C(st,1); trailing: end of R(st,1)); the first read is at t = 4 st + 4.
            const int grp = wid_s >> 2;
            constexpr int PA = (NPIECE + 1) / 2;           // pieces 0 .. PA-1 in the first of the two request intervals
            // Stage 0 was requested in front of the previous tile's epilogue, whose stores are the youngest entries of this wave's
            // queue.  vmcnt retires in order (hipcc's own counted waits in the epilogue rely on it), so once at most NTAIL entries
            // remain -- NTAIL below the number of stores every wave issues for a FULL tile -- stage 0 has landed, and the wave does
            // not sit out the drain of its last stores (3500 - 5000 of a K = 320 tile's 45 000 - 59 000 cycles,
            // tools/patches/ring_cycle_stamps.patch).  Tiles cut by M store fewer rows: vmcnt(0) behind those.  The same wait makes
            // this wave's share of the next column vectors (written behind the epilogue) visible after the barrier.
            constexpr int NTAIL = (EPI == 1 || EPI == 4) ? 6 : 16;
            if (prev_full) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NTAIL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (grp) {
                if (nph > 1 && !(ABL & 1)) static_for<PA>([&](auto ic) { issue_piece(1, ic); });
                asm volatile("s_barrier" ::: "memory");          // the trailing group skips one interval
            }
            for (int st = 0; st < nph; ++st) {
                const char* at = lds + (st & 1) * SLOT;
                const bool req1 = st + 1 < nph && !(ABL & 1), req2 = st + 2 < nph && !(ABL & 1);
                const int rs1 = (st + 1) & 1, rs2 = st & 1;
                u32x4 xf[2][TM], wf[2][TN];
                // ---- R(st, 0)
                if (req1) {
                    if (grp) {
                        static_for<NPIECE - PA>([&](auto ic) { issue_piece(rs1, std::integral_constant<int, PA + decltype(ic)::value>{}); });
                        issue_advance();
                    } else {
                        static_for<PA>([&](auto ic) { issue_piece(rs1, ic); });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ABL & 4) {             // (ablation build: constant fragments instead of the LDS reads)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                        for (int b = 0; b < TM; ++b) xf[kc][b] = u32x4{(uint32_t)st, 1u, 2u, 3u};
#pragma unroll
                        for (int a = 0; a < TN; ++a) wf[kc][a] = u32x4{(uint32_t)a, 1u, (uint32_t)st, 3u};
                    }
                } else
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(at + xrow + koff[kc] + b * (32 * ROWB));
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(at + wrow + koff[kc] + a * (32 * ROWB));
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- C(st, 0): the leading group's pieces PA .. NPIECE-1 of stage st + 1 between the MFMAs
                __builtin_amdgcn_s_setprio(1);
                static_for<2 * TN * TM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                    if (!(ABL & 2)) acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]); else { keep_alive(wf[kc][a]); keep_alive(xf[kc][b]); }
                    if constexpr (m % 4 == 3 && m / 4 < NPIECE - PA) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (req1 && !grp) issue_piece(rs1, std::integral_constant<int, PA + m / 4>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (req1 && !grp) issue_advance();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                // ---- R(st, 1)
                if (!(ABL & 4))
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(at + xrow + koff[2 + kc] + b * (32 * ROWB));
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(at + wrow + koff[2 + kc] + a * (32 * ROWB));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing group: its pieces of stage st + 1
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- C(st, 1): the trailing group's pieces 0 .. PA-1 of stage st + 2 between the MFMAs
                __builtin_amdgcn_s_setprio(1);
                static_for<2 * TN * TM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                    if (!(ABL & 2)) acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]); else { keep_alive(wf[kc][a]); keep_alive(xf[kc][b]); }
                    if constexpr (m % 4 == 3 && m / 4 < PA) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (req2 && grp) issue_piece(rs2, std::integral_constant<int, m / 4>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // leading group: its pieces of stage st + 1
                asm volatile("s_barrier" ::: "memory");
            }
            if (!grp) asm volatile("s_barrier" ::: "memory");         // the leading group waits for the trailing one: aligned again
        } else if constexpr (PLAIN64) {
            // conv_igemm_kernel's K loop (a stage of 64 channels per barrier, the next stage requested right behind the barrier, the
            // fragments of 16-channel chunk kc + 1 fetched while the MFMAs of chunk kc run: 48 fragment registers, which leaves room
            // for the chunk-major convolution producer) inside this kernel's persistent shell: the next tile's first stage and column
            // data are requested under the epilogue, no workgroup turnover between tiles, the tile-start wait leaves the last
            // stores in flight.  Same accumulation order as every other loop.
            constexpr int NTAIL = (EPI == 1 || EPI == 4) ? 6 : 16;
            if (prev_full) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NTAIL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            for (int st = 0; st < nph; ++st) {
                if (st > 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");       // stage st landed for every wave; buffer (st + 1) & 1 is free again
                }
                if (st + 1 < nph) issue((st + 1) & 1);
                const char* at = lds + (st & 1) * SLOT;
                constexpr int LATE = TN >= 4 ? 2 : 0;
                u32x4 wf[2][TN - LATE], xf[2][TM], wl[LATE ? LATE : 1];
                auto fetch = [&](auto kcc) {
                    constexpr int kc = decltype(kcc)::value;
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[kc & 1][b] = *(const u32x4*)(at + xrow + koff[kc] + b * (32 * ROWB));
#pragma unroll
                    for (int a = 0; a < TN - LATE; ++a) wf[kc & 1][a] = *(const u32x4*)(at + wrow + koff[kc] + a * (32 * ROWB));
                };
                fetch(std::integral_constant<int, 0>{});
                static_for<KC>([&](auto kcc) {
                    constexpr int kc = decltype(kcc)::value;
#pragma unroll
                    for (int a = 0; a < LATE; ++a) wl[a] = *(const u32x4*)(at + wrow + koff[kc] + (TN - LATE + a) * (32 * ROWB));
                    if constexpr (kc + 1 < KC) fetch(std::integral_constant<int, kc + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < TN; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b)
                            acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, a < TN - LATE ? wf[kc & 1][a < TN - LATE ? a : 0] : wl[a >= TN - LATE ? a - (TN - LATE) : 0]),
                                                        __builtin_bit_cast(uint4, xf[kc & 1][b]), acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        } else if constexpr (MODE == 2) {
            // The LDS-DMA path of a CU moves ~21 B/clk: a wave that issues its 4-5 KiB of a phase back to back sits in
            // the issue queue for most of a phase while its MFMAs wait behind it (ablation: DMA-only and MFMA-only loop
            // times ADD UP when every wave requests right after the barrier).  Here every piece is issued between two
            // groups of MFMAs, so the matrix pipe has work queued while a piece waits for the memory pipeline.
            for (int ph = 0; ph < nph; ++ph) {
                const int rem = nph - 1 - ph;
                wait_inflight(ph == 0 ? (rem >= 2 ? 1 : 0) : (rem >= 2 ? 2 : rem));
                asm volatile("s_barrier" ::: "memory");      // phase ph landed for every wave; slot (ph + 3) & 3 is free again
                const bool req = ph + 3 < nph;
                const int rslot = (ph + 3) & (NSLOT - 1);
                const char* at = lds + (ph & (NSLOT - 1)) * SLOT;
                constexpr int EARLY = TN >= 4 ? TN - 2 : TN;
                u32x4 xf[KC][TM], wf[KC][TN];
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[0][b] = *(const u32x4*)(at + xrow + koff[0] + b * (32 * ROWB));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[0][a] = *(const u32x4*)(at + wrow + koff[0] + a * (32 * ROWB));
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[1][b] = *(const u32x4*)(at + xrow + koff[1] + b * (32 * ROWB));
#pragma unroll
                for (int a = 0; a < EARLY; ++a) wf[1][a] = *(const u32x4*)(at + wrow + koff[1] + a * (32 * ROWB));
                __builtin_amdgcn_sched_barrier(0);
                // piece 0 goes out while the fragment reads are in flight
                if (req) issue_piece(rslot, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<KC * TN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, kc = j / TN, a = j % TN;
                    if constexpr (kc == 1 && a == 0 && EARLY < TN) {
#pragma unroll
                        for (int a2 = EARLY; a2 < TN; ++a2) wf[1][a2] = *(const u32x4*)(at + wrow + koff[1] + a2 * (32 * ROWB));
                    }
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                    // pieces 1 .. NPIECE-1 after MFMA pairs 1, 3, 5, 7 of the ten
                    constexpr int piece = (j % 2 == 1) ? (j / 2 + 1) : -1;
                    if constexpr (piece >= 1 && piece < NPIECE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (req) issue_piece(rslot, std::integral_constant<int, piece>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                if (req) issue_advance();
            }
        } else {
        for (int ph = 0; ph < nph; ++ph) {
            // phase 0 also waits for whatever the previous tile's epilogue left in the queue (its stores sit between
            // the prefetched phases 0 / 1 and phase 2)
            const int rem = nph - 1 - ph;
            wait_inflight(ph == 0 ? (rem >= 2 ? 1 : 0) : (rem >= 2 ? 2 : rem));
            asm volatile("s_barrier" ::: "memory");      // phase ph landed for every wave; slot (ph + 3) & 3 is free again
            if (ph + 3 < nph && !(p.dbg & 1)) issue((ph + 3) & (NSLOT - 1));
            const char* at = lds + (ph & (NSLOT - 1)) * SLOT;
            // fragment reads of the second 16-channel chunk are issued behind the first chunk's (its last weight
            // fragments behind the first MFMAs), which caps the live fragment registers at 44 of the 256
            static_assert(KC == 2, "two chunks per phase");
            constexpr int EARLY = TN >= 4 ? TN - 2 : TN;
            u32x4 xf[KC][TM], wf[KC][TN];
            if (p.dbg & 4) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[kc][b] = u32x4{(uint32_t)ph, 1u, 2u, 3u};
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf[kc][a] = u32x4{(uint32_t)a, 1u, (uint32_t)ph, 3u};
                }
            } else {
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[0][b] = *(const u32x4*)(at + xrow + koff[0] + b * (32 * ROWB));
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[0][a] = *(const u32x4*)(at + wrow + koff[0] + a * (32 * ROWB));
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[1][b] = *(const u32x4*)(at + xrow + koff[1] + b * (32 * ROWB));
#pragma unroll
            for (int a = 0; a < EARLY; ++a) wf[1][a] = *(const u32x4*)(at + wrow + koff[1] + a * (32 * ROWB));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p.dbg & 2) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) keep_alive(xf[kc][b]);
#pragma unroll
                    for (int a = 0; a < EARLY; ++a) keep_alive(wf[kc][a]);
                }
                continue;
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[0][a]), __builtin_bit_cast(uint4, xf[0][b]), acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = EARLY; a < TN; ++a) wf[1][a] = (p.dbg & 4) ? u32x4{1u, 2u, 3u, 4u} : *(const u32x4*)(at + wrow + koff[1] + a * (32 * ROWB));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[1][a]), __builtin_bit_cast(uint4, xf[1][b]), acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        float ln_pre[2 * TM];
        if constexpr (EPI == 3 || EPI == 4) {
            float mus[TM], rstds[TM];
            epi_ln_row_stats<TM>(p, m0, wid_s / WN, lane & 31, mus, rstds);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                ln_pre[b] = mus[b];
                ln_pre[TM + b] = rstds[b];
            }
        }
        asm volatile("s_barrier" ::: "memory");          // every wave is done reading operand slots
        const long next = tile + tile_step;
        if (next < ntiles) {                             // keep the operand stream going under the epilogue
            init_tile(next);
            // (staggered loop) the next tile's column vectors into the OTHER set -- IN FRONT of the stage requests: its address
            // comes back from scratch in some instantiations, and the wait of that reload would also wait for a stage requested
            // a moment ago (the wave that fills the vectors then starts its epilogue a memory latency late)
            if constexpr (BK64 && (LNF || BIAS_LDS)) cv_fill(tile_m0(next), tile_n0(next), cpar ^ 1);
            issue(0);
            if (!BK64 && nph > 1) issue(1);
        }
        // the epilogue's per-lane addressing (rows / pieces / swizzles of ten store rounds) is invariant across tiles: keep
        // the compiler from hoisting ~40 registers of it out of the tile loop (they would be spilled around the K loop)
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        if (p.dbg & 8) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) keep_alive(acc[a][b]);
        } else
        tile_epilogue<T, NT, TM, TN, EPI, true, false, GNS, WN, RESM>(p, acc, lds + EPI_OFF, m0, n0, wid_e / WN, wid_e % WN, wid_e, lane_e, cvec, BN, (EPI == 3 || EPI == 4) ? ln_pre : nullptr);
        if (next >= ntiles) break;
        if constexpr (BK64) {
            cpar ^= 1;
            prev_full = m0 + BM <= p.M;
        }
        tile = next;
    }
}

// ---- the four-wave, register-staged tile for the token-major GEMMs (round 6)
#include "conv3x3_g4.hip"

// ---- ablation-only kernels (`make ablate`, -DIM360_ABLATE): gemm_a3_kernel (two activation stages in flight, knob conv_ring 10) and
//      conv_halo_kernel (halo-patch 3 x 3 convolution, knob conv_halo) with its launcher -- measured, slower, not shipped; they live
//      in conv3x3_ablate.hip and see this file's ConvParams / tile_epilogue / helpers
#ifdef IM360_ABLATE
#define IM360_CONV3X3_INCLUDES_ABLATE 1
#include "conv3x3_ablate.hip"
#endif

template <typename T, int WM, int WN, int TM, int TN, int EPI = 0>
static int launch_conv_t(ConvParams p, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (p.nblocks > 0x7fffffffL) {
        im360_set_error("conv_fwd: problem too large");
        return IM360_ERR_ARG;
    }
    const int bk_env = knob(KNOB_CONV_BK);       // tuning override
    constexpr bool has_bk32 = (BN * 4) % NT == 0 && (BM * 4) % NT == 0 && (EPI != 1 || NT == 256);      // the 8-wave GEGLU tile is BK = 64 only
    // 3x3 convolutions without wrap / upsample addressing: taps innermost (see the kernel); the 256 x 320 and 128 x 128 tiles
    constexpr bool has_cm = EPI == 0 && ((WM == 4 && WN == 2 && TN == 5) || (WM == 2 && WN == 2 && TN == 2) || (WM == 2 && WN == 2 && TM == 3 && TN == 5));
    const bool cm = has_cm && knob(KNOB_CONV_CM) && p.ntaps == 9 && !p.wrap && !p.up && p.Cin % 64 == 0 && bk_env != 32;
    if (!(WM == 4 && WN == 2 && TM == 2 && TN == 5 && EPI == 0) || p.ntaps != 9 || p.up || p.Cin % 64 != 0 || bk_env == 32) p.ksplit = 0;       // K-split: the 256 x 320 tile's 3 x 3 kernels only
    const unsigned grid_cm = (unsigned)(p.nblocks * (p.ksplit > 1 ? p.ksplit : 1));
    p.dbg = knob(KNOB_CONV_DBG);
#ifdef IM360_ABLATE
    if constexpr (((WM == 4 && WN == 2) || (WM == 2 && WN == 2 && TM == 3)) && TN == 5 && EPI == 0) {
        if (p.dbg && p.Cin % 64 == 0) {         // ablation build of the 256 x 320 conv tile (results are garbage)
            if (cm) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, false, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
#endif
    if constexpr (WM == 4 && WN == 2 && TM == 2 && TN == 5 && EPI == 0) {
#ifdef IM360_ABLATE
        // staggered wave groups on this kernel's producer (round 4; knob conv_stag): identical bits, 0.97 - 1.05 x the plain loop's
        // speed on the nine cfg2 convolution shapes (profiles/r04_conv_stag_ab.log) -- with two stage buffers a step's operands have
        // one step to arrive either way, and that latency, not the fragment reads the stagger hides, is what the loop waits for
        if (knob(KNOB_CONV_STAG) && p.Cin % 64 == 0 && bk_env != 32) {
            if (p.gn_out) {
                if (cm) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, true, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
                else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, false, false, false, true, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
            } else {
                if (cm) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, false, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
                else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, false, false, false, false, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
            }
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
#endif
        if (p.gn_out) {            // the 256 x 320 tile with GroupNorm statistics from its epilogue (per residual mode of the epilogue)
            if (cm) {
                if (p.res) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, true, false, 1>), dim3(grid_cm), dim3(NT), 0, stream, p);
                else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, true, false, 2>), dim3(grid_cm), dim3(NT), 0, stream, p);
            } else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, false, false, false, true>), dim3(grid_cm), dim3(NT), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
        if (cm) {
            if (p.res) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, false, false, 1>), dim3(grid_cm), dim3(NT), 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true, false, false, false, false, 2>), dim3(grid_cm), dim3(NT), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (p.gn_out) {
        im360_set_error("conv_fwd: GroupNorm statistics are produced by the 256 x 320 tile only (ask im360_conv_gn_slabs first)");
        return IM360_ERR_UNSUPPORTED;
    }
    if (cm) {
        if constexpr (has_cm)
            hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI, true>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
    } else if ((p.Cin % 64 == 0 && bk_env != 32 && !(WM == 2 && TN >= 4 && TM == 2) && !(WM == 4 && WN == 1 && knob(KNOB_CONV_SMALL) == 0)) || !has_bk32) {      // (the 128 x 320 / 128 x 256 tiles exist for two workgroups per CU: 32-channel stages)
        hipLaunchKernelGGL((conv_igemm_kernel<T, 64, WM, WN, TM, TN, EPI>), dim3(grid_cm), dim3(NT), 0, stream, p);
    } else if constexpr (has_bk32) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, 32, WM, WN, TM, TN, EPI>), dim3((unsigned)p.nblocks), dim3(NT), 0, stream, p);
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// persistent ring kernel: one workgroup per CU (its LDS footprint allows no second one), grid a multiple of 8 (XCDs)
template <typename T, int TN, int EPI, bool LINEAR>
static int launch_ring_t(ConvParams p, hipStream_t stream, int variant) {
    constexpr int BM = 256, BN = 2 * TN * 32;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (p.nblocks > 0x7fffffffL) {                // (the kernel divides tile indices in 32 bits)
        im360_set_error("linear_fwd / conv_fwd: problem too large");
        return IM360_ERR_ARG;
    }
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    p.dbg = knob(KNOB_CONV_DBG);
    const long want = (p.nblocks + 7) / 8 * 8;
    const unsigned grid = (unsigned)(want < ncu ? want : ncu);
    // cout groups of the tile walk (see the kernel): the fewest of 1 / 2 / 4 / 8 that divide tiles_n and bring one group's
    // weights under ~3.4 MB of the XCD's 4 MB L2 (knob ring_groups: 0 = this rule, else forced when it divides tiles_n)
    {
        const long wbytes = (long)p.tiles_n * BN * p.ntaps * p.Cin * 2;
        int ng = 1;
        const int force = knob(KNOB_RING_GROUPS);
        if (force > 0) {
            if ((force == 2 || force == 4 || force == 8) && p.tiles_n % force == 0) ng = force;
        } else {
            while (ng < 8 && wbytes / ng > 3400000L && p.tiles_n % (2 * ng) == 0) ng *= 2;
            if (wbytes / ng > 3400000L) ng = 1;          // no split brings a group under the budget: keep the plain walk
        }
        p.ngroups = grid >= 8u * ng ? ng : 1;
    }
    // variant (knob conv_ring): 1 = default: token-major GEMMs on the staggered 64-channel-stage loop (MODE 3, round 4), convolutions
    // sent here by knob value 5 on the interleaved ring (MODE 2); 8 = MODE 2 for the GEMMs too (round 2 / 3's default), 6 = MODE 3
    // for everything; make ablate: 2 = builtin LDS-DMA, plain ring; 3 = asm LDS-DMA, plain ring; 4 = staggered wave groups on the ring
    const bool stag = p.Cin % 64 == 0 && (variant == 6 || (LINEAR && variant != 8 && (variant < 2 || variant > 4)));
    // the default loop's kernels exist per residual mode of the epilogue (tile_epilogue, RESM): present / absent are different code
    auto launch_stag = [&](auto gns_c) {
        constexpr bool G = decltype(gns_c)::value;
#ifdef IM360_ABLATE
        // measured and not shipped (profiles/r04_gemm_a3_ab.log): identical bits on every shape and epilogue at the first run, but
        // 0.85 - 0.97 x the staggered loop's speed (only the level-0 GEGLU ties): with the weight tile in 64-byte row segments the
        // L2 -> LDS path carries 2170 clk of transfers per stage instead of 1385, and that path's throughput -- not only the latency
        // of the activation rows -- is what the K loop runs against
        if constexpr (LINEAR && EPI != 3) {             // (EPI 3's two sets of c1 | c2 do not fit next to the A3 kernel's 156 KB)
            if (variant == 10) {                        // two activation stages in flight (gemm_a3_kernel)
                if constexpr (EPI == 1 || EPI == 4) {
                    hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 0>), dim3(grid), dim3(512), 0, stream, p);
                } else {
                    if (p.res) hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 1>), dim3(grid), dim3(512), 0, stream, p);
                    else hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 2>), dim3(grid), dim3(512), 0, stream, p);
                }
                return;
            }
            if (variant == 11) {                        // the same with a CHUNK-MAJOR weight operand (the caller repacked it: kernels.chunk_major)
                if constexpr (EPI == 1 || EPI == 4) {
                    hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 0, true>), dim3(grid), dim3(512), 0, stream, p);
                } else {
                    if (p.res) hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 1, true>), dim3(grid), dim3(512), 0, stream, p);
                    else hipLaunchKernelGGL((gemm_a3_kernel<T, TN, EPI, G, 2, true>), dim3(grid), dim3(512), 0, stream, p);
                }
                return;
            }
        }
#endif
        if constexpr (EPI == 1 || EPI == 4) {
            hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3, G, 0>), dim3(grid), dim3(512), 0, stream, p);
        } else if constexpr (EPI == 3) {
            hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3, G, 2>), dim3(grid), dim3(512), 0, stream, p);
        } else {
            if (p.res) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3, G, 1>), dim3(grid), dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3, G, 2>), dim3(grid), dim3(512), 0, stream, p);
        }
    };
#ifdef IM360_ABLATE
    if constexpr (!LINEAR && TN == 5 && EPI == 0) {
        if (variant == 9) {                     // 3 x 3 convolution, taps innermost, on the plain 64-channel loop in the persistent shell (the caller checked that the order applies)
            if (p.gn_out) {
                if (p.res) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 4 + 128, true, 1>), dim3(grid), dim3(512), 0, stream, p);
                else hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 4 + 128, true, 2>), dim3(grid), dim3(512), 0, stream, p);
            } else {
                if (p.res) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 4 + 128, false, 1>), dim3(grid), dim3(512), 0, stream, p);
                else hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 4 + 128, false, 2>), dim3(grid), dim3(512), 0, stream, p);
            }
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
#endif
    if constexpr (LINEAR && TN == 5 && (EPI == 2 || EPI == 5)) {
        if (p.gn_out) {                         // GroupNorm partial sums from the epilogue (see ConvParams::gn_out)
            if (stag) launch_stag(std::true_type{});
            else hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 2, true>), dim3(grid), dim3(512), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (p.gn_out) {
        im360_set_error("linear_fwd: GroupNorm statistics are produced by the plain / row-statistics epilogues only");
        return IM360_ERR_UNSUPPORTED;
    }
#ifdef IM360_ABLATE
    if constexpr (LINEAR && TN == 5 && EPI == 2) {
        if (stag && p.dbg) {                    // ablation builds of the staggered loop (tools/ab_stag.py --ablate)
            switch (p.dbg & 7) {
#define IM360_ABL_CASE(a) case a: hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3 + 16 * a>), dim3(grid), dim3(512), 0, stream, p); break;
                IM360_ABL_CASE(1) IM360_ABL_CASE(2) IM360_ABL_CASE(3) IM360_ABL_CASE(4) IM360_ABL_CASE(5) IM360_ABL_CASE(6) IM360_ABL_CASE(7)
#undef IM360_ABL_CASE
            }
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if constexpr (EPI < 3) {
        if (variant == 2) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, false, 0>), dim3(grid), dim3(512), 0, stream, p);
        else if (variant == 3) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 0>), dim3(grid), dim3(512), 0, stream, p);
        else if (variant == 4) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 1>), dim3(grid), dim3(512), 0, stream, p);
        if (variant >= 2 && variant <= 4) {
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
#endif
    if constexpr (LINEAR) {
        if (stag) launch_stag(std::false_type{});
        else hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 2>), dim3(grid), dim3(512), 0, stream, p);
    } else {
        // (3x3 convolutions on the staggered loop: 27 spilled registers, 0.90 - 1.13 PF/s against the two-stage kernel's 1.05 - 1.16: make ablate only)
#ifdef IM360_ABLATE
        if (stag) hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 3>), dim3(grid), dim3(512), 0, stream, p);
        else
#endif
        hipLaunchKernelGGL((conv_ring_kernel<T, TN, EPI, LINEAR, true, 2>), dim3(grid), dim3(512), 0, stream, p);
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// K-split plan of a convolution launch (ConvParams::ksplit): 1 = none.  Eligible: 3 x 3, taps innermost (no wrap / upsample addressing), the
// 256 x 320 tile.  T tiles of one K range each take ceil(T / 256) rounds of the chip; S parts per tile take ceil(T S / 256) rounds of 1 / S the
// length (+ ~ 4 % per extra part for the partial sums' round trip).  Below 512 tiles the alternative is the 128 x 128 tile, which runs at about
// three quarters of the large tile's rate.  Knob conv_ksplit: 0 = off, 1 = this rule, 2 .. 4 = that many parts wherever eligible.
static int ksplit_plan(long M, int Cin, int Cout, int ntaps, int up, int wrap, bool gn_stats) {
    const int kn = knob(KNOB_CONV_KSPLIT);
    if (kn <= 0 || !knob(KNOB_CONV_BIG) || !knob(KNOB_CONV_CM) || knob(KNOB_CONV_BK) == 32 || knob(KNOB_CONV_RING) >= 5) return 1;
    if (ntaps != 9 || up || Cout % 320 != 0 || Cin % 64 != 0 || M > 0x7fffffffL) return 1;
    if (wrap && kn == 1) return 1;              // (the panorama's wrap-addressed, tap-major launches split correctly but gain nothing inside the step -- they run beside the
                                                //  perspective branch: 290.2 vs 289.9 ms, profiles/r06_conv_ksplit_ab.log -- so the rule leaves them alone; knob 9 = the rule for them too)
    const long T = ((M + 255) / 256) * (Cout / 320);
    const int nch = Cin / 64;
    if (T < 64 || (gn_stats && T < 512)) return 1;
    // Liveness of the waiting owners (conv_igemm_kernel): block ids are part-major and every XCD dispatches its share of them in order, so an XCD
    // starts a launch's owners only after ALL of that launch's parts it was dealt -- one launch alone can never block itself.  Two launches side
    // by side (the two branches' streams) could only block each other if some XCD's 32 CUs were ALL held by waiting owners of ONE launch (those of
    // the other come behind that launch's own parts there): impossible below 32 owners per XCD, i.e. up to 248 tiles.  The rule stays below that;
    // the forced part counts (knobs 2 - 4, 7, 8: single-stream tests and A/B tools) do not.
    if (T > 248 && (kn == 1 || kn == 9)) return 1;
    if (kn >= 2 && kn <= 4) return kn <= nch ? kn : 1;
    // measured inside the step (profiles/r06_conv_ksplit_ab.log): launches below two rounds of the chip -- which otherwise take the 128 x 128 tile --
    // gain (- 3 ... - 4.7 ms per cfg2 step), the 640-tile launches (2.5 rounds -> five half rounds) LOSE 2 ms to the partial sums' round trip:
    // the rule is for the former only (knob 5: for every tile count, the A/B; 7 / 8: two / four parts for the former)
    if (kn == 9) { if (T >= 512) return 1; }
    else if (T >= 512 && kn != 5) return 1;
    if (T < 512 && (kn == 7 || kn == 8)) return (kn == 7 ? 2 : 4) <= nch ? (kn == 7 ? 2 : 4) : 1;
    double best = T >= 512 ? (double)((T + 255) / 256) : 1.35 * (double)T / 256.0 + 0.15;
    int bs = 1;
    for (int S = 2; S <= 4 && S <= nch; ++S) {
        const double c = (double)((T * S + 255) / 256) / S * (1.0 + 0.04 * (S - 1));
        if (c < best * 0.93) {          // (at least 7 % on paper)
            best = c;
            bs = S;
        }
    }
    return bs;
}

template <typename T>
static int launch_conv(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    const int big_env = knob(KNOB_CONV_BIG);           // tuning overrides
    const int ring_env = knob(KNOB_CONV_RING);
    if (!(p.ks_ws && p.ks_cnt && p.ksplit > 1)) p.ksplit = 0;       // (the entry point checked the plan and the buffers)
    // 256 x 320 tiles once they fill the chip at least twice (one workgroup per CU) -- or in K parts (ksplit_plan)
    if (big_env && p.Cout % 320 == 0 && p.Cin % 64 == 0 && (((p.M + 255) / 256) * (p.Cout / 320) >= 512 || p.ksplit > 1)) {
        const bool linear = p.ntaps == 1 && p.Hin == 1 && p.Win == 1 && !p.temb;       // EPI 2 has no temb add
#ifdef IM360_ABLATE
        if (knob(KNOB_CONV_HALO)) {
            ConvParams ph = p;
            if (halo_geometry(ph)) return launch_halo<T>(ph, stream);
        }
        if (big_env == 2) return linear ? launch_conv_t<T, 2, 2, 2, 5, 2>(p, stream) : launch_conv_t<T, 2, 2, 2, 5>(p, stream);   // A/B: 128 x 320, 2 workgroups per CU
        // A/B: the same 256 x 320 tile on FOUR waves (128 x 160 each, 320 accumulators in the unified 512-register file, one
        // wave per SIMD): 144 instead of 224 KB of fragment reads per K step
        if (big_env == 4 && !linear) return launch_conv_t<T, 2, 2, 3, 5>(p, stream);
#endif
        // measured (tools/ab_ring.py, profiles/README.md): the persistent ring kernel wins 3-8 % on the token-major GEMMs
        // (short K, epilogue-heavy) and loses 1-8 % on the deep-K convolutions; knob value 5 forces it for both
#ifdef IM360_ABLATE
        // round 4, measured and not shipped: 3 x 3 convolutions with the taps innermost on the PERSISTENT kernel (next tile's first
        // stage requested under the epilogue, no workgroup turnover between tiles; knob conv_persist).  Identical bits on nine
        // shapes incl. stride 2 and the statistics epilogue.  First form 0.93 - 1.04 x conv_igemm_kernel's speed: hipcc kept the 64-bit
        // tap / chunk offsets in VGPR pairs, spilled one and reloaded it inside the K loop -- a scratch reload is a VMEM load whose
        // wait also waits for the stage requested in front of it.  With the offsets 32-bit and pinned to SGPRs the loop is clean and
        // the kernel TIES: 0.96 - 1.05 x (profiles/r04_conv_persist_ab.log) -- workgroup turnover is not what the convolutions lose.
        if (knob(KNOB_CONV_PERSIST) && !linear && knob(KNOB_CONV_CM) && p.ntaps == 9 && !p.wrap && !p.up && !p.x2 && p.Cin % 64 == 0 && knob(KNOB_CONV_BK) != 32 && p.M <= 0x7fffffffL)
            return launch_ring_t<T, 5, 0, false>(p, stream, 9);
#endif
        if (p.gn_out && !linear) return launch_conv_t<T, 4, 2, 2, 5>(p, stream);
        if (ring_env && linear) return launch_ring_t<T, 5, 2, true>(p, stream, ring_env == 5 ? 1 : (ring_env == 7 ? 6 : ring_env));
        if (ring_env >= 5) return launch_ring_t<T, 5, 0, false>(p, stream, ring_env == 7 ? 6 : 1);
        if (linear) return launch_conv_t<T, 4, 2, 2, 5, 2>(p, stream);
        return launch_conv_t<T, 4, 2, 2, 5>(p, stream);
    }
    // Cout % 128 in (0, 64] (the UNet's 320 on grids too small for the 256 x 320 tile, i.e. every cfg1-sized launch): rounds 1 - 2
    // used 256 x 64 tiles so that no cout tile is half empty.  Since round 3 the 64-channel instantiation of that tile spills 688
    // bytes inside its K loop (ConvParams grew; hipcc aims it at 168 registers whatever amdgpu_waves_per_eu says): 1.98 ms for a
    // 320 -> 320 convolution of 320 16 x 16 images -- cfg1's conv class went 21.8 -> 61.1 ms unnoticed, the headline config does not
    // use this tile.  Measured (bench_kernels.py conv_small): 256 x 64 with 32-channel K steps 0.281 ms, plain 128 x 128 tiles
    // 0.228 ms (taps innermost, 17 % of the MFMA work wasted on the half-empty tile all the same).  Knob conv_small: 2 = 128 x 128
    // (default), 0 = 256 x 64 / 32 channels, 1 = 256 x 64 / 64 channels.
    const int rem = p.Cout % 128;
    const int small = knob(KNOB_CONV_SMALL);
    if (rem != 0 && rem <= 64 && p.Cout > 64 && small != 2) return launch_conv_t<T, 4, 1, 2, 2>(p, stream);
    return launch_conv_t<T, 2, 2, 2, 2>(p, stream);
}

// weights [Cout, Cin, kh, kw] (PyTorch) -> [CoutPad128][kh*kw][CinPad] zero padded, K contiguous
template <typename T>
__global__ void pack_conv_weight_kernel(const T* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int taps,
                                        int CoutPad, int CinPad) {
    const long total = (long)CoutPad * taps * CinPad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinPad);
        const long t2 = i / CinPad;
        const int tap = (int)(t2 % taps);
        const int co = (int)(t2 / taps);
        T v = from_f32<T>(0.f);
        if (co < Cout && ci < Cin) v = w[((long)co * Cin + ci) * taps + tap];
        out[i] = v;
    }
}

}  // namespace im360

// Slabs per image of the GroupNorm partial sums a conv / token-major linear launch of these dimensions can write from its
// epilogue (gn_partial: fp32 [N * S][2][Cout]), or 0 when it cannot: needs the 256 x 320 tile (Cout % 320 == 0, Cin % 64 == 0,
// at least 512 tiles) and images that are whole numbers of 256-pixel tiles.  For a linear, pass the [N, Hout, Wout] image
// shape of its token rows and ntaps = 1.
extern "C" __attribute__((visibility("default"))) int64_t im360_conv_gn_slabs(int64_t N, int64_t Hout, int64_t Wout, int64_t Cin, int64_t Cout, int64_t ntaps) {
    using namespace im360;
    const int64_t hw = Hout * Wout, M = N * hw;
    if (N <= 0 || hw <= 0 || (hw % 256) != 0 || (Cout % 320) != 0 || (Cin % 64) != 0 || (ntaps != 1 && ntaps != 9)) return 0;
    if (!knob(KNOB_CONV_BIG) || (M / 256) * (Cout / 320) < 512) return 0;
    return hw / 256;
}

static int conv_fwd_impl(const void* x, const void* w_packed, const void* bias, const void* temb,
                         const void* res, void* y,
                         int64_t N, int64_t Hin, int64_t Win, int64_t Cin,
                         int64_t Hout, int64_t Wout, int64_t Cout, int64_t ntaps,
                         int64_t stride, int64_t up, int64_t wrap, int64_t x_off, int64_t y_off,
                         int64_t imgs_per_temb, int dtype, void* stream, void* gn_partial,
                         void* ks_ws, int64_t ks_ws_bytes, void* ks_cnt, int64_t ks_cnt_n) {
    using namespace im360;
    IM360_CHECK_ARG(x && w_packed && y, "conv_fwd: null pointer");
    IM360_CHECK_ARG(!gn_partial || (im360_conv_gn_slabs(N, Hout, Wout, Cin, Cout, ntaps) > 0 && !up), "conv_fwd: this launch cannot produce GroupNorm statistics (im360_conv_gn_slabs == 0)");
    IM360_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Cout > 0, "conv_fwd: empty problem");
    IM360_CHECK_ARG(Cin > 0 && (Cin % 32) == 0, "conv_fwd: Cin=%ld must be a multiple of 32 (pad channels)", (long)Cin);
    IM360_CHECK_ARG(ntaps == 9 || ntaps == 1, "conv_fwd: ntaps must be 9 or 1");
    IM360_CHECK_ARG(stride == 1 || stride == 2, "conv_fwd: stride must be 1 or 2");
    IM360_CHECK_ARG(!(up && stride != 1), "conv_fwd: upsample input requires stride 1");
    IM360_CHECK_ARG(!temb || imgs_per_temb > 0, "conv_fwd: imgs_per_temb must be positive");
    IM360_CHECK_ARG(Hout <= 0xffff && Wout <= 0xffff && N <= 0x7fffffffL, "conv_fwd: Hout, Wout must fit 16 bits");
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                    ((uintptr_t)res % 16) == 0 && ((uintptr_t)bias % 8) == 0 && ((uintptr_t)temb % 8) == 0,
                    "conv_fwd: misaligned pointer");
    ConvParams p = conv_params_zero();
    p.x = x; p.w = w_packed; p.bias = bias; p.temb = temb; p.res = res; p.y = y;
    p.N = (int)N; p.Hin = (int)Hin; p.Win = (int)Win; p.Cin = (int)Cin;
    p.Hout = (int)Hout; p.Wout = (int)Wout; p.Cout = (int)Cout; p.ntaps = (int)ntaps;
    p.stride = (int)stride; p.up = up ? 1 : 0; p.wrap = wrap ? 1 : 0; p.x_off = (int)x_off; p.y_off = (int)y_off;
    p.imgs_per_temb = temb ? (int)imgs_per_temb : 1;
    p.M = N * Hout * Wout;
    p.gn_out = (float*)gn_partial;
    if (ks_ws || ks_cnt) {
        const int S = ksplit_plan(p.M, p.Cin, p.Cout, p.ntaps, p.up, p.wrap, gn_partial != nullptr);
        const int64_t tiles = Cout % 320 == 0 ? ((p.M + 255) / 256) * (Cout / 320) : 0;
        IM360_CHECK_ARG(S > 1, "conv_fwd_ksplit: im360_conv_ksplit_plan gives no K-split for this launch");
        IM360_CHECK_ARG(ks_ws && ks_cnt && ((uintptr_t)ks_ws % 16) == 0 && ((uintptr_t)ks_cnt % 4) == 0 && ks_cnt_n >= tiles &&
                        ks_ws_bytes >= tiles * (S - 1) * (int64_t)(160 * 512 * 4),
                        "conv_fwd_ksplit: %ld tiles x %d parts need %ld workspace bytes and %ld zeroed counters", (long)tiles, S,
                        (long)(tiles * (S - 1) * (int64_t)(160 * 512 * 4)), (long)tiles);
        p.ksplit = S;
        p.ks_ws = (float*)ks_ws;
        p.ks_cnt = (int*)ks_cnt;
    }
    hipStream_t s = (hipStream_t)stream;
    // token-major linears routed through the kernel (1x1 taps on a [M, 1, 1, K] view) are accounted separately
    ProfScope prof(ntaps == 1 && Hin == 1 && Win == 1 ? PROF_GEMM : PROF_CONV, stream);
    if (dtype == 0) return launch_conv<__bf16>(p, s);
    if (dtype == 1) return launch_conv<_Float16>(p, s);
    im360_set_error("conv_fwd: dtype %d unsupported", dtype);
    return IM360_ERR_UNSUPPORTED;
}

extern "C" __attribute__((visibility("default"))) int im360_conv_fwd(const void* x, const void* w_packed, const void* bias, const void* temb,
                              const void* res, void* y,
                              int64_t N, int64_t Hin, int64_t Win, int64_t Cin,
                              int64_t Hout, int64_t Wout, int64_t Cout, int64_t ntaps,
                              int64_t stride, int64_t up, int64_t wrap, int64_t x_off, int64_t y_off,
                              int64_t imgs_per_temb, int dtype, void* stream, void* gn_partial) {
    return conv_fwd_impl(x, w_packed, bias, temb, res, y, N, Hin, Win, Cin, Hout, Wout, Cout, ntaps, stride, up, wrap, x_off, y_off, imgs_per_temb, dtype, stream,
                         gn_partial, nullptr, 0, nullptr, 0);
}

// K parts per output tile im360_conv_fwd_ksplit would run this launch in (1 = none: call im360_conv_fwd), knob conv_ksplit included.
extern "C" __attribute__((visibility("default"))) int64_t im360_conv_ksplit_plan(int64_t N, int64_t Hout, int64_t Wout, int64_t Cin, int64_t Cout, int64_t ntaps,
                                                                                 int64_t up, int64_t wrap, int64_t gn_stats) {
    using namespace im360;
    if (N <= 0 || Hout <= 0 || Wout <= 0 || Cin <= 0 || Cout <= 0 || Cin > 0x7fffffffL || Cout > 0x7fffffffL) return 1;
    return ksplit_plan(N * Hout * Wout, (int)Cin, (int)Cout, (int)ntaps, up ? 1 : 0, wrap ? 1 : 0, gn_stats != 0);
}

// im360_conv_fwd with every output tile's K range split over im360_conv_ksplit_plan(...) workgroups (ConvParams::ksplit).  ks_ws: tiles x (parts - 1)
// x 327 680 bytes of scratch (tiles = ceil(N Hout Wout / 256) x Cout / 320); ks_cnt: `tiles` int32 counters, ZERO on entry, zero again on return.
// Deterministic (fixed summation order), not bit-identical to the unsplit launch (another order of the fp32 partial sums).
extern "C" __attribute__((visibility("default"))) int im360_conv_fwd_ksplit(const void* x, const void* w_packed, const void* bias, const void* temb,
                              const void* res, void* y,
                              int64_t N, int64_t Hin, int64_t Win, int64_t Cin,
                              int64_t Hout, int64_t Wout, int64_t Cout, int64_t ntaps,
                              int64_t stride, int64_t up, int64_t wrap, int64_t x_off, int64_t y_off,
                              int64_t imgs_per_temb, int dtype, void* stream, void* gn_partial,
                              void* ks_ws, int64_t ks_ws_bytes, void* ks_cnt, int64_t ks_cnt_n) {
    using namespace im360;
    IM360_CHECK_ARG(ks_ws && ks_cnt, "conv_fwd_ksplit: null workspace");
    return conv_fwd_impl(x, w_packed, bias, temb, res, y, N, Hin, Win, Cin, Hout, Wout, Cout, ntaps, stride, up, wrap, x_off, y_off, imgs_per_temb, dtype, stream,
                         gn_partial, ks_ws, ks_ws_bytes, ks_cnt, ks_cnt_n);
}

// nearest-x2 upsample + conv3x3 (pad 1) as four 2 x 2 convolutions of the low-resolution input, one per output parity:
// x [N, Hin, Win, Cin] -> y [N, 2 Hin, 2 Win, Cout]; w4 = four packed 4-tap weights [4][CoutPad][4][Cin] in parity order
// (py, px) = (0,0) (0,1) (1,0) (1,1), each the sum of the 3 x 3 taps that land on the same source pixel.
extern "C" __attribute__((visibility("default"))) int im360_conv_up2_fwd(const void* x, const void* w4, const void* bias, void* y, int64_t N, int64_t Hin,
                                  int64_t Win, int64_t Cin, int64_t Cout, int64_t wrap, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && w4 && y, "conv_up2_fwd: null pointer");
    IM360_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Cout > 0 && Cin > 0 && (Cin % 64) == 0, "conv_up2_fwd: Cin=%ld must be a positive multiple of 64", (long)Cin);
    IM360_CHECK_ARG((Cout % 8) == 0, "conv_up2_fwd: Cout=%ld must be a multiple of 8", (long)Cout);
    IM360_CHECK_ARG(2 * Hin <= 0xffff && 2 * Win <= 0xffff && N <= 0x7fffffffL, "conv_up2_fwd: image too large");
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w4 % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 8) == 0,
                    "conv_up2_fwd: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "conv_up2_fwd: dtype %d unsupported", dtype);
    ConvParams p = conv_params_zero();
    p.x = x; p.bias = bias; p.temb = nullptr; p.res = nullptr; p.y = y;
    p.N = (int)N; p.Hin = (int)Hin; p.Win = (int)Win; p.Cin = (int)Cin;
    p.Hout = (int)Hin; p.Wout = (int)Win; p.Cout = (int)Cout; p.ntaps = 4;          // the tile walks the LOW-resolution grid
    p.stride = 1; p.up = 0; p.wrap = wrap ? 1 : 0; p.x_off = 0; p.y_off = 0; p.imgs_per_temb = 1;
    p.M = N * Hin * Win;
    p.dbg = 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_CONV, stream);
    const long cout_pad = (Cout + 127) / 128 * 128;
    const size_t esize = 2;
    const bool big = Cout % 320 == 0 && ((p.M + 255) / 256) * (Cout / 320) >= 512;
    for (int parity = 0; parity < 4; ++parity) {
        p.w = (const char*)w4 + (size_t)parity * cout_pad * 4 * Cin * esize;
        p.up2_py = parity >> 1;
        p.up2_px = parity & 1;
        if (big) {
            constexpr int BM = 256, BN = 320;
            p.tiles_n = (p.Cout + BN - 1) / BN;
            p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
            if (dtype == 0) hipLaunchKernelGGL((conv_igemm_kernel<__bf16, 64, 4, 2, 2, 5, 0, false, false, true>), dim3((unsigned)p.nblocks), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<_Float16, 64, 4, 2, 2, 5, 0, false, false, true>), dim3((unsigned)p.nblocks), dim3(512), 0, s, p);
        } else {
            constexpr int BM = 128, BN = 128;
            p.tiles_n = (p.Cout + BN - 1) / BN;
            p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
            if (dtype == 0) hipLaunchKernelGGL((conv_igemm_kernel<__bf16, 64, 2, 2, 2, 2, 0, false, false, true>), dim3((unsigned)p.nblocks), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<_Float16, 64, 2, 2, 2, 2, 0, false, false, true>), dim3((unsigned)p.nblocks), dim3(256), 0, s, p);
        }
        IM360_CHECK_LAUNCH();
    }
    return IM360_OK;
}

extern "C" __attribute__((visibility("default"))) int im360_linear_geglu(const void* x, const void* w_packed, const void* bias_packed, void* y,
                                  int64_t M, int64_t K, int64_t I, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && w_packed && y, "linear_geglu: null pointer");
    IM360_CHECK_ARG(M > 0 && K > 0 && (K % 64) == 0, "linear_geglu: K=%ld must be a positive multiple of 64", (long)K);
    IM360_CHECK_ARG(I > 0 && (I % 128) == 0, "linear_geglu: I=%ld must be a positive multiple of 128", (long)I);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                    ((uintptr_t)bias_packed % 8) == 0, "linear_geglu: misaligned pointer");
    ConvParams p = conv_params_zero();
    p.x = x; p.w = w_packed; p.bias = bias_packed; p.temb = nullptr; p.res = nullptr; p.y = y;
    p.N = (int)M; p.Hin = 1; p.Win = 1; p.Cin = (int)K; p.Hout = 1; p.Wout = 1; p.Cout = (int)(2 * I); p.ntaps = 1;
    p.stride = 1; p.up = 0; p.wrap = 0; p.x_off = 0; p.y_off = 0; p.imgs_per_temb = 1;
    p.M = M;
    IM360_CHECK_ARG(M <= 0x7fffffffL, "linear_geglu: M too large");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, stream);
#ifdef IM360_ABLATE
    if (knob(KNOB_CONV_BIG) == 3) {            // A/B: 128 x 256 tiles, two 4-wave workgroups per CU (one's GELU epilogue under the other's K loop)
        if (dtype == 0) return launch_conv_t<__bf16, 2, 2, 2, 4, 1>(p, s);
        if (dtype == 1) return launch_conv_t<_Float16, 2, 2, 2, 4, 1>(p, s);
    }
#endif
    // the four-wave register-staged tile: knob g4 bit 0 (or, A/B tools, conv_ring 12)
    if (((knob(KNOB_G4) & 1) || knob(KNOB_CONV_RING) == 12) && (K % 64) == 0 && K >= 128 && (M % 256) == 0 && ((2 * I) % 256) == 0 && (M / 256) * (2 * I / 256) >= 256)
        return dtype == 0 ? launch_g4_t<__bf16, 1>(p, s) : launch_g4_t<_Float16, 1>(p, s);
    if (((knob(KNOB_G4) & 4) || knob(KNOB_CONV_RING) == 13) && (K % 64) == 0 && K >= 128 && (M % 256) == 0 && (M / 256) * (2 * I / 128) >= 512)       // the same loop, two workgroups per CU (256 x 128 tiles): knob g4 bit 2
        return dtype == 0 ? launch_g4b_t<__bf16, 1>(p, s) : launch_g4b_t<_Float16, 1>(p, s);
    if (knob(KNOB_CONV_RING) && ((M + 255) / 256) * (2 * I / 256) >= 512) {
        const int v = knob(KNOB_CONV_RING) == 5 ? 1 : (knob(KNOB_CONV_RING) == 7 ? 6 : knob(KNOB_CONV_RING));      // (5 / 7: the ring kernel for the convolutions too)
        if (dtype == 0) return launch_ring_t<__bf16, 4, 1, true>(p, s, v);
        if (dtype == 1) return launch_ring_t<_Float16, 4, 1, true>(p, s, v);
    }
    if (dtype == 0) return launch_conv_t<__bf16, 4, 2, 2, 4, 1>(p, s);
    if (dtype == 1) return launch_conv_t<_Float16, 4, 2, 2, 4, 1>(p, s);
    im360_set_error("linear_geglu: dtype %d unsupported", dtype);
    return IM360_ERR_UNSUPPORTED;
}

// 1x1 convolution of the channel concatenation [xa | xb] that is never materialised (the skip connections of the decoder:
// src/models/MVGenModel.py:407, 415, 431, 437 concatenate, animatediff/models/resnet.py:248-251 runs conv_shortcut on the
// result): xa [N, H, W, C1], xb [N, H, W, C2], w_packed [CoutPad][1][C1 + C2] -> y [N, H, W, Cout] (+ bias, + res).
extern "C" __attribute__((visibility("default"))) int im360_conv1x1_cat_fwd(const void* xa, const void* xb, const void* w_packed, const void* bias, const void* res,
                                     void* y, int64_t N, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t Cout,
                                     int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(xa && xb && w_packed && y, "conv1x1_cat_fwd: null pointer");
    IM360_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0, "conv1x1_cat_fwd: empty problem");
    IM360_CHECK_ARG(C1 > 0 && C2 > 0 && (C1 % 64) == 0 && (C2 % 64) == 0, "conv1x1_cat_fwd: C1=%ld, C2=%ld must be positive multiples of 64", (long)C1, (long)C2);
    IM360_CHECK_ARG(H <= 0xffff && W <= 0xffff && N <= 0x7fffffffL, "conv1x1_cat_fwd: H, W must fit 16 bits");
    IM360_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                    ((uintptr_t)res % 16) == 0 && ((uintptr_t)bias % 8) == 0, "conv1x1_cat_fwd: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "conv1x1_cat_fwd: dtype %d unsupported", dtype);
    ConvParams p = conv_params_zero();
    p.x = xa; p.x2 = xb; p.Cin1 = (int)C1; p.w = w_packed; p.bias = bias; p.res = res; p.y = y;
    p.N = (int)N; p.Hin = (int)H; p.Win = (int)W; p.Cin = (int)(C1 + C2);
    p.Hout = (int)H; p.Wout = (int)W; p.Cout = (int)Cout; p.ntaps = 1;
    p.stride = 1; p.imgs_per_temb = 1;
    p.M = N * H * W;
    ProfScope prof(PROF_CONV, stream);
    return dtype == 0 ? launch_conv<__bf16>(p, (hipStream_t)stream) : launch_conv<_Float16>(p, (hipStream_t)stream);
}

// Token-major Linear y[M, N] = x[M, K] w^T + bias (+ res) on the persistent ring kernel (N % 320 == 0, K % 32 == 0), with
// optional per-row statistics of the stored output: rowstats [M][N / 160][2] fp32 = (sum, sum of squares) of each
// 160-column slice -- what a following LayerNorm needs, so that the consumer can fold the normalisation into its GEMM
// (im360_linear_ln_fwd / im360_linear_geglu_ln) and the LayerNorm pass over the activations disappears.
// Replaces: nn.Linear + residual add feeding nn.LayerNorm, animatediff/models/attention.py:461-508, motion_module.py:230-258.
extern "C" __attribute__((visibility("default"))) int im360_linear_fwd(const void* x, const void* w_packed, const void* bias, const void* res, void* y,
                                void* rowstats, int64_t M, int64_t K, int64_t N, int dtype, void* stream, void* gn_partial) {
    using namespace im360;
    IM360_CHECK_ARG(x && w_packed && y, "linear_fwd: null pointer");
    IM360_CHECK_ARG(M > 0 && M <= 0x7fffffffL && K > 0 && (K % 32) == 0, "linear_fwd: K=%ld must be a positive multiple of 32", (long)K);
    IM360_CHECK_ARG(N > 0 && (N % 320) == 0, "linear_fwd: N=%ld must be a positive multiple of 320", (long)N);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                    ((uintptr_t)res % 16) == 0 && ((uintptr_t)bias % 8) == 0 && ((uintptr_t)rowstats % 8) == 0, "linear_fwd: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "linear_fwd: dtype %d unsupported", dtype);
    ConvParams p = conv_params_zero();
    p.x = x; p.w = w_packed; p.bias = bias; p.res = res; p.y = y; p.rs_out = (float*)rowstats;
    p.N = (int)M; p.Hin = 1; p.Win = 1; p.Cin = (int)K; p.Hout = 1; p.Wout = 1; p.Cout = (int)N; p.ntaps = 1;
    p.stride = 1; p.imgs_per_temb = 1;
    p.M = M;
    IM360_CHECK_ARG(!gn_partial || ((M % 256) == 0 && (K % 64) == 0), "linear_fwd: GroupNorm statistics need M %% 256 == 0 and K %% 64 == 0");
    p.gn_out = (float*)gn_partial;          // (per 256-row tile: the caller's images are whole numbers of tiles)
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, stream);
    const int kr = knob(KNOB_CONV_RING);
    const int v = kr == 5 ? 1 : (kr == 7 ? 6 : kr);
    if (rowstats) return dtype == 0 ? launch_ring_t<__bf16, 5, 5, true>(p, s, (v == 8 || v == 10 || v == 11) ? v : 1) : launch_ring_t<_Float16, 5, 5, true>(p, s, (v == 8 || v == 10 || v == 11) ? v : 1);
    return dtype == 0 ? launch_ring_t<__bf16, 5, 2, true>(p, s, v) : launch_ring_t<_Float16, 5, 2, true>(p, s, v);
}

// LayerNorm folded into the consuming Linear: x are the RAW rows, w_packed = pack(gamma (.) W), and with the rows'
// statistics from the producer (rowstats [M][rs_p][2], im360_linear_fwd)
//   y[r] = rstd_r * (x[r] w^T - mu_r * c1) + (tab ? tab[(r / tab_div) % tab_mod] : c2)
// c1[n] = sum_k w'[n][k] (of the ROUNDED 16-bit w'), c2 = W beta + bias, tab (optional, fp32 [tab_mod][N]) = c2 + e.g. the
// motion module's frame positional encoding pushed through the projection (ABI version 2: the table rows INCLUDE c2).  All fp32 vectors.  N % 320 == 0, K % 32 == 0.
// Replaces: nn.LayerNorm -> nn.Linear (to_q / fused qkv), animatediff/models/attention.py:470-488, motion_module.py:236-250.
extern "C" __attribute__((visibility("default"))) int im360_linear_ln_fwd(const void* x, const void* w_packed, const void* c1, const void* c2, const void* rowstats,
                                   int64_t rs_p, float eps, const void* tab, int64_t tab_div, int64_t tab_mod, void* y,
                                   int64_t M, int64_t K, int64_t N, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && w_packed && c1 && c2 && rowstats && y, "linear_ln_fwd: null pointer");
    IM360_CHECK_ARG(M > 0 && M <= 0x7fffffffL && K > 0 && (K % 32) == 0, "linear_ln_fwd: K=%ld must be a positive multiple of 32", (long)K);
    IM360_CHECK_ARG(N > 0 && (N % 320) == 0, "linear_ln_fwd: N=%ld must be a positive multiple of 320", (long)N);
    IM360_CHECK_ARG(rs_p > 0 && rs_p <= 64, "linear_ln_fwd: rs_p=%ld out of range", (long)rs_p);
    IM360_CHECK_ARG(!tab || (tab_div > 0 && tab_mod > 0 && (tab_div % 256) == 0), "linear_ln_fwd: tab_div=%ld must be a positive multiple of 256 (one table row per 256-row tile), tab_mod positive", (long)tab_div);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)c1 % 16) == 0 &&
                    ((uintptr_t)c2 % 16) == 0 && ((uintptr_t)tab % 16) == 0 && ((uintptr_t)rowstats % 8) == 0, "linear_ln_fwd: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "linear_ln_fwd: dtype %d unsupported", dtype);
    ConvParams p = conv_params_zero();
    p.x = x; p.w = w_packed; p.y = y;
    p.rs_in = (const float*)rowstats; p.rs_p = (int)rs_p; p.ln_eps = eps; p.ln_invc = 1.0f / (float)K;
    p.ln_c1 = (const float*)c1; p.ln_c2 = (const float*)c2; p.ln_tab = (const float*)tab;
    p.tab_div = tab ? (int)tab_div : 1; p.tab_mod = tab ? (int)tab_mod : 1;
    p.N = (int)M; p.Hin = 1; p.Win = 1; p.Cin = (int)K; p.Hout = 1; p.Wout = 1; p.Cout = (int)N; p.ntaps = 1;
    p.stride = 1; p.imgs_per_temb = 1;
    p.M = M;
    ProfScope prof(PROF_GEMM, stream);
    const int v6 = (knob(KNOB_CONV_RING) == 8 || knob(KNOB_CONV_RING) == 10 || knob(KNOB_CONV_RING) == 11) ? knob(KNOB_CONV_RING) : 1;
    return dtype == 0 ? launch_ring_t<__bf16, 5, 3, true>(p, (hipStream_t)stream, v6) : launch_ring_t<_Float16, 5, 3, true>(p, (hipStream_t)stream, v6);
}

// LayerNorm folded into the fused GEGLU projection (im360_linear_geglu with w_packed = pack_geglu(gamma (.) W) and the
// fp32 vectors c1, c2 in the same interleaved row order): out = v * gelu(g), (v | g) = rstd * (x w^T - mu c1) + c2.
// Replaces: nn.LayerNorm -> GEGLU, animatediff/models/attention.py:503-506, motion_module.py:255-257.
extern "C" __attribute__((visibility("default"))) int im360_linear_geglu_ln(const void* x, const void* w_packed, const void* c1, const void* c2, const void* rowstats,
                                     int64_t rs_p, float eps, void* y, int64_t M, int64_t K, int64_t I, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(x && w_packed && c1 && c2 && rowstats && y, "linear_geglu_ln: null pointer");
    IM360_CHECK_ARG(M > 0 && M <= 0x7fffffffL && K > 0 && (K % 32) == 0, "linear_geglu_ln: K=%ld must be a positive multiple of 32", (long)K);
    IM360_CHECK_ARG(I > 0 && (I % 128) == 0, "linear_geglu_ln: I=%ld must be a positive multiple of 128", (long)I);
    IM360_CHECK_ARG(rs_p > 0 && rs_p <= 64, "linear_geglu_ln: rs_p=%ld out of range", (long)rs_p);
    IM360_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)c1 % 16) == 0 &&
                    ((uintptr_t)c2 % 16) == 0 && ((uintptr_t)rowstats % 8) == 0, "linear_geglu_ln: misaligned pointer");
    IM360_CHECK_ARG(dtype == 0 || dtype == 1, "linear_geglu_ln: dtype %d unsupported", dtype);
    ConvParams p = conv_params_zero();
    p.x = x; p.w = w_packed; p.y = y;
    p.rs_in = (const float*)rowstats; p.rs_p = (int)rs_p; p.ln_eps = eps; p.ln_invc = 1.0f / (float)K;
    p.ln_c1 = (const float*)c1; p.ln_c2 = (const float*)c2;
    p.N = (int)M; p.Hin = 1; p.Win = 1; p.Cin = (int)K; p.Hout = 1; p.Wout = 1; p.Cout = (int)(2 * I); p.ntaps = 1;
    p.stride = 1; p.imgs_per_temb = 1;
    p.M = M;
    ProfScope prof(PROF_GEMM, stream);
    if (((knob(KNOB_G4) & 2) || knob(KNOB_CONV_RING) == 12) && (K % 64) == 0 && K >= 128 && (M % 256) == 0 && ((2 * I) % 256) == 0 && (M / 256) * (2 * I / 256) >= 256)       // knob g4 bit 1
        return dtype == 0 ? launch_g4_t<__bf16, 4>(p, (hipStream_t)stream) : launch_g4_t<_Float16, 4>(p, (hipStream_t)stream);
    if (((knob(KNOB_G4) & 8) || knob(KNOB_CONV_RING) == 13) && (K % 64) == 0 && K >= 128 && (M % 256) == 0 && (M / 256) * (2 * I / 128) >= 512)       // knob g4 bit 3
        return dtype == 0 ? launch_g4b_t<__bf16, 4>(p, (hipStream_t)stream) : launch_g4b_t<_Float16, 4>(p, (hipStream_t)stream);
    const int v6 = (knob(KNOB_CONV_RING) == 8 || knob(KNOB_CONV_RING) == 10 || knob(KNOB_CONV_RING) == 11) ? knob(KNOB_CONV_RING) : 1;
    return dtype == 0 ? launch_ring_t<__bf16, 4, 4, true>(p, (hipStream_t)stream, v6) : launch_ring_t<_Float16, 4, 4, true>(p, (hipStream_t)stream, v6);
}

extern "C" __attribute__((visibility("default"))) int im360_pack_conv_weight(const void* w, void* out, int64_t Cout, int64_t Cin, int64_t taps,
                                      int64_t CoutPad, int64_t CinPad, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(w && out, "pack_conv_weight: null pointer");
    IM360_CHECK_ARG(CoutPad >= Cout && CinPad >= Cin && (CoutPad % 128) == 0 && (CinPad % 32) == 0,
                    "pack_conv_weight: CoutPad %% 128 and CinPad %% 32 required");
    const long total = CoutPad * taps * CinPad;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((pack_conv_weight_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)w, (__bf16*)out,
                           (int)Cout, (int)Cin, (int)taps, (int)CoutPad, (int)CinPad);
    else if (dtype == 1)
        hipLaunchKernelGGL((pack_conv_weight_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, (const _Float16*)w,
                           (_Float16*)out, (int)Cout, (int)Cin, (int)taps, (int)CoutPad, (int)CinPad);
    else {
        im360_set_error("pack_conv_weight: dtype %d unsupported", dtype);
        return IM360_ERR_UNSUPPORTED;
    }
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}
