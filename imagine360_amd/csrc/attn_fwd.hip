// Flash-attention forward for gfx950 (CDNA4): softmax(Q K^T * scale + bias) V on 16-bit
// operands with fp32 accumulation, head dim D in {32, 64}.
//
// Replaces the reference's xformers / SDPA call sites:
//   spatial self-attention        diffusers/models/attention_processor.py:1195-1371 (d=64)
//   text + IP cross-attention     animatediff/models/attention.py:65-156 (two KV sets, `accumulate`)
//   cross-view WarpAttn attention src/modules/transformer.py:59-74 (d=32, additive bias shared by
//                                 every (batch, head) -- mask[0] broadcast, transformer.py:68-70)
//
// Design (wave64, v_mfma_f32_32x32x16).  The loop is VALU-issue bound at d = 64 (PMC: ~16 VALU instructions per
// MFMA before this layout), so everything below is about instructions per score, not about the matrix pipe:
//   * one wave owns QB blocks of 32 query rows; a workgroup of NW waves shares one (batch, head) and streams K/V in
//     64-key tiles through double-buffered LDS (register prefetch of tile t+2 overlaps compute of tile t, one
//     barrier per tile).
//   * S^T = K Q^T ("swapped QK^T"): every lane holds 16 scores of ONE query column, so the row max / sum are
//     in-lane reductions plus a single lane^32 exchange, and P is already in MFMA B-operand order for O^T += V^T P^T.
//   * Q is pre-multiplied by scale*log2(e) once (in registers) and the first QK^T MFMA of a chain takes a register
//     block holding -running_max as its C operand: the MFMA result IS the exp2 argument -- no per-score fma.
//   * the running max moves only when a tile exceeds it by more than RESCALE_THR (deferred rescale): the usual
//     per-tile O / l rescale pass, and the refresh of the -max block, are rare events behind one uniform branch.
//   * V is staged row-major (plain 16-byte LDS writes, no register transpose); the PV A-operand is gathered with
//     gfx950's transposing LDS read (ds_read_b64_tr_b16), rows padded to 192 B (d = 64) so the 4 key rows one
//     read touches fall into distinct bank windows.  All fragment reads are base-register + immediate offset.
#include "common.h"
#include "attn_params.h"
#include <stdlib.h>
#include <type_traits>

namespace im360 {

constexpr int KVB = 64;            // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 5.0f;   // log2 units: P stays <= 32 between rescales

// BF ("bias fragments"): the bias matrices hold fp16 values already multiplied by log2(e) (im360_attn_pack_bias) and enter
// the scores through the matrix pipe -- two f16 MFMAs per 32 x 32 score block with a 32 x 16 slice of the identity as the
// A operand and 16 bytes of a bias row as the B operand -- instead of 16 unpack + 16 FMA VALU instructions per lane and
// block.  WarpAttn (d = 32) does half the MFMA work of d = 64 per score on the same softmax VALU work and is VALU-bound.
// BL ("bias through LDS", BF with two query blocks; knob attn_hl = 2, off by default): a lane's mask fragment is 16 bytes of
// ITS query row, so a fragment load straight from global memory touches 64 different cache lines per instruction.  With BL
// the wave fetches its 64 rows x 64 bytes of the next 32-key half COALESCED (four lanes per row segment: 16 lines per
// instruction), parks them in its own 4 KiB of LDS (XOR-swizzled, conflict-free both ways) and reads the per-lane fragments
// from there.  Measured on WarpAttn level 1: 1.016 vs 0.958 ms -- the divergent loads are NOT what limits the kernel (the
// hypothesis was one cache-line lookup per cycle on the vector memory path); kept as an A/B variant, identical bits.
// DS ("dot sums", knob attn_ds): the row sums are taken over the ROUNDED weights -- the packed P words the PV product uses -- with
// eight v_dot2c against (1, 1) per 16 scores instead of sixteen v_add, and the two half-waves exchange their maxima through
// v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip whose lgkmcnt wait also drains the prefetched fragments).
// ONE (Nk <= 64: a single K / V tile, levels 2 / 3 of the perspective branch): one LDS buffer instead of two and three waves per
// SIMD -- these launches are thousands of tiny workgroups whose only lever is how many of them a CU holds (43 KB of LDS each
// allowed three).
// ABL (knob attn_dbg, tools/bench_kernels.py attn_ablate; results are garbage): 1 no exp2, 2 no QK^T MFMAs, 4 no PV MFMAs, 8 no K / V staging.
// HG ("head group", WarpAttn on large grids; knob attn_hg): the NW waves of a workgroup take NW different (batch, head) pairs over the
// SAME 32 QB query rows instead of NW row blocks of one pair.  The [Nq, Nk] mask is shared by every (batch, head), so the waves'
// fragment loads could hit the CU's L1 instead of each going to the L2 (tools/warp_bias_probe.py: with the mask fully cached the
// level-1 launches take 0.82 instead of 0.99 ms).  Every wave stages its own pair's K / V tiles (its own LDS region, 64 lanes per
// tile).  MEASURED SLOWER, off by default: 1.21 vs 0.96 ms at level 1 (1.08 vs 0.83 with a cached mask) -- four times the K / V
// staging per query row costs more than the mask traffic it was meant to save, and the mask's share did not shrink (0.13 ms).
// Bit-identical to the two-query-block kernel (test_attention_head_groups_share_the_mask).
// W3 (knob attn_w3, one query block per wave, d = 64): three waves per SIMD (<= 168 registers) instead of two, i.e. three workgroups
// per CU -- for the short sequences of the perspective branch, where a workgroup's prologue is a large share of its life.
template <typename T, int D, int NW, int QB, bool HAS_BIAS, bool DUAL = false, bool BF = false, bool BL = false, bool DS = false, bool ONE = false, int ABL = 0, bool HG = false, bool W3 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 1 ? (ONE ? 2 : 1) : ((ONE || W3) ? 3 : 2), NW == 1 ? (ONE ? 2 : 1) : ((ONE || W3) ? 3 : 2)))) void attn_fwd_kernel(AttnParams p) {      // (one-wave workgroups stage 16 chunks per lane: 190 - 300 registers)
    static_assert(!W3 || (QB == 1 && !DUAL && !BL), "three-waves-per-SIMD variant");
    static_assert(!ONE || (QB == 1 && !DUAL), "single-tile variant");
    static_assert(!HG || (HAS_BIAS && !DUAL && !BL && !ONE), "head groups share a mask");
    constexpr int SNT = HG ? 64 : NW * 64;     // threads that stage one K / V tile together
    static_assert(!DUAL || (QB == 1 && !HAS_BIAS), "the two-set kernel is the plain one-block-per-wave kernel run twice");
    static_assert(!BF || HAS_BIAS, "bias fragments need a bias");
    [[maybe_unused]] constexpr int NT = NW * 64;
    constexpr int KP = D + 8;          // K tile pitch (elements): 16-B slots rotate by an odd count per row
    constexpr int VP = D == 64 ? 96 : 32;   // V tile pitch: 192-B / 64-B rows -> rows r..r+3 hit 4 distinct 64-B bank windows
    constexpr int DC = D / 16;         // k-steps of the QK^T contraction
    constexpr int DV = D / 32;         // 32-wide blocks of the output head dim
    constexpr int CH = KVB * D / 8;    // 16-byte chunks in a K (or V) tile
    constexpr int CLD = CH / SNT;      // chunks per thread
    constexpr int RSTEP = SNT / (D / 8);    // tile rows between a thread's consecutive chunks
    static_assert(CH % SNT == 0 && SNT % (D / 8) == 0, "staging pattern");
    constexpr bool QK_ALL = QB == 1;             // both halves' scores up front (2 * 16 live score registers per block)
    constexpr bool SHARE_K = QB == 1 || D < 64;  // one K fragment read feeds all query blocks
    constexpr int KT = KVB * KP, VT = KVB * VP;  // tile sizes (elements)

    // double-buffered K / V tiles: one barrier per KV tile (the next tile is written while this one is consumed)
    __shared__ __attribute__((aligned(16))) T k_lds_all[(HG ? NW : 1) * (ONE ? 1 : 2) * KT];
    __shared__ __attribute__((aligned(16))) T v_lds_all[(HG ? NW : 1) * (ONE ? 1 : 2) * VT];
    constexpr bool BF_LDS = BF && QB > 1 && BL;
    constexpr int BROWS = 32 * QB;             // mask rows of one wave
    constexpr int BLD = BROWS * 4 / 64;        // 16-byte loads per lane and 32-key half (4 lanes per 64-byte row segment)
    __shared__ __attribute__((aligned(16))) uint4 b_lds[BF_LDS ? NW * BROWS * 4 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    T* const k_lds2 = k_lds_all + (HG ? wid * (2 * KT) : 0);       // this wave's (HG) or the workgroup's tile pair
    T* const v_lds2 = v_lds_all + (HG ? wid * (2 * VT) : 0);
    const int stid = HG ? lane : tid;
    // XCD-aware block order: the dispatcher deals consecutive block ids round-robin to the 8 XCDs (private L2s).
    // Give every XCD one contiguous range of the (batch*head major, q-tile minor) space, so the ~100 workgroups
    // resident on an XCD at any moment share ONE head's K/V (2 MB at 8192 keys) in that XCD's 4 MB L2 instead of
    // each streaming its own from HBM.
    long lb = blockIdx.x;
    {
        const long nb = gridDim.x, qn = nb / 8, rn = nb % 8, xcd = lb % 8, idx = lb / 8;
        lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    }
    // With a shared bias the [Nq, Nk] mask is the larger stream (256 query rows x Nk x 2 B per workgroup against
    // Nk x d x 4 B of K / V: 4x at d = 32): query tile major, (batch, head) minor, so the workgroups resident on an XCD
    // read the same mask rows out of its L2 and K / V come from the Infinity Cache instead of the other way round.
    const long nbh = (long)p.B * p.H;
    const long nbg = nbh / NW;                  // HG: groups of NW pairs (the host checks divisibility)
    const int bh = HG ? (int)(lb % nbg) * NW + wid : (HAS_BIAS ? (int)(lb % nbh) : (int)(lb / p.nqt));
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = HG ? (int)(lb / nbg) * (32 * QB) : (int)(HAS_BIAS ? lb / nbh : lb % p.nqt) * (32 * NW * QB) + wid * (32 * QB);

    const T* qb_ = (const T*)p.q + (long)b * p.q_bs + (long)h * D;
    // the key / value set being processed (DUAL kernels switch to the second set after the first)
    const T* kb_ = (const T*)p.k + (long)(b / p.kv_group) * p.k_bs + (long)h * D;
    const T* vb = (const T*)p.v + (long)(b / p.kv_group) * p.v_bs + (long)h * D;
    long set_k_rs = p.k_rs, set_v_rs = p.v_rs;
    int set_nk = p.Nk;
    const T* bias = (const T*)p.bias;
    const uint32_t* blk = p.bias_blocks;
    // (readfirstlane: the selector and, below, the map words are wave-uniform -- as scalars the block decisions are s_cbranch on SGPR bits;
    //  left to hipcc they were VGPR words spilled to scratch and tested with v_cmp / exec masks every tile)
    if (HAS_BIAS && p.bias_sel != nullptr && __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(p.bias_sel)) != 0) {
        bias = (const T*)p.bias_alt;
        blk = p.bias_blocks_alt;
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 dc + 8 hi .. +7] * scale * log2(e)
    int qrow[QB];
    bool q_valid[QB];
    uint4 qf[QB][DC];
    f32x16 o[QB][DV];
    f32x16 negm[QB];                   // every register = -(running max), in scaled log2 units
    float m_sc[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = q0 + qb * 32 + col;
        q_valid[qb] = qrow[qb] < p.Nq;
        if (!q_valid[qb]) qrow[qb] = p.Nq - 1;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) {
            float f[8];
            unpack8<T>(*(const uint4*)(qb_ + (long)qrow[qb] * p.q_rs + dc * 16 + hi * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= p.scale_log2;
            qf[qb][dc] = pack8<T>(f);
        }
#pragma unroll
        for (int i = 0; i < DV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
        m_sc[qb] = 0.f;                // the first tile always moves it to that tile's max
        l_run[qb] = 0.f;
    }

    // BF: identity slices I[32 keys][16 c .. 16 c + 15] as f16 A operands: lane (row, hi) holds I[row][16 c + 8 hi .. + 7]
    uint4 idA[2];
    if constexpr (BF) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = col - 16 * c - 8 * hi;                  // position of the 1.0 inside this lane's 8 halves, if any
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (t == 2 * j) ? 0x00003C00u : (t == 2 * j + 1) ? 0x3C000000u : 0u;
            idA[c] = uint4{w[0], w[1], w[2], w[3]};
        }
    }

    int ntiles = (set_nk + KVB - 1) / KVB;
    bool ragged = (set_nk % KVB) != 0;          // the last tile is partial: clamp its rows, mask its scores
    u32x4 kreg[CLD], vreg[CLD];

    // per-thread staging slots: thread tid owns (row, 16-byte chunk) slots tid + i * NT of a tile (rows RSTEP apart),
    // so one source pointer per operand advanced by a tile per load is all the address math
    const int srow = stid / (D / 8), sc8 = stid % (D / 8);
    const T* ksrc = kb_ + (long)srow * set_k_rs + sc8 * 8;
    const T* vsrc = vb + (long)srow * set_v_rs + sc8 * 8;
    long k_tile = (long)KVB * set_k_rs, v_tile = (long)KVB * set_v_rs;
    T* const kdst = k_lds2 + srow * KP + sc8 * 8;
    T* const vdst = v_lds2 + srow * VP + sc8 * 8;

    auto load_tile = [&](int t) {
        if constexpr ((ABL & 8) != 0) return;
        // a ragged last tile clamps its rows into range (always-executed loads: a predicated load inside an unrolled
        // loop makes hipcc branch around it and drain vmcnt per load).  Out-of-range keys are masked to -inf in the
        // tile body, so their (finite, duplicated) K/V rows never contribute.
        const bool clamp = ragged && t == ntiles - 1;
        const int kv0 = t * KVB;
        static_for<CLD>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const T* ks = ksrc + (long)(i * RSTEP) * set_k_rs;
            const T* vs = vsrc + (long)(i * RSTEP) * set_v_rs;
            if (clamp) {
                const long rr = min(kv0 + srow + i * RSTEP, set_nk - 1);
                ks = kb_ + rr * set_k_rs + sc8 * 8;
                vs = vb + rr * set_v_rs + sc8 * 8;
            }
            kreg[i] = *(const u32x4*)ks;
            vreg[i] = *(const u32x4*)vs;
        });
        ksrc += k_tile;
        vsrc += v_tile;
    };
    auto store_tile = [&](int buf) {
        if constexpr ((ABL & 8) != 0) return;
        static_for<CLD>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            *(u32x4*)(kdst + buf * KT + i * RSTEP * KP) = kreg[i];
            *(u32x4*)(vdst + buf * VT + i * RSTEP * VP) = vreg[i];
        });
    };

    // fragment read bases (per lane); everything else is an immediate offset
    const int kfrag = col * KP + hi * 8;
    const int l16 = lane & 15, half = (lane >> 4) & 1;
    const int vfrag = (4 * hi + (l16 >> 2)) * VP + 16 * half + 4 * (l16 & 3);

    // BF with two query blocks per wave: the mask fragments of a 32-key half are requested ONE HALF AHEAD into the other of
    // two register buffers (the buffer index is the half's parity, a compile-time constant): requested right in front of
    // the two QK^T MFMAs that precede their use, their L2 latency stalled every half tile (SQ_WAIT_ANY 60 %).
    constexpr bool BF_AHEAD = BF && QB > 1 && !BL;
    uint4 bfh[2][(BF_AHEAD || BF_LDS) ? QB : 1][2];
    u32x4 breg[BF_LDS ? BLD : 1];
    const int brow = lane >> 2, bch = lane & 3;      // BL loader: instruction i covers mask rows 16 i + brow, 16-byte chunk bch
    uint4* const bw_lds = b_lds + (BF_LDS ? wid * BROWS * 4 : 0);
    // Block map (BF_AHEAD = the WarpAttn kernel): WarpAttn's cached masks are shifted so that the ~97 % background is exactly zero
    // (softmax is invariant under a per-row constant; mv_model.py), and one bit per (32-query block, 32-key half) says whether a
    // block holds anything else.  A clear bit skips the block's two fragment loads (64 scattered lines per instruction) and its two
    // bias MFMAs: 4 instead of 6 MFMAs per block.  The map row of the wave's QB blocks is read a word (32 halves = 16 tiles) at a time
    // through the scalar cache; decisions are wave-uniform branches.
    const bool use_blk = BF_AHEAD && blk != nullptr;
    const int qblk = __builtin_amdgcn_readfirstlane(q0 >> 5);
    const int nqblk = (p.Nq + 31) >> 5;
    uint32_t blkw[QB], blkn[QB];           // the words that hold halves 2 t .. 2 t + 1 and (at a word's last tile) half 2 t + 2
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) blkw[qb] = blkn[qb] = 0xffffffffu;
    auto blk_word = [&](int qb, int w) -> uint32_t {
        const int r = qblk + qb < nqblk ? qblk + qb : nqblk - 1;
        return w < p.blocks_rs ? __builtin_nontemporal_load(blk + (long)r * p.blocks_rs + w) : 0u;
    };
    // bit qb of the result: block qb of this wave needs the bias of 32-key half `h` (readfirstlane: the bits are wave-uniform; as a
    // scalar the decisions below are s_cbranch on SGPR bits instead of v_cmp + exec-mask regions)
    auto blk_need = [&](int h, int t) -> unsigned {
        if (!use_blk) return (1u << QB) - 1u;
        unsigned m = 0;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const uint32_t w = (h >> 5) == (t >> 4) ? blkw[qb] : blkn[qb];
            m |= ((w >> (h & 31)) & 1u) << qb;
        }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    };
    auto fetch_half = [&](int key_base, auto bufc, unsigned need = ~0u) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (BF_LDS) {
            // global -> registers, one 32-key half ahead of its use
            const int key0 = min(key_base + bch * 8, set_nk - 8);
#pragma unroll
            for (int i = 0; i < BLD; ++i) {
                const int row = min(q0 + i * 16 + brow, p.Nq - 1);
                breg[i] = *(const u32x4*)(bias + (long)row * p.bias_rs + key0);
            }
        } else if constexpr (BF_AHEAD) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if ((need >> qb) & 1u) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int key0 = min(key_base + 16 * c + 8 * hi, set_nk - 8);
                        bfh[buf][qb][c] = *(const uint4*)(bias + (long)qrow[qb] * p.bias_rs + key0);
                    }
                }
        }
    };

    // one 64-key tile against the wave's QB query blocks.  MASKED is only instantiated for a ragged last tile.
    auto tile_body = [&](int t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const T* kf = k_lds2 + (t & 1) * KT + kfrag;
        const T* vf = v_lds2 + (t & 1) * VT + vfrag;
        const int kv0 = t * KVB;
        unsigned need_h[3] = {~0u, ~0u, ~0u};        // query blocks that need the bias of halves 2 t, 2 t + 1, 2 t + 2 (the last one is fetched ahead)
        if constexpr (BF_AHEAD) {
            if (use_blk) {
                if ((t & 15) == 0) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) blkw[qb] = t == 0 ? blk_word(qb, 0) : blkn[qb];
                }
                if ((t & 15) == 15) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) blkn[qb] = blk_word(qb, (t >> 4) + 1);
                }
                need_h[0] = blk_need(2 * t, t);
                need_h[1] = blk_need(2 * t + 1, t);
                need_h[2] = blk_need(2 * t + 2, t);
            }
        }

        // bias words first (per tile at QB = 1, per half otherwise): their L2 latency hides under the QK^T MFMAs
        uint2 bw[HAS_BIAS && !BF ? QB : 1][2][4];
        uint4 bfm[BF ? QB : 1][2][2];          // BF: [query block][half][16-key chunk]: keys 16 c + 8 hi .. + 7 of the lane's query row
        auto load_bias = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            if constexpr (BF_LDS) {
                // this half's rows (requested one half ago) -> the wave's LDS patch, row r chunk c at slot c ^ ((r >> 1) & 3);
                // then every lane picks the two fragments of its query row in each block; the next half is requested
#pragma unroll
                for (int i = 0; i < BLD; ++i) {
                    const int r = i * 16 + brow;
                    *(u32x4*)&bw_lds[r * 4 + (bch ^ ((r >> 1) & 3))] = breg[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int r = qb * 32 + col;
                        bfh[0][qb][c] = bw_lds[r * 4 + ((2 * c + hi) ^ ((r >> 1) & 3))];
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                fetch_half(kv0 + (kb + 1) * 32, std::integral_constant<int, 0>{});
            } else if constexpr (BF_AHEAD) {
                // this half's fragments are already in bfh[kb]; request the next half (of this tile or the next one)
                fetch_half(kv0 + (kb + 1) * 32, std::integral_constant<int, 1 - kb>{}, need_h[kb + 1]);
            } else if constexpr (BF) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int key0 = min(kv0 + kb * 32 + 16 * c + 8 * hi, set_nk - 8);     // Nk % 8 == 0 (checked on the host)
                        bfm[qb][QK_ALL ? kb : 0][c] = *(const uint4*)(bias + (long)qrow[qb] * p.bias_rs + key0);
                    }
            } else if (HAS_BIAS) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int key0 = min(kv0 + kb * 32 + 8 * g + 4 * hi, set_nk - 4);     // Nk % 4 == 0 (checked on the host)
                        bw[qb][QK_ALL ? kb : 0][g] = *(const uint2*)(bias + (long)qrow[qb] * p.bias_rs + key0);
                    }
            }
        };

        // s[qb][kb][r] = log2-domain score minus the running max, for (query col of block qb, key kv0 + 32 kb + row(r, hi))
        f32x16 s[QB][2];
        // QK^T of half kb for query block qb (or all blocks sharing each K fragment read)
        auto qk = [&](auto kbc, auto qbc, auto allc) {
            constexpr int kb = decltype(kbc)::value, q1 = decltype(qbc)::value;
            constexpr bool all = decltype(allc)::value;
            if constexpr ((ABL & 2) != 0) {
#pragma unroll
                for (int qb = all ? 0 : q1; qb < (all ? QB : q1 + 1); ++qb) s[qb][kb] = negm[qb];
                return;
            }
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const uint4 a = *(const uint4*)(kf + kb * 32 * KP + dc * 16);
#pragma unroll
                for (int qb = all ? 0 : q1; qb < (all ? QB : q1 + 1); ++qb)
                    s[qb][kb] = Elem<T>::mfma32(a, qf[qb][dc], dc == 0 ? negm[qb] : s[qb][kb]);
            }
            if constexpr (BF) {
#pragma unroll
                for (int qb = all ? 0 : q1; qb < (all ? QB : q1 + 1); ++qb)
                    if (!BF_AHEAD || ((need_h[kb] >> qb) & 1u)) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            s[qb][kb] = Elem<_Float16>::mfma32(idA[c], BF_LDS ? bfh[0][BF_LDS ? qb : 0][c] : (BF_AHEAD ? bfh[kb][BF_AHEAD ? qb : 0][c] : bfm[qb][QK_ALL ? kb : 0][c]), s[qb][kb]);
                    }
            }
        };
        // online-softmax update of block qb over the halves [K0, K1), then O^T += V^T P^T for them
        auto softmax_pv = [&](auto qbc, auto k0c, auto k1c) {
            constexpr int qb = decltype(qbc)::value, K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
#pragma unroll
            for (int kb = K0; kb < K1; ++kb) {
                f32x16& sv = s[qb][kb];
                if constexpr (HAS_BIAS && !BF) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 w = bw[qb][QK_ALL ? kb : 0][g];
                        sv[4 * g + 0] = fmaf(unpack_lo<T>(w.x), LOG2E, sv[4 * g + 0]);
                        sv[4 * g + 1] = fmaf(unpack_hi<T>(w.x), LOG2E, sv[4 * g + 1]);
                        sv[4 * g + 2] = fmaf(unpack_lo<T>(w.y), LOG2E, sv[4 * g + 2]);
                        sv[4 * g + 3] = fmaf(unpack_hi<T>(w.y), LOG2E, sv[4 * g + 3]);
                    }
                }
                if (MASKED) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + kb * 32 + mfma32_row(r, hi) >= set_nk) sv[r] = -INFINITY;
                }
            }
            float mloc = s[qb][K0][0];
#pragma unroll
            for (int kb = K0; kb < K1; ++kb)
#pragma unroll
                for (int r = (kb == K0 ? 1 : 0); r < 16; ++r) mloc = fmaxf(mloc, s[qb][kb][r]);
            if constexpr (DS) {
                float ma, mb;
                half_wave_pair(mloc, ma, mb);
                mloc = fmaxf(ma, mb);
            }
            // deferred rescale: the running max only moves when these keys exceed it by more than RESCALE_THR (log2
            // units; P stays <= 2^THR), so the O / l rescale and the refresh of the -max block are rare.  The very
            // first keys always set it.  The vote runs over the HALF-row maxima (a query's 32 scores of a block sit in lanes q and
            // q + 32; any(max(a, b) > thr) == any(a > thr) over the whole wave), so the exchange between the two half-waves -- a
            // ds_bpermute whose lgkmcnt(0) wait also drains the prefetched fragment reads -- is only paid when the max does move.
            const bool first = (t == 0) && (K0 == 0);
            if (first || __any(mloc > RESCALE_THR)) {
                if constexpr (!DS) mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
                const float delta = first ? mloc : fmaxf(mloc, 0.f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_sc[qb] += delta;
                l_run[qb] *= alpha;
#pragma unroll
                for (int i = 0; i < DV; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qb][r] = -m_sc[qb];
#pragma unroll
                for (int kb = K0; kb < 2; ++kb)        // includes a later half whose QK^T already used the old max
                    if (kb < K1 || QK_ALL)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[qb][kb][r] -= delta;
            }
#pragma unroll
            for (int kb = K0; kb < K1; ++kb) {
                // P = exp2(score - max), packed straight into MFMA B-operand order
                float pv[16];
                float lsum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = (ABL & 1) ? s[qb][kb][r] : __builtin_amdgcn_exp2f(s[qb][kb][r]);
                    if constexpr (!DS) lsum += pv[r];
                }
                const uint4 pf[2] = {pack8<T>(pv), pack8<T>(pv + 8)};
                if constexpr (DS) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        lsum = dot2_acc<T>(pf[c].x, Elem<T>::ones2, lsum);
                        lsum = dot2_acc<T>(pf[c].y, Elem<T>::ones2, lsum);
                        lsum = dot2_acc<T>(pf[c].z, Elem<T>::ones2, lsum);
                        lsum = dot2_acc<T>(pf[c].w, Elem<T>::ones2, lsum);
                    }
                }
                l_run[qb] += lsum;
                // O^T += V^T P^T: MFMA c covers the 16 keys 16c + {0..3, 8..11} + 4 hi (the C-fragment's own key
                // order), gathered by two transposing reads; the two 32-channel accumulators alternate
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr ((ABL & 4) != 0) {
                        asm volatile("" ::"v"(pf[c].x), "v"(pf[c].y), "v"(pf[c].z), "v"(pf[c].w));
                        continue;
                    }
#pragma unroll
                    for (int dvb = 0; dvb < DV; ++dvb) {
                        const T* src = vf + (kb * 32 + 16 * c) * VP + dvb * 32;
                        const u32x2 w0 = lds_read_tr16(src), w1 = lds_read_tr16(src + 8 * VP);
                        uint4 a;
                        a.x = w0.x; a.y = w0.y; a.z = w1.x; a.w = w1.y;
                        o[qb][dvb] = Elem<T>::mfma32(a, pf[c], o[qb][dvb]);
                    }
                }
            }
        };

        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if constexpr (QK_ALL) {
            load_bias(I0{});
            load_bias(I1{});
            // QB == 1: both halves' scores first (the two accumulators alternate), one max / rescale decision per tile
#pragma unroll
            for (int dc = 0; dc < DC; ++dc)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint4 a = *(const uint4*)(kf + kb * 32 * KP + dc * 16);
                    s[0][kb] = Elem<T>::mfma32(a, qf[0][dc], dc == 0 ? negm[0] : s[0][kb]);
                }
            if constexpr (BF) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int c = 0; c < 2; ++c) s[0][kb] = Elem<_Float16>::mfma32(idA[c], bfm[0][kb][c], s[0][kb]);
            }
            softmax_pv(I0{}, I0{}, I2{});
        } else {
            static_for<2>([&](auto kbc) {
                constexpr int kb = decltype(kbc)::value;
                load_bias(kbc);
                if constexpr (SHARE_K) qk(kbc, I0{}, std::true_type{});
                static_for<QB>([&](auto qbc) {
                    if constexpr (!SHARE_K) qk(kbc, qbc, std::false_type{});
                    softmax_pv(qbc, kbc, std::integral_constant<int, kb + 1>{});
                });
            });
        }
    };

    auto run_set = [&]() {
        load_tile(0);
        fetch_half(0, std::integral_constant<int, 0>{});
        store_tile(0);
        if (ntiles > 1) load_tile(1);
        // a ragged last tile is peeled out of the loop: with both bodies inside it hipcc gave them different registers and
        // copied the output accumulators and the max block twice per tile (68 moves, a quarter of the loop's VALU work)
        const int nfull = ragged ? ntiles - 1 : ntiles;
        for (int t = 0; t < nfull; ++t) {
            __syncthreads();                 // tile t is visible; every wave is done with tile t-1 (buffer (t+1)&1)
            if (t + 1 < ntiles) store_tile((t + 1) & 1);
            if (t + 2 < ntiles) load_tile(t + 2);      // in flight during the whole compute phase below
            tile_body(t, std::false_type{});
        }
        if (ragged) {
            __syncthreads();
            tile_body(ntiles - 1, std::true_type{});
        }
    };
    run_set();
    f32x16 osum[DUAL ? DV : 1];          // DUAL: out_scale * (normalised result of the first set)
    if constexpr (DUAL) {
        const float l_tot = l_run[0] + __shfl_xor(l_run[0], 32);
        const float inv = p.out_scale / l_tot;
#pragma unroll
        for (int i = 0; i < DV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                osum[i][r] = o[0][i][r] * inv;
                o[0][i][r] = 0.f;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[0][r] = 0.f;
        m_sc[0] = 0.f;
        l_run[0] = 0.f;
        // second key / value set: same queries, its own softmax
        kb_ = (const T*)p.k2 + (long)(b / p.kv_group) * p.k2_bs + (long)h * D;
        vb = (const T*)p.v2 + (long)(b / p.kv_group) * p.v2_bs + (long)h * D;
        set_k_rs = p.k2_rs;
        set_v_rs = p.v2_rs;
        set_nk = p.Nk2;
        ntiles = (set_nk + KVB - 1) / KVB;
        ragged = (set_nk % KVB) != 0;
        ksrc = kb_ + (long)srow * set_k_rs + sc8 * 8;
        vsrc = vb + (long)srow * set_v_rs + sc8 * 8;
        k_tile = (long)KVB * set_k_rs;
        v_tile = (long)KVB * set_v_rs;
        __syncthreads();                     // every wave is done with the first set's last tile
        run_set();
    }

    // ---- epilogue: normalise, optional accumulate, store 4 consecutive channels per (lane, group)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv = (DUAL ? p.out_scale2 : p.out_scale) / l_tot;
        if (q_valid[qb]) {
            T* ob = (T*)p.out + (long)b * p.o_bs + (long)qrow[qb] * p.o_rs + (long)h * D;
#pragma unroll
            for (int dvb = 0; dvb < DV; ++dvb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float f[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = o[qb][dvb][4 * g + j] * inv + (DUAL ? osum[DUAL ? dvb : 0][4 * g + j] : 0.f);
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
}


// ---- text + IP-adapter cross attention with BOTH key / value sets resident in LDS ---------------------------------------
// Replaces the same call site as the DUAL kernel above (animatediff/models/attention.py:113-148: two independent softmaxes
// of one query over the 77 text tokens and the 64 image tokens, summed with the adapter scale), for the shapes the model
// runs: head dim 64, <= 96 + <= 64 keys, one context per video (kv_group = frames).  There the generic flash kernel is
// launch / staging dominated (9 - 13 % of the MFMA peak): every 128-query workgroup stages the head's 36 KB of K / V for
// 16 KB of Q, runs one or two ragged 64-key tiles with a barrier each and the online-softmax machinery those keys never need.
// Here a workgroup stages the K / V of one (video, head) ONCE (K pre-multiplied by scale * log2(e), rows beyond the key
// count zeroed) and its four waves then stream 32-query blocks of all that video's frames against it with no barrier and no
// LDS write in the loop: per block 4 Q loads (the next block's are requested before this one's math), 20 QK^T + 18 PV MFMAs,
// an exact two-pass softmax (the ragged text set is masked by an MFMA C operand of 0 / -inf, the row sums are dot2's of the
// packed P), and the output stored as whole 16-byte pieces (the two half-waves trade half of their 8-byte fragments through
// v_permlane32_swap).  Work is dealt as contiguous ranges of the (pair, frame, query block) sequence, so a workgroup
// re-stages only when its range crosses into the next pair.  What is left is the HBM time of Q and O.
// QDMA (NWV = 12 waves, one workgroup per CU): the register prefetch above keeps ONE 4 KB block per wave in flight (48 KB per CU:
// 3.0 TB/s of Q + O traffic measured, latency-bound) and fetches 32-byte row segments (32 cache lines per instruction).  Here
// each wave owns a two-slot LDS ring (8 KB) that global_load_lds fills two blocks ahead with whole 128-byte rows (8 rows per
// instruction), no registers involved; the lane -> source mapping applies the XOR swizzle ((row >> 1) & 7 on the 16-byte
// chunk index) that makes the fragment-order ds_read_b128 of a block conflict-free.  The asm LDS-DMA is invisible to hipcc's
// vmcnt bookkeeping, so the loop issues the same VMEM sequence every iteration (4 DMA pieces, clamped to the wave's last block,
// then 4 stores) and waits with hand-counted immediates; vmcnt(0) before the wave ends (a DMA landing after the workgroup's
// LDS has been handed to another one would corrupt it).
template <typename T, int NB1, int NCH1, int NB2, bool RAG2, bool WIDE, int NWV = 4, bool QDMA = false>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(RAG2 && !QDMA ? 2 : 3, RAG2 && !QDMA ? 2 : 3))) void xattn_resident_kernel(AttnParams p) {
    static_assert(!QDMA || NWV == 12, "the ring variant is sized for twelve waves");
    constexpr int NTH = NWV * 64;
    constexpr int D = 64, KP = D + 8, VP = 96, DC = D / 16, DV = D / 32;
    constexpr int NKR = 32 * (NB1 + NB2);          // key rows held (set 2 starts at row 32 NB1)
    constexpr int KBYTES = NKR * KP * 2, VBYTES = NKR * VP * 2, QSLOT = 32 * D * 2;       // one query block = 4 KB
    __shared__ __attribute__((aligned(16))) char lds_all[KBYTES + VBYTES + (QDMA ? NWV * 2 * QSLOT : 0)];
    T* const k_lds = (T*)lds_all;
    T* const v_lds = (T*)(lds_all + KBYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const long r0 = p.x_total * (long)blockIdx.x / (long)gridDim.x, r1 = p.x_total * ((long)blockIdx.x + 1) / (long)gridDim.x;

    const int kfrag = col * KP + hi * 8;
    const int l16 = lane & 15, half = (lane >> 4) & 1;
    const int vfrag = (4 * hi + (l16 >> 2)) * VP + 16 * half + 4 * (l16 & 3);

    // C operands that mask the keys past the end of a set: register r of lane (., hi) is key 32 (NB - 1) + row(r, hi)
    f32x16 cmask1, cmask2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        cmask1[r] = 32 * (NB1 - 1) + mfma32_row(r, hi) >= p.Nk ? -INFINITY : 0.f;
        cmask2[r] = RAG2 && 32 * (NB2 - 1) + mfma32_row(r, hi) >= p.Nk2 ? -INFINITY : 0.f;
    }
    static_assert(NCH1 > 2 * (NB1 - 1) && NCH1 <= 2 * NB1, "16-key chunks of the first set");

    for (long j = r0; j < r1;) {
        const long pair = j / p.x_bpp;
        const long seg_end = r1 < (pair + 1) * p.x_bpp ? r1 : (pair + 1) * p.x_bpp;
        const int bk = (int)(pair / p.H), h = (int)(pair % p.H);
        // ---- stage this pair's K (scaled) and V, both sets; rows past a set's keys are zeros
        __syncthreads();                     // every wave is done with the previous pair
        for (int c = tid; c < NKR * (D / 8); c += NTH) {
            const int row = c / (D / 8), c8 = c % (D / 8);
            const bool second = row >= 32 * NB1;
            const int kr = second ? row - 32 * NB1 : row;
            const bool real = kr < (second ? p.Nk2 : p.Nk);
            const int krc = real ? kr : 0;
            const T* ks = second ? (const T*)p.k2 + (long)bk * p.k2_bs + (long)krc * p.k2_rs : (const T*)p.k + (long)bk * p.k_bs + (long)krc * p.k_rs;
            const T* vs = second ? (const T*)p.v2 + (long)bk * p.v2_bs + (long)krc * p.v2_rs : (const T*)p.v + (long)bk * p.v_bs + (long)krc * p.v_rs;
            uint4 kk = *(const uint4*)(ks + h * D + c8 * 8), vv = *(const uint4*)(vs + h * D + c8 * 8);
            float f[8];
            unpack8<T>(kk, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2;
            kk = pack8<T>(f);
            if (!real) kk = vv = uint4{0u, 0u, 0u, 0u};
            *(uint4*)(k_lds + row * KP + c8 * 8) = kk;
            *(uint4*)(v_lds + row * VP + c8 * 8) = vv;
        }
        __syncthreads();

        // ---- the wave's query blocks of this segment: jj, jj + 4, ... as (frame g, 32-row block qblk), advanced without divisions
        const long base = pair * p.x_bpp;
        const int wid_s = __builtin_amdgcn_readfirstlane(wid);
        auto q_src = [&](int g, int qblk, int& qrow_out, int& b_out) {
            b_out = bk * p.kv_group + g;
            const int row = qblk * 32 + col;
            qrow_out = row;
            return (const T*)p.q + (long)b_out * p.q_bs + (long)row * p.q_rs + h * D + hi * 8;      // (Nq % 32 == 0: no ragged block)
        };
        long jj = j + wid_s;
        int g_cur = (int)((jj - base) / p.x_nqb), qblk_cur = (int)((jj - base) % p.x_nqb);
        uint4 qf[DC], qn[DC];
        int qrow = 0, qb_img = 0;
        auto advance = [&](int n) {          // (g_cur, qblk_cur) n blocks on
            qblk_cur += n;
            while (qblk_cur >= p.x_nqb) {
                qblk_cur -= p.x_nqb;
                ++g_cur;
            }
        };
        // ---- QDMA: the wave's ring and its request stream (runs two blocks ahead of the compute position)
        const uint32_t ring = QDMA ? __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_all) + KBYTES + VBYTES + wid_s * (2 * QSLOT) : 0u;
        const char* ring_p = lds_all + KBYTES + VBYTES + wid_s * (2 * QSLOT);
        const int frag_off = col * (D * 2);                          // row of this lane's fragments; + ((2 dc + hi) ^ ((col >> 1) & 7)) * 16
        const int fsw = (col >> 1) & 7;
        int g_req = g_cur, qblk_req = qblk_cur;                      // the block requested next
        long jj_req = jj;
        auto request = [&](int slot) {
            // piece i: lanes -> rows 8 i + lane / 8, LDS chunk position lane % 8 holding data chunk (lane % 8) ^ ((row >> 1) & 7)
            const T* qb = (const T*)p.q + (long)(bk * p.kv_group + g_req) * p.q_bs + (long)(qblk_req * 32) * p.q_rs + h * D;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3);
                const int ch = (lane & 7) ^ ((row >> 1) & 7);
                lds_dma16_asm(qb + (long)row * p.q_rs + ch * 8, ring + slot * QSLOT + i * 1024);
            }
            if (jj_req + NWV < seg_end) {        // (past the wave's last block the same block is requested again: uniform VMEM sequence)
                jj_req += NWV;
                qblk_req += NWV;
                while (qblk_req >= p.x_nqb) {
                    qblk_req -= p.x_nqb;
                    ++g_req;
                }
            }
        };
        int it = 0;
        if constexpr (QDMA) {
            if (jj < seg_end) {
                request(0);
                request(1);
            }
        } else {
            if (jj < seg_end) {
                const T* qs = q_src(g_cur, qblk_cur, qrow, qb_img);
#pragma unroll
                for (int dc = 0; dc < DC; ++dc) qf[dc] = *(const uint4*)(qs + dc * 16);
            }
        }
        for (; jj < seg_end; jj += NWV, ++it) {
            int nrow = 0, nimg = 0;
            if constexpr (QDMA) {
                // block `it` of this wave sits in slot it & 1: queue behind it = [the next block's 4 pieces] (+ 4 stores from the
                // second iteration on; the stores of two iterations back are then required too -- long done)
                if (it == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                const char* sl = ring_p + (it & 1) * QSLOT + frag_off;
#pragma unroll
                for (int dc = 0; dc < DC; ++dc) qf[dc] = *(const uint4*)(sl + (((2 * dc + hi) ^ fsw) << 4));
                qrow = qblk_cur * 32 + col;
                qb_img = bk * p.kv_group + g_cur;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers: the slot can be refilled
                request(it & 1);
                if (jj + NWV < seg_end) advance(NWV);
            } else {
            // the next block's Q (the wave's last block re-requests itself: always-executed loads keep hipcc's waits counted)
            if (jj + NWV < seg_end) advance(NWV);
            const T* qs = q_src(g_cur, qblk_cur, nrow, nimg);
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) qn[dc] = *(const uint4*)(qs + dc * 16);
            __builtin_amdgcn_sched_barrier(0);          // (hipcc sinks the requests to the middle of the block otherwise)
            }

            f32x16 osum[DV];
            float inv2 = 0.f;
            f32x16 o[DV];
            // one key / value set: scores, exact softmax, O^T = V^T P^T; returns the row sum
            auto run = [&](auto nbc, auto nchc, int kbase, const f32x16& cmask) {
                constexpr int NB = decltype(nbc)::value, NCH = decltype(nchc)::value;      // NCH: 16-key chunks holding real keys
                f32x16 s[NB];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
                    for (int dc = 0; dc < DC; ++dc) {
                        const uint4 a = *(const uint4*)(k_lds + kfrag + (kbase + kb * 32) * KP + dc * 16);
                        if (dc == 0) {
                            if (kb == NB - 1) s[kb] = Elem<T>::mfma32(a, qf[dc], cmask);
                            else s[kb] = Elem<T>::mfma32(a, qf[dc], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                        } else {
                            s[kb] = Elem<T>::mfma32(a, qf[dc], s[kb]);
                        }
                    }
                }
                float m = s[0][0];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                    for (int r = (kb == 0 ? 1 : 0); r < 16; ++r) m = fmaxf(m, s[kb][r]);
                {
                    float ma, mb;               // both halves of the wave (keys 4 hi + ...)
                    half_wave_pair(m, ma, mb);
                    m = fmaxf(ma, mb);
                }
                float l = 0.f;
#pragma unroll
                for (int i = 0; i < DV; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
                static_for<NB>([&](auto kbc) {
                    constexpr int kb = decltype(kbc)::value;
                    float pv[16];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 d = f32x2{s[kb][r], s[kb][r + 1]} - f32x2{m, m};        // v_pk_add_f32
                        pv[r] = __builtin_amdgcn_exp2f(d.x);
                        pv[r + 1] = __builtin_amdgcn_exp2f(d.y);
                    }
                    const uint4 pf[2] = {pack8<T>(pv), pack8<T>(pv + 8)};
                    // row sum of the ROUNDED weights (the ones the PV product uses): eight dot2's instead of sixteen adds
                    static_for<2>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr (kb * 2 + c < NCH) {         // (a chunk of nothing but masked keys adds zeros: its exp2's are dead code)
                            l = dot2_acc<T>(pf[c].x, Elem<T>::ones2, l);
                            l = dot2_acc<T>(pf[c].y, Elem<T>::ones2, l);
                            l = dot2_acc<T>(pf[c].z, Elem<T>::ones2, l);
                            l = dot2_acc<T>(pf[c].w, Elem<T>::ones2, l);
                        }
                    });
                    static_for<2>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr (kb * 2 + c < NCH) {         // (a chunk of nothing but padding is skipped)
#pragma unroll
                            for (int dvb = 0; dvb < DV; ++dvb) {
                                const T* src = v_lds + vfrag + (kbase + kb * 32 + 16 * c) * VP + dvb * 32;
                                const u32x2 w0 = lds_read_tr16(src), w1 = lds_read_tr16(src + 8 * VP);
                                uint4 a;
                                a.x = w0.x; a.y = w0.y; a.z = w1.x; a.w = w1.y;
                                o[dvb] = Elem<T>::mfma32(a, pf[c], o[dvb]);
                            }
                        }
                    });
                });
                float la, lb;
                half_wave_pair(l, la, lb);
                return la + lb;
            };
            {
                const float l1 = run(std::integral_constant<int, NB1>{}, std::integral_constant<int, NCH1>{}, 0, cmask1);
                const float inv1 = p.out_scale / l1;
#pragma unroll
                for (int i = 0; i < DV; ++i)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 t = f32x2{o[i][r], o[i][r + 1]} * f32x2{inv1, inv1};
                        osum[i][r] = t.x;
                        osum[i][r + 1] = t.y;
                    }
                const float l2 = run(std::integral_constant<int, NB2>{}, std::integral_constant<int, 2 * NB2>{}, 32 * NB1, cmask2);
                inv2 = p.out_scale2 / l2;
            }

            // ---- out = osum + o * inv2.  Lane (query col, hi) holds channels 32 dvb + 8 g + 4 hi .. + 3 (8 bytes).
            T* ob = (T*)p.out + (long)qb_img * p.o_bs + (long)qrow * p.o_rs + h * D;
#pragma unroll
            for (int dvb = 0; dvb < DV; ++dvb) {
                uint2 w[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 i2 = {inv2, inv2};
                    const f32x2 f01 = f32x2{o[dvb][4 * g], o[dvb][4 * g + 1]} * i2 + f32x2{osum[dvb][4 * g], osum[dvb][4 * g + 1]};
                    const f32x2 f23 = f32x2{o[dvb][4 * g + 2], o[dvb][4 * g + 3]} * i2 + f32x2{osum[dvb][4 * g + 2], osum[dvb][4 * g + 3]};
                    w[g].x = pack2<T>(f01.x, f01.y);
                    w[g].y = pack2<T>(f23.x, f23.y);
                }
                if constexpr (WIDE) {
                    // groups (g, g + 1): the hi = 0 lanes hand their 8 bytes of group g + 1 to the partner lane and receive its
                    // 8 bytes of group g (v_permlane32_swap: rows 2-3 of the first operand <-> rows 0-1 of the second) -- every
                    // lane then owns the 16 contiguous bytes of channels 8 (g + hi) .. + 7
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const u32x2 sx = __builtin_amdgcn_permlane32_swap(w[g].x, w[g + 1].x, false, false);
                        const u32x2 sy = __builtin_amdgcn_permlane32_swap(w[g].y, w[g + 1].y, false, false);
                        const uint4 piece = {sx.x, sy.x, sx.y, sy.y};
                        *(uint4*)(ob + dvb * 32 + 8 * (g + hi)) = piece;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(uint2*)(ob + dvb * 32 + 8 * g + 4 * hi) = w[g];
                }
            }
            if constexpr (!QDMA) {
#pragma unroll
                for (int dc = 0; dc < DC; ++dc) qf[dc] = qn[dc];
                qrow = nrow;
                qb_img = nimg;
            }
        }
        j = seg_end;
    }
    if constexpr (QDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Rejected / ablation variants (knobs attn_dbg, attn_hl, attn_hg, attn_ds, attn_w3 = 2) are compiled only into `make ablate`
// builds (-DIM360_ABLATE; im360_build_flags() bit 0): in the shipped library those knobs read as 0.
#ifdef IM360_ABLATE
#define IM360_ABL_KNOB(k) knob(k)
#else
#define IM360_ABL_KNOB(k) 0
#endif

template <typename T, int D, bool HAS_BIAS>
static int launch_attn_b(AttnParams p, hipStream_t stream) {
    const int nw = p.Nq <= 32 ? 1 : (p.Nq <= 64 ? 2 : 4);
    // Query blocks per wave.  Two blocks share each K fragment read and halve the K/V staging per query.  Measured
    // (tools/bench_kernels.py attn_variants): WarpAttn level 1 (d = 32 + bias, 222 VGPRs, no scratch) 1.62 vs 1.87 ms,
    // panorama level-0 self-attention (d = 64, 48 B of scratch) 2.75 vs 3.10 ms = 1.0 PF/s, perspective level 0 1.09 vs
    // 1.19 ms; a tie on small grids, where one block per wave keeps more workgroups in flight.  Knob attn_qb: 0 = this
    // rule, 1 / 2 = force.
    const int qb_env = knob(KNOB_ATTN_QB);       // tuning override
    int qb = 1;
    if (nw == 4 && (qb_env == 2 || (qb_env == 0 && (long)p.B * p.H * ((p.Nq + 255) / 256) >= 1024))) qb = 2;
    // Round 3: at d = 64 without a bias, ONE block per wave at THREE waves per SIMD (148 registers, no scratch; three 43 KB
    // workgroups per CU) beats the two-block form on every self-attention shape of the step: panorama level 0 2.99 -> 2.95 ms,
    // perspective level 0 1.10 -> 1.07, level 1 0.414 -> 0.397 / 0.218 -> 0.209, panorama level 2 0.072 -> 0.064
    // (bench_kernels.py attn_w3; one block at two waves per SIMD: 3.14 / 1.19).  Knob attn_w3 0 restores the rule above.
    const bool w3 = !HAS_BIAS && D == 64 && nw == 4 && qb_env != 2 && knob(KNOB_ATTN_W3) != 0 && IM360_ABL_KNOB(KNOB_ATTN_DBG) == 0 && !(qb_env == 1 && IM360_ABL_KNOB(KNOB_ATTN_DS));
    if (w3) qb = 1;
    p.nqt = (p.Nq + 32 * nw * qb - 1) / (32 * nw * qb);
    const long nblk = (long)p.B * p.H * p.nqt;
    if (nblk > 0x7fffffffL) {
        im360_set_error("attn_fwd: %ld workgroups exceed the grid limit", nblk);
        return IM360_ERR_ARG;
    }
    dim3 grid((unsigned)nblk, 1, 1);
    if constexpr (HAS_BIAS && D == 32) {
        if (p.bias_packed) {
            if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 1, 1, true, false, true>), grid, dim3(64), 0, stream, p);
            else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2, 1, true, false, true>), grid, dim3(128), 0, stream, p);
#ifdef IM360_ABLATE
            else if (qb == 1 && knob(KNOB_ATTN_W3) == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, true, false, true, false, false, false, 0, false, true>), grid, dim3(256), 0, stream, p);      // A/B (attn_qb 1 + attn_w3 2): WarpAttn with one block per wave at three waves per SIMD -- 1.13 ms against 0.98 for the two-block form (which shares every K fragment and mask prefetch between its blocks) and 1.25 at two waves: off
#endif
            else if (qb == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, true, false, true>), grid, dim3(256), 0, stream, p);
#ifdef IM360_ABLATE
            else if (knob(KNOB_ATTN_HL) == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, true, false, true, true>), grid, dim3(256), 0, stream, p);      // A/B: mask rows through the wave's LDS patch (measured 6 % slower: the divergent fragment loads are not the limiter)
            else if (knob(KNOB_ATTN_HG) && ((long)p.B * p.H) % 4 == 0 && (long)p.B * p.H * ((p.Nq + 63) / 64) / 4 <= 0x7fffffffL) {
                // head groups: four (batch, head) pairs per workgroup over the same 64 query rows
                dim3 hgrid((unsigned)((long)p.B * p.H / 4 * ((p.Nq + 63) / 64)), 1, 1);
                hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, true, false, true, false, false, false, 0, true>), hgrid, dim3(256), 0, stream, p);
            }
            else if (knob(KNOB_ATTN_DS)) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, true, false, true, false, true>), grid, dim3(256), 0, stream, p);
#endif
            else hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, true, false, true>), grid, dim3(256), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (p.bias_packed) {
        im360_set_error("attn_fwd: packed bias matrices are supported for head dim 32 only");
        return IM360_ERR_UNSUPPORTED;
    }
    if constexpr (!HAS_BIAS && D == 64) {
        // software-pipelined kernel (attn_pipe.hip; knob attn_pipe): the self-attention shapes with whole 64-key tiles
        if (nw == 4 && IM360_ABL_KNOB(KNOB_ATTN_DBG) == 0) {
            const int rc = launch_attn_pipe(p, std::is_same<T, __bf16>::value ? 0 : 1, stream);
            if (rc != 1) return rc;
        }
    }
    if constexpr (!HAS_BIAS && D == 64) {
        if (p.Nk <= KVB && qb == 1 && knob(KNOB_ATTN_ONE)) {
            if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 1, 1, false, false, false, false, false, true>), grid, dim3(64), 0, stream, p);
            else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2, 1, false, false, false, false, false, true>), grid, dim3(128), 0, stream, p);
            else hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, false, false, false, false, false, true>), grid, dim3(256), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if constexpr (!HAS_BIAS && D == 64) {
        if (w3 && !(p.Nk <= KVB && knob(KNOB_ATTN_ONE))) {
            hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, false, false, false, false, false, false, 0, false, true>), grid, dim3(256), 0, stream, p);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        }
    }
    if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 1, 1, HAS_BIAS>), grid, dim3(64), 0, stream, p);
    else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2, 1, HAS_BIAS>), grid, dim3(128), 0, stream, p);
#ifdef IM360_ABLATE
    else if (qb == 1 && knob(KNOB_ATTN_DS)) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, HAS_BIAS, false, false, false, true>), grid, dim3(256), 0, stream, p);
#endif
    else if (qb == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1, HAS_BIAS>), grid, dim3(256), 0, stream, p);
#ifdef IM360_ABLATE
    else if (!HAS_BIAS && D == 64 && knob(KNOB_ATTN_DBG)) {
        if constexpr (!HAS_BIAS && D == 64) {
            switch (knob(KNOB_ATTN_DBG)) {
                case 1: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 1>), grid, dim3(256), 0, stream, p); break;
                case 2: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 2>), grid, dim3(256), 0, stream, p); break;
                case 4: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 4>), grid, dim3(256), 0, stream, p); break;
                case 6: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 6>), grid, dim3(256), 0, stream, p); break;
                case 7: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 7>), grid, dim3(256), 0, stream, p); break;
                case 8: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 8>), grid, dim3(256), 0, stream, p); break;
                default: hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, false, false, false, false, false, false, 15>), grid, dim3(256), 0, stream, p); break;
            }
        }
    }
    else if (knob(KNOB_ATTN_DS)) hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, HAS_BIAS, false, false, false, true>), grid, dim3(256), 0, stream, p);
#endif
    else hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 2, HAS_BIAS>), grid, dim3(256), 0, stream, p);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

template <typename T, int D>
static int launch_attn(const AttnParams& p, hipStream_t stream) {
    if (p.k2) {
        if constexpr (D == 64) {
            AttnParams q = p;
            // both sets resident in LDS (knob attn_x: 1 = default, 2 = the same with 8-byte stores, 0 = the generic two-pass kernel)
            const int xk = knob(KNOB_ATTN_X);
            if (xk && q.Nk > 64 && q.Nk <= 96 && q.Nk2 > 32 && q.Nk2 <= 64 && (q.o_rs % 8) == 0 && (q.o_bs % 8) == 0 &&
                ((uintptr_t)q.out % 16) == 0 && q.B % q.kv_group == 0 && q.Nq % 32 == 0) {
                q.x_nqb = (q.Nq + 31) / 32;
                q.x_bpp = q.kv_group * q.x_nqb;
                q.x_total = (long)(q.B / q.kv_group) * q.H * q.x_bpp;
                // the model's shape (77 + 64 keys: five real 16-key chunks, second set unmasked) or the general instance
                const bool model_shape = q.Nk <= 80 && q.Nk2 == 64;
                // knob 3: twelve-wave workgroups (one per CU) whose waves fetch their query blocks through private LDS rings
                // (global_load_lds, two blocks ahead); needs 16-byte aligned query rows
                const bool ring = xk == 3 && model_shape && (q.q_rs % 8) == 0 && (q.q_bs % 8) == 0 && ((uintptr_t)q.q % 16) == 0 && q.x_total >= 12 * 256;
                // up to 48 query blocks (12 per wave) per four-wave workgroup -- the pair's 36 KB of K / V are then staged for 384 KB
                // of Q / O, with enough workgroups left for the dispatcher to level the tail -- and no fewer than 8 on small problems,
                // which would otherwise leave CUs idle (K / V come from the L2 there); the ring variant: 12 blocks per wave likewise
                long per = q.x_total / 1024;
                per = per < 8 ? 8 : (per > 48 ? 48 : per);
                long g = (q.x_total + per - 1) / per;
                if (ring) {
                    // one workgroup per CU: a whole number of rounds of <= 144 blocks (12 per wave) each
                    static const int ncu = [] {
                        int dev = 0, n = 0;
                        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
                        return n > 0 ? n : 256;
                    }();
                    g = (long)ncu * ((q.x_total + 144L * ncu - 1) / (144L * ncu));
                }
                if (g < 1) g = 1;
                if (g > 0x7fffffffL) {
                    im360_set_error("attn_fwd2: %ld workgroups exceed the grid limit", g);
                    return IM360_ERR_ARG;
                }
                dim3 xgrid((unsigned)g, 1, 1);
                if (ring) {
                    hipLaunchKernelGGL((xattn_resident_kernel<T, 3, 5, 2, false, true, 12, true>), xgrid, dim3(768), 0, stream, q);
                } else if (xk == 2) {
                    if (model_shape) hipLaunchKernelGGL((xattn_resident_kernel<T, 3, 5, 2, false, false>), xgrid, dim3(256), 0, stream, q);
                    else hipLaunchKernelGGL((xattn_resident_kernel<T, 3, 6, 2, true, false>), xgrid, dim3(256), 0, stream, q);
                } else {
                    if (model_shape) hipLaunchKernelGGL((xattn_resident_kernel<T, 3, 5, 2, false, true>), xgrid, dim3(256), 0, stream, q);
                    else hipLaunchKernelGGL((xattn_resident_kernel<T, 3, 6, 2, true, true>), xgrid, dim3(256), 0, stream, q);
                }
                IM360_CHECK_LAUNCH();
                return IM360_OK;
            }
            const int nw = q.Nq <= 32 ? 1 : (q.Nq <= 64 ? 2 : 4);
            q.nqt = (q.Nq + 32 * nw - 1) / (32 * nw);
            const long nblk = (long)q.B * q.H * q.nqt;
            if (nblk > 0x7fffffffL) {
                im360_set_error("attn_fwd: %ld workgroups exceed the grid limit", nblk);
                return IM360_ERR_ARG;
            }
            dim3 grid((unsigned)nblk, 1, 1);
            if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, 64, 1, 1, false, true>), grid, dim3(64), 0, stream, q);
            else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, 64, 2, 1, false, true>), grid, dim3(128), 0, stream, q);
            else hipLaunchKernelGGL((attn_fwd_kernel<T, 64, 4, 1, false, true>), grid, dim3(256), 0, stream, q);
            IM360_CHECK_LAUNCH();
            return IM360_OK;
        } else {
            im360_set_error("attn_fwd2: two key/value sets need head dim 64");
            return IM360_ERR_UNSUPPORTED;
        }
    }
    return p.bias ? launch_attn_b<T, D, true>(p, stream) : launch_attn_b<T, D, false>(p, stream);
}

}  // namespace im360

extern "C" __attribute__((visibility("default"))) int im360_attn_fwd(const void* q, const void* k, const void* v, const void* bias, void* out,
                              int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                              int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                              int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, int64_t bias_rs,
                              int64_t kv_group, float scale, float out_scale, int accumulate, int dtype, void* stream,
                              const void* bias_alt, const void* bias_sel,
                              const void* bias_blocks, const void* bias_blocks_alt, int64_t blocks_rs) {
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
    const int bias_packed = (dtype & 0x100) ? 1 : 0;       // dtype + 256: packed fp16 bias matrices (see the header)
    dtype &= 0xff;
    if (bias) {
        IM360_CHECK_ARG((Nk % 4) == 0 && Nk >= 4 && (bias_rs % 4) == 0 && ((uintptr_t)bias % 8) == 0,
                        "attn_fwd: bias needs Nk %% 4 == 0 and 8-byte aligned rows");
        IM360_CHECK_ARG(!bias_packed || ((Nk % 8) == 0 && (bias_rs % 8) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)bias_alt % 16) == 0),
                        "attn_fwd: packed bias needs Nk %% 8 == 0 and 16-byte aligned rows");
    }
    IM360_CHECK_ARG(bias || !bias_packed, "attn_fwd: packed-bias flag without a bias");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.bias = bias; p.out = out;
    p.bias_alt = bias_alt; p.bias_sel = (const int*)bias_sel;
    IM360_CHECK_ARG(!bias_sel || (bias && bias_alt && ((uintptr_t)bias_alt % 8) == 0), "attn_fwd: bias_sel needs bias and an aligned bias_alt");
    IM360_CHECK_ARG(!bias_blocks || (bias_packed && blocks_rs * 32 * 32 >= Nk && ((uintptr_t)bias_blocks % 4) == 0 && (!bias_alt || bias_blocks_alt)),
                    "attn_fwd: a block map needs a packed bias, ceil(Nk / 1024) <= blocks_rs words per row and one map per bias matrix");
    p.bias_blocks = (const uint32_t*)bias_blocks; p.bias_blocks_alt = (const uint32_t*)bias_blocks_alt; p.blocks_rs = (int)blocks_rs;
    p.B = (int)B; p.H = (int)H; p.Nq = (int)Nq; p.Nk = (int)Nk; p.kv_group = (int)kv_group;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs;
    p.o_bs = o_bs; p.o_rs = o_rs; p.bias_rs = bias_rs;
    p.scale_log2 = scale * LOG2E; p.out_scale = out_scale; p.accumulate = accumulate; p.bias_packed = bias_packed;
    p.k2 = nullptr; p.v2 = nullptr; p.Nk2 = 0; p.k2_bs = p.k2_rs = p.v2_bs = p.v2_rs = 0; p.out_scale2 = 0.f;
    p.x_nqb = p.x_bpp = 0; p.x_total = 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(bias ? PROF_ATTN_WARP : PROF_ATTN, stream);
    if (dtype == 0) return D == 64 ? launch_attn<__bf16, 64>(p, s) : launch_attn<__bf16, 32>(p, s);
    if (dtype == 1) return D == 64 ? launch_attn<_Float16, 64>(p, s) : launch_attn<_Float16, 32>(p, s);
    im360_set_error("attn_fwd: dtype %d unsupported (0=bf16, 1=f16)", dtype);
    return IM360_ERR_UNSUPPORTED;
}

// Two key / value sets for the same queries in ONE launch: out = out_scale * softmax(q k^T scale) v + out_scale2 *
// softmax(q k2^T scale) v2 (head dim 64, no bias).  Strides as in im360_attn_fwd.
extern "C" __attribute__((visibility("default"))) int im360_attn_fwd2(const void* q, const void* k, const void* v, const void* k2, const void* v2, void* out,
                               int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Nk2, int64_t D,
                               int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs,
                               int64_t k2_bs, int64_t k2_rs, int64_t v2_bs, int64_t v2_rs, int64_t o_bs, int64_t o_rs,
                               int64_t kv_group, float scale, float out_scale, float out_scale2, int dtype, void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(q && k && v && k2 && v2 && out, "attn_fwd2: null pointer");
    IM360_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0 && Nk2 > 0, "attn_fwd2: empty problem");
    IM360_CHECK_ARG(kv_group >= 1, "attn_fwd2: kv_group must be >= 1");
    IM360_CHECK_ARG(D == 64, "attn_fwd2: head dim %ld unsupported (64)", (long)D);
    IM360_CHECK_ARG(B * H <= 0x7fffffffL, "attn_fwd2: B*H too large");
    IM360_CHECK_ARG((q_rs % 8) == 0 && (k_rs % 8) == 0 && (v_rs % 8) == 0 && (k2_rs % 8) == 0 && (v2_rs % 8) == 0 && (o_rs % 4) == 0 &&
                    (q_bs % 8) == 0 && (k_bs % 8) == 0 && (v_bs % 8) == 0 && (k2_bs % 8) == 0 && (v2_bs % 8) == 0 && (o_bs % 4) == 0,
                    "attn_fwd2: strides must keep 16-byte (q,k,v) / 8-byte (out) alignment");
    IM360_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)k2 % 16) == 0 &&
                    ((uintptr_t)v2 % 16) == 0 && ((uintptr_t)out % 8) == 0, "attn_fwd2: misaligned base pointer");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.bias = nullptr; p.out = out; p.bias_alt = nullptr; p.bias_sel = nullptr;
    p.B = (int)B; p.H = (int)H; p.Nq = (int)Nq; p.Nk = (int)Nk; p.kv_group = (int)kv_group;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs;
    p.o_bs = o_bs; p.o_rs = o_rs; p.bias_rs = 0;
    p.scale_log2 = scale * LOG2E; p.out_scale = out_scale; p.accumulate = 0; p.bias_packed = 0;
    p.k2 = k2; p.v2 = v2; p.Nk2 = (int)Nk2; p.k2_bs = k2_bs; p.k2_rs = k2_rs; p.v2_bs = v2_bs; p.v2_rs = v2_rs; p.out_scale2 = out_scale2;
    p.x_nqb = p.x_bpp = 0; p.x_total = 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN_X2, stream);
    if (dtype == 0) return launch_attn<__bf16, 64>(p, s);
    if (dtype == 1) return launch_attn<_Float16, 64>(p, s);
    im360_set_error("attn_fwd2: dtype %d unsupported (0=bf16, 1=f16)", dtype);
    return IM360_ERR_UNSUPPORTED;
}
