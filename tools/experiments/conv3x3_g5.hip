// The four-wave, register-staged loop of conv3x3_g4.hip on the 256 x 320 tile: 3 x 3 convolutions (taps innermost) and deep-K token-major
// Linears with Cout % 320 == 0.  Included by conv3x3.hip behind conv3x3_g4.hip (sees its g4:: helpers).  Round 6.
//
// Replaces the call sites of conv_igemm_kernel<..., CM = true>: ResnetBlock3D conv1 / conv2 and the other InflatedConv3d 3 x 3
// convolutions without wrap / upsample addressing (animatediff/models/resnet.py:19-27, 183-251).
//
// Differences from the 256 x 256 kernel:
//   * wave = 128 pixels x 160 couts = 4 x 5 accumulators: 16 in the AGPRs, the fifth cout block's 4 in VGPRs (256 + 64 registers);
//   * ONE register set of a stage (18 pieces = 72 registers; two do not fit beside 72 fragment registers): the request of piece i of
//     stage g + 2 follows the ds_write of piece i of stage g + 1 -- a stage time to arrive, what the two LDS-DMA buffers of
//     conv_igemm_kernel give it too; the difference is the loop around it (fragment reads one chunk ahead at one wave per SIMD, one
//     barrier per stage, the stream's instructions one per MFMA slot);
//   * convolution operands: K runs over (64-channel chunk, tap) like conv_igemm_kernel's CM order (identical bits); the pixel rows of
//     a tap are the tile's rows shifted by a wave-uniform offset, read with buffer_load through a descriptor whose base is the
//     tile's first row minus one image row and pixel; a (pixel, tap) outside the image gets the lane offset 0x80000000 >=
//     num_records and the load returns zeros -- no zero chunk, no per-lane pointers;
//   * tile boundaries: the last stage requests nothing, the epilogue runs with an empty queue through the W panel of the buffer that
//     stage no longer reads (40 KB = 4 waves x 32 rows x 640 bytes), the next tile's first stage requests AND writes its second
//     stage (one exposed memory latency per tile of 45 - 360 stages).
#pragma once

namespace g4 {
template <typename T, bool VG> __device__ __forceinline__ void mfma_acc_av(f32x16& d, const u32x4& a, const u32x4& b) {
    if constexpr (VG) {
        if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    } else {
        mfma_acc<T>(d, a, b);
    }
}
// 16 bytes per lane through a buffer descriptor: address = base + soff + voff; voff >= num_records returns zeros
__device__ __forceinline__ void bload128(u32x4& dst, uint32_t voff, const u32x4& srd, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
}  // namespace g4

// CONV: 3 x 3 convolution (stride 1, same size, no wrap / upsample / offsets) with EPI 0; else token-major Linear with EPI 2.
template <typename T, bool CONV, bool NPH_ODD, bool GNS, int RESM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_g5_kernel(ConvParams p) {
    constexpr int NT = 256, WN = 2, TM = 4, TN = 5;
    constexpr int EPI = CONV ? 0 : 2;
    constexpr int BM = 256, BN = 320, BK = 64;
    constexpr int ROWB = BK * 2;
    constexpr int PANEL_A = BM * ROWB, PANEL_W = BN * ROWB;   // 32 KB, 40 KB; LDS = [A0 | A1 | W0 | W1]
    constexpr int NPA = 8, NPW = 10, NP = NPA + NPW;          // 16-byte pieces per lane and stage
    constexpr int LDS_BYTES = 2 * PANEL_A + 2 * PANEL_W;
    static_assert(LDS_BYTES <= 160 * 1024 && (NT / 64) * 32 * TN * 64 <= PANEL_W, "LDS budget / epilogue staging in one W panel");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid_s / WN, wn = wid_s % WN;
    const char* xg = (const char*)p.x;
    const char* wg = (const char*)p.w;
    const int Cb = p.Cin * 2;                                 // bytes per pixel / token row
    const int Kb = p.ntaps * Cb;                              // bytes per weight row
    const int nchunk = p.Cin / BK;
    const int nph = p.ntaps * nchunk;                         // stages per tile (>= 2)

    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / p.ngroups, tn_g = p.tiles_n / p.ngroups;
    const int grp = (blockIdx.x % 8) / xg_n;
    const int ntiles = (int)(p.nblocks / p.ngroups);
    const int tile_first = ((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const int tile_step = xg_n * per_xcd;
    if (tile_first >= ntiles) return;
    const int my_tiles = (ntiles - tile_first + tile_step - 1) / tile_step;
    auto tile_m0 = [&](int j) __attribute__((always_inline)) { return (long)((uint32_t)j / (uint32_t)tn_g) * BM; };
    auto tile_n0 = [&](int j) __attribute__((always_inline)) { return (grp * tn_g + (int)((uint32_t)j % (uint32_t)tn_g)) * BN; };

    const int r0 = tid >> 3, c8 = tid & 7;
    const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t wr_off = (uint32_t)(r0 * ROWB + ((c8 ^ ((r0 >> 1) & 7)) << 4));
    const uint32_t wrA = lds_u32 + wr_off;
    const uint32_t wrW[2] = {lds_u32 + 2 * PANEL_A + wr_off, lds_u32 + 2 * PANEL_A + PANEL_W + wr_off};
    const uint32_t voffA = (uint32_t)(r0 * Cb + c8 * 16), voffW = (uint32_t)(r0 * Kb + c8 * 16);
    const int swz = (col >> 1) & 7;
    uint32_t rdX[4], rdW[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t ko = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
        rdX[ks] = lds_u32 + (wm * 128 + col) * ROWB + ko;
        rdW[ks] = lds_u32 + 2 * PANEL_A + (wn * 160 + col) * ROWB + ko;
    }

    // ---- producer: stage (tile, chunk kc, tap) in conv_igemm_kernel's chunk-major order; saturates on the last stage of the last tile
    int ptile = tile_first, ptap = 0, pkc = 0, pleft = my_tiles;
    const int Wimg = p.Wout;
    const uint32_t back = CONV ? (uint32_t)((Wimg + 1) * Cb) : 0u;        // the descriptor's base sits one image row + one pixel in front of the tile
    uint32_t srd_lo = 0, srd_hi = 0;                                    // (CONV) descriptor base of the producer's tile (the quad is put together at the load: a u32x4 variable
                                                                        //  assigned in a lambda stayed in scratch, and hipcc handed its VGPR reload to the asm's "s" operand)
    const char* pA = xg;                                                // (Linear) scalar base of the producer's stage
    const char* pW = wg;
    uint32_t soffA = 0;                                                 // (CONV) tap / chunk offset behind the descriptor's base
    uint32_t vm[CONV ? 3 : 1];                                          // (CONV) nine validity bits per piece of the producer's tile, three pieces per register
    auto producer_tile = [&]() __attribute__((always_inline)) {
        const long m0 = tile_m0(ptile);
        pW = wg + (long)tile_n0(ptile) * Kb;
        if constexpr (CONV) {
            const uint64_t base = (uint64_t)(uintptr_t)(xg + m0 * Cb) - back;
            srd_lo = (uint32_t)base;
            srd_hi = (uint32_t)(base >> 32) & 0xffffu;
            soffA = back - (uint32_t)((Wimg + 1) * Cb);                  // tap 0 = (-1, -1), chunk 0
            const uint32_t hw = (uint32_t)(p.Hout * p.Wout);
            vm[0] = vm[1] = vm[2] = 0;
            static_for<NPA>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const uint32_t m = (uint32_t)m0 + r0 + 32 * i;          // (pixel indices fit 32 bits: launcher)
                const uint32_t rem = m % hw;
                const int y = (int)(rem / (uint32_t)Wimg), x = (int)(rem % (uint32_t)Wimg);
                uint32_t bits = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int gy = y + tp / 3 - 1, gx = x + tp % 3 - 1;
                    bits |= (gy >= 0 && gy < p.Hout && gx >= 0 && gx < Wimg) ? (1u << tp) : 0u;
                }
                vm[i / 3] |= bits << (9 * (i % 3));
            });
        } else {
            pA = xg + m0 * Cb;
        }
    };
    producer_tile();
    auto producer_advance = [&]() __attribute__((always_inline)) {
        if constexpr (CONV) {
            if (ptap + 1 < 9) {
                ++ptap;
                soffA += (uint32_t)((ptap % 3 == 0 ? Wimg - 2 : 1) * Cb);
                pW += Cb;
            } else if (pkc + 1 < nchunk) {
                ptap = 0;
                ++pkc;
                soffA = back - (uint32_t)((Wimg + 1) * Cb) + (uint32_t)(pkc * ROWB);
                pW += ROWB - 8 * Cb;
            } else if (pleft > 1) {
                --pleft;
                ptap = 0;
                pkc = 0;
                ptile += tile_step;
                producer_tile();
            }
        } else {
            if (pkc + 1 < nchunk) {
                ++pkc;
                pA += ROWB;
                pW += ROWB;
            } else if (pleft > 1) {
                --pleft;
                pkc = 0;
                ptile += tile_step;
                producer_tile();
            }
        }
    };
    u32x4 S[NP];
    // the pieces of a stage are requested in ascending order: their row offsets (32 rows apart) are running VGPR offsets -- per-piece
    // scalar bases cost 32 SGPRs and pushed the kernel over the scalar register file
    const uint32_t stepA = (uint32_t)(32 * Cb), stepW = (uint32_t)(32 * Kb);
    uint32_t runA = voffA, runW = voffW;
    auto load_piece = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < NPA) {
            if constexpr (i == 0) runA = voffA;
            if constexpr (CONV) {
                // the lane's offset, or 0x80000000 (out of range: zeros) when the tap leaves the image at this pixel
                const uint32_t ok = (vm[i / 3] >> (9 * (i % 3) + ptap)) & 1u;
                const u32x4 srd = {srd_lo, srd_hi, 0x80000000u, 0x00020000u};
                g4::bload128(S[i], runA | ((ok ^ 1u) << 31), srd, soffA);
            } else {
                g4::gload128(S[i], runA, pA);
            }
            runA += stepA;
        } else {
            if constexpr (i == NPA) runW = voffW;
            g4::gload128(S[i], runW, pW);
            runW += stepW;
        }
    };
    auto write_piece = [&](auto bc, auto ic) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, i = decltype(ic)::value;
        if constexpr (i < NPA) g4::lds_write128<b * PANEL_A + i * 4096>(wrA, S[i]);
        else g4::lds_write128<(i - NPA) * 4096>(wrW[b], S[i]);
    };
    u32x4 F[2][9];                                            // [set][0..3] pixel blocks, [4..8] cout blocks
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            if (a < 4) zero_acc_mfma<T, false>(acc[a][b]);
            else zero_acc_mfma<T, true>(acc[a][b]);
        }
    auto read_frag = [&](auto bc, auto ksc, auto fsc, auto jc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, ks = decltype(ksc)::value, fs = decltype(fsc)::value, j = decltype(jc)::value;
        if constexpr (j < 4) g4::lds_read128<b * PANEL_A + j * 4096>(F[fs][j], rdX[ks]);
        else g4::lds_read128<b * PANEL_W + (j - 4) * 4096>(F[fs][j], rdW[ks]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // prologue: stage 0 in buffer 0, nothing in flight
    static_for<NP>([&](auto ic) { load_piece(ic); });
    producer_advance();
    g4::wait_vm<0>();
    static_for<NP>([&](auto ic) { write_piece(I0{}, ic); });
    g4::wait_lgkm<0>();
    asm volatile("s_barrier" ::: "memory");
    static_for<9>([&](auto jc) { read_frag(I0{}, I0{}, I0{}, jc); });
    g4::wait_lgkm<0>();

    // ---- one stage g (buffer BUF): four chunks of 20 MFMAs; behind MFMA j < 9 the next chunk's fragment j; the stream takes the 44
    //      slots behind MFMAs 9 .. 19, slot s = 11 q + j - 9.  MID, slots 0 .. 32 (in front of the barrier): W0 W1 W2 L0 W3 L1 ... W17
    //      L14 (W = write piece i of stage g + 1, L = request piece i of stage g + 2 into the register W just emptied), slots 33 - 35:
    //      L15 L16 L17.  LAST: the W slots only.  FIRST (nothing was requested): L0 .. L17 of stage g + 1 in slots 0 - 17, its W0 ..
    //      W17 in slots 15 - 32, the requests of stage g + 2 two per slot behind the barrier.
    auto chunk = [&](auto bufc, auto modec, auto qc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, MODE = decltype(modec)::value, q = decltype(qc)::value;
        constexpr int fs = q & 1;
        using BufC = std::integral_constant<int, BUF>;
        using OthC = std::integral_constant<int, BUF ^ 1>;
        static_for<20>([&](auto jc) {
            constexpr int j = decltype(jc)::value, a = j / 4, b = j % 4;
            g4::mfma_acc_av<T, (a == 4)>(acc[a][b], F[fs][4 + a], F[fs][b]);
            if constexpr (j < 9) {
                if constexpr (q < 3) read_frag(BufC{}, std::integral_constant<int, q + 1>{}, std::integral_constant<int, fs ^ 1>{}, jc);
                else read_frag(OthC{}, I0{}, std::integral_constant<int, fs ^ 1>{}, jc);
            } else {
                constexpr int sl = 11 * q + j - 9;            // stream slot 0 .. 43
                if constexpr (MODE == g4::FIRST) {
                    if constexpr (sl < 18) load_piece(std::integral_constant<int, (sl < 18 ? sl : 0)>{});
                    if constexpr (sl == 17) producer_advance();
                    if constexpr (sl >= 15 && sl < 33) {
                        constexpr int i = sl - 15;
                        g4::wait_vm<(i <= 2 ? 15 : 17 - i)>();       // younger than request i: what has been issued behind it so far
                        write_piece(OthC{}, std::integral_constant<int, (sl >= 15 && sl < 33 ? i : 0)>{});
                    }
                    if constexpr (sl >= 33) {
                        constexpr int k = sl - 33;            // 2 2 2 2 2 2 2 1 1 1 1 requests of stage g + 2
                        constexpr int i0 = k < 7 ? 2 * k : 7 + k;
                        load_piece(std::integral_constant<int, (sl >= 33 ? i0 : 0)>{});
                        if constexpr (k < 7) load_piece(std::integral_constant<int, (sl >= 33 && k < 7 ? i0 + 1 : 0)>{});
                        if constexpr (k == 10) producer_advance();
                    }
                } else if constexpr (sl < 33) {
                    constexpr bool isW = sl < 3 || (sl % 2 == 0);                 // W0 W1 W2, then L at odd, W at even slots
                    constexpr int i = sl < 3 ? sl : (isW ? (sl - 4) / 2 + 3 : (sl - 3) / 2);
                    if constexpr (isW) {
                        // younger than its request: the rest of stage g + 1's and, MID, this stage's first max(0, i - 2)
                        g4::wait_vm<(MODE == g4::MID ? 17 - i + (i > 2 ? i - 2 : 0) : 17 - i)>();
                        write_piece(OthC{}, std::integral_constant<int, i>{});
                    } else if constexpr (MODE == g4::MID) {
                        load_piece(std::integral_constant<int, i>{});
                    }
                } else if constexpr (sl < 36 && MODE == g4::MID) {
                    load_piece(std::integral_constant<int, sl - 18>{});           // L15 L16 L17
                    if constexpr (sl == 35) producer_advance();
                }
            }
        });
        // LDS writes issued behind the chunk's fragment reads
        constexpr int lo = 11 * q, hi_ = 11 * q + 11;
        constexpr int NWR = MODE == g4::FIRST ? ((hi_ < 33 ? hi_ : 33) - (lo > 15 ? lo : 15) > 0 ? (hi_ < 33 ? hi_ : 33) - (lo > 15 ? lo : 15) : 0)
                                              : (q == 0 ? 7 : (q == 1 ? 5 : 6));
        if constexpr (q == 2) {
            g4::wait_lgkm<0>();
            asm volatile("s_barrier" ::: "memory");
        } else if constexpr (q == 3) {
            g4::wait_lgkm<0>();
        } else {
            g4::wait_lgkm<NWR>();
        }
    };
    auto stage = [&](auto bufc, auto modec) __attribute__((always_inline)) {
        chunk(bufc, modec, std::integral_constant<int, 0>{});
        chunk(bufc, modec, std::integral_constant<int, 1>{});
        chunk(bufc, modec, std::integral_constant<int, 2>{});
        chunk(bufc, modec, std::integral_constant<int, 3>{});
    };
    using MFirst = std::integral_constant<int, g4::FIRST>;
    using MMid = std::integral_constant<int, g4::MID>;
    using MLast = std::integral_constant<int, g4::LAST>;

    int ctile = tile_first;
    auto tile_end = [&](auto lastbufc) __attribute__((always_inline)) {                      // lastbufc: the buffer of the tile's last stage (free since its barrier)
        constexpr int LB = decltype(lastbufc)::value;
        const long m0 = tile_m0(ctile);
        const int n0 = tile_n0(ctile);
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                       "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]),
                       "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3])
                     :: "memory");
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        const int wm_e = wid_e / WN, wn_e = wid_e % WN;
        tile_epilogue<T, NT, TM, TN, EPI, false, false, GNS, WN, RESM, true>(p, acc, lds + 2 * PANEL_A + LB * PANEL_W, m0, n0, wm_e, wn_e, wid_e, lane_e);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the next tile's first stage writes its second stage into this buffer
        ctile += tile_step;
        if constexpr (LB == 0) static_for<9>([&](auto jc) { read_frag(I1{}, I0{}, I0{}, jc); });
        else static_for<9>([&](auto jc) { read_frag(I0{}, I0{}, I0{}, jc); });
        g4::wait_lgkm<0>();
    };

    if constexpr (!NPH_ODD) {
        const int npair = (nph - 2) / 2;
        for (int t = 0; t < my_tiles; ++t) {
            stage(I0{}, MFirst{});
            for (int j = 0; j < npair; ++j) {
                stage(I1{}, MMid{});
                stage(I0{}, MMid{});
            }
            stage(I1{}, MLast{});
            tile_end(I1{});
        }
    } else {
        const int npair = (nph - 3) / 2;
        for (int t = 0; t < my_tiles; t += 2) {
            stage(I0{}, MFirst{});
            for (int j = 0; j < npair; ++j) {
                stage(I1{}, MMid{});
                stage(I0{}, MMid{});
            }
            stage(I1{}, MMid{});
            stage(I0{}, MLast{});
            tile_end(I0{});
            if (t + 1 >= my_tiles) break;
            stage(I1{}, MFirst{});
            for (int j = 0; j < npair; ++j) {
                stage(I0{}, MMid{});
                stage(I1{}, MMid{});
            }
            stage(I0{}, MMid{});
            stage(I1{}, MLast{});
            tile_end(I1{});
        }
    }
}

// eligibility of a launch for the kernel above (the caller falls back to the eight-wave kernels otherwise)
static inline bool g5_eligible(const ConvParams& p, bool conv) {
    if (p.Cout % 320 != 0 || p.Cin % 64 != 0 || p.M % 256 != 0 || p.M > 0x7fffffffL || p.x2) return false;
    if ((p.M / 256) * (p.Cout / 320) < 512) return false;
    if (conv) return p.ntaps == 9 && p.stride == 1 && !p.up && !p.wrap && p.x_off == 0 && p.y_off == 0 && p.Hin == p.Hout && p.Win == p.Wout &&
                     (long)(p.Wout + 2) * p.Cin * 2 + 256L * p.Cin * 2 + 64 * 1024 < 0x7fffffffL;
    return p.ntaps == 1 && p.Cin >= 128 && !p.temb && !p.rs_out && !p.gn_out;
}

template <typename T, bool CONV>
static int launch_g5_t(ConvParams p, hipStream_t stream) {
    constexpr int BM = 256, BN = 320;
    p.tiles_n = p.Cout / BN;
    p.nblocks = (p.M / BM) * p.tiles_n;
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    const long want = (p.nblocks + 7) / 8 * 8;
    const unsigned grid = (unsigned)(want < ncu ? want : ncu);
    {
        const long wbytes = (long)p.tiles_n * BN * p.ntaps * p.Cin * 2;
        int ng = 1;
        const int force = knob(KNOB_RING_GROUPS);
        if (force > 0) {
            if ((force == 2 || force == 4 || force == 8) && p.tiles_n % force == 0) ng = force;
        } else {
            while (ng < 8 && wbytes / ng > 3400000L && p.tiles_n % (2 * ng) == 0) ng *= 2;
            if (wbytes / ng > 3400000L) ng = 1;
        }
        p.ngroups = grid >= 8u * ng ? ng : 1;
    }
    const bool odd = ((p.ntaps * (p.Cin / 64)) & 1) != 0;
#define IM360_G5_LAUNCH(ODD, G, R) hipLaunchKernelGGL((conv_g5_kernel<T, CONV, ODD, G, R>), dim3(grid), dim3(256), 0, stream, p)
    if constexpr (CONV) {
        if (p.gn_out) {
            if (p.res) { if (odd) IM360_G5_LAUNCH(true, true, 1); else IM360_G5_LAUNCH(false, true, 1); }
            else { if (odd) IM360_G5_LAUNCH(true, true, 2); else IM360_G5_LAUNCH(false, true, 2); }
        } else {
            if (p.res) { if (odd) IM360_G5_LAUNCH(true, false, 1); else IM360_G5_LAUNCH(false, false, 1); }
            else { if (odd) IM360_G5_LAUNCH(true, false, 2); else IM360_G5_LAUNCH(false, false, 2); }
        }
    } else {
        if (p.res) { if (odd) IM360_G5_LAUNCH(true, false, 1); else IM360_G5_LAUNCH(false, false, 1); }
        else { if (odd) IM360_G5_LAUNCH(true, false, 2); else IM360_G5_LAUNCH(false, false, 2); }
    }
#undef IM360_G5_LAUNCH
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}
