// Four-wave, one-wave-per-SIMD, register-staged GEMM tile for the token-major Linears (included by conv3x3.hip: sees ConvParams,
// tile_epilogue and the helpers).  Round 6.
//
// Replaces the same call sites as conv_ring_kernel<..., LINEAR = true>: the feed-forward / projection nn.Linear layers of
// BasicTransformerBlock and the motion module (animatediff/models/attention.py:304-423, 461-508, motion_module.py:230-258,
// diffusers/models/attention_lora.py:493-547 GEGLU / FeedForward) as y[M, N] = x[M, K] w^T with the epilogues of tile_epilogue.
//
// Why another loop (profiles/HISTORY.md 3d): in the eight-wave kernels the LDS-DMA stream and the MFMAs do not hide each
// other -- the parts of the ablation add up -- and the operand stream has ONE stage time to arrive (two LDS buffers).  The vendor's
// GEMM that beats them on the wide / deep shapes (MT256x256x64, four waves, 512 registers per wave) keeps two K tiles of global
// loads in flight in REGISTERS.  This kernel has that structure, written for gfx950 directly:
//   * 256 threads = four waves, one per SIMD, 512 registers each: 256 x 256 tile, wave = 128 tokens x 128 couts = 4 x 4 blocks of
//     v_mfma_f32_32x32x16, the 256 accumulator registers in AGPRs (asm constraint "a"), 8 instead of 14 fragment reads per 16 / 20
//     MFMAs of the 64 x 160 wave tile;
//   * operands go global -> VGPR (global_load_dwordx4, whole 128-byte lines: 8 lanes per row) -> LDS (ds_write_b128 at a swizzled
//     address) through TWO register sets of one 64-channel stage each: in a tile's middle stages the load of piece i of stage g + 3
//     is issued right behind the ds_write of piece i of stage g + 1, so a load has two stage times (2 x 2048 MFMA cycles) to arrive
//     and 128 KB per CU are in flight; the wait in front of each write is a CONSTANT s_waitcnt vmcnt(31) (loads return in order);
//   * two LDS stage buffers; ONE barrier per stage, in front of the stage's last 16-channel chunk: behind it the other buffer is
//     complete (every wave waited for its writes) and this buffer is free (every wave holds its last fragments in registers), so
//     the chunk's MFMAs run while the next stage's first fragments are read -- the matrix pipe has work across the barrier;
//   * every instruction of the K loop is one asm volatile statement (order = program order; hipcc allocates registers and
//     computes addresses), fragment reads one chunk ahead with counted lgkmcnt;
//   * NOTHING is in flight to a register while compiler-scheduled code runs.  A load writes its destination when the data arrives;
//     hipcc believes the asm statement wrote it and may copy or spill the register at once (first build: scratch stores of the
//     destinations right behind the loads, around the epilogue's register pressure; capping the allocator under named registers
//     does not work either -- amdgpu_num_vgpr splits its budget evenly between VGPRs and AGPRs, and hipcc parks values in AGPRs).
//     So the stream DRAINS into LDS at a tile's end: the tile's last stage requests nothing, writes stage g + 1 during its first
//     three chunks and stage g + 2 -- its loads are a stage old -- into the buffer its barrier has just freed during the fourth;
//     both buffers then hold the NEXT tile's first two stages, the epilogue (staged through 32 KB beside the buffers) runs with an
//     empty queue, and the next tile's first stage requests its third and fourth stage (32 loads) while it computes from LDS;
//   * persistent over tiles in the ring kernel's XCD-aware order; the accumulators are cleared by the epilogue (tile_epilogue ZACC).
// K order per accumulator = conv_ring_kernel's (ascending 16-channel chunks): identical bits.
#pragma once

namespace g4 {

template <typename T> __device__ __forceinline__ void mfma_acc(f32x16& d, const u32x4& a, const u32x4& b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
template <int OFF> __device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write128(uint32_t addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// 16 bytes per lane from (64-bit scalar base) + (32-bit per-lane byte offset).  The destination is written when the data arrives:
// nothing but the ds_write behind the counted vmcnt wait reads it, and no compiler-scheduled code runs in between (see the header).
__device__ __forceinline__ void gload128(u32x4& dst, uint32_t voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// (the builtin, not an asm statement, so that hipcc sees an instruction between an asm definition and its asm use -- see attn_pipe.hip)
template <int N> __device__ __forceinline__ void wait_lgkm() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

enum { FIRST = 0, MID = 1, LAST = 2 };

}  // namespace g4

// EPI: 1 = GEGLU, 4 = GEGLU with the LayerNorm folded in (see tile_epilogue; 2 = bias (+ residual) compiles, but hipcc moves the
// accumulators through scratch in that epilogue and the launcher does not offer it).  NPH_ODD: K / 64 is odd (a tile's
// first stage then alternates between the two buffers from tile to tile; the code is straight-line per parity).
// ABL (ablation builds, knob conv_dbg; results are garbage): 1 no load / write stream, 2 no MFMA, 4 no fragment reads, 8 no epilogue
template <typename T, int EPI, bool NPH_ODD, int RESM = 0, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_g4_kernel(ConvParams p) {
    constexpr int NT = 256, WN = 2, TM = 4, TN = 4;
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int ROWB = BK * 2;                              // 128-byte rows: one line per row and stage
    constexpr int PANEL = 256 * ROWB;                         // 32 KB: the token rows, then the weight rows
    constexpr int NP = 16;                                    // 16-byte pieces per lane and stage (8 token rows + 8 weight rows)
    constexpr bool LNF = EPI == 4;
    constexpr int EPI_OFF = 4 * PANEL;                        // the epilogue's staging rows: 32 KB beside the stage buffers
    constexpr int EPI_ROWB = (EPI == 1 || EPI == 4) ? (TN / 2) * 64 : TN * 64;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EPI_ROWB;      // 16 KB (GEGLU) / 32 KB
    constexpr int CVB = LNF ? 2 * BN * 4 : 0;                 // the tile's fp32 column vectors c1 | c2 (LayerNorm-folded epilogue)
    constexpr int STAMP_OFF = EPI_OFF + EPI_BYTES + CVB;      // (ABL bit 4: 256 cycle stamps of workgroup 0's wave 0)
    constexpr int LDS_BYTES = STAMP_OFF + ((ABL & 16) ? 1024 : 0);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    float* const cvec = (float*)(lds + EPI_OFF + EPI_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid_s / WN, wn = wid_s % WN;
    const char* xg = (const char*)p.x;
    const char* wg = (const char*)p.w;
    const int Kb = p.Cin * 2;                                 // bytes per operand row
    const int nph = p.Cin / BK;                               // >= 2 (launcher)

    // persistent tile walk (conv_ring_kernel's: XCD x takes the x-th eighth of each round, cout groups pinned to XCD groups)
    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / p.ngroups, tn_g = p.tiles_n / p.ngroups;
    const int grp = (blockIdx.x % 8) / xg_n;
    const int ntiles = (int)(p.nblocks / p.ngroups);
    const int tile_first = ((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const int tile_step = xg_n * per_xcd;
    if (tile_first >= ntiles) return;
    const int my_tiles = (ntiles - tile_first + tile_step - 1) / tile_step;
    auto tile_m0 = [&](int j) { return (long)((uint32_t)j / (uint32_t)tn_g) * BM; };
    auto tile_n0 = [&](int j) { return (grp * tn_g + (int)((uint32_t)j % (uint32_t)tn_g)) * BN; };

    // ---- staging geometry: lane t moves chunk c = t & 7 of rows r0 + 32 i (r0 = t >> 3): 8 lanes = one 128-byte line.
    // LDS: [tokens, stage buffer 0 | tokens, buffer 1 | weights, buffer 0 | weights, buffer 1], 32 KB each: the buffer and the
    // piece / fragment block are IMMEDIATE offsets of the ds instructions (<= 61440), one address register per panel.
    const int r0 = tid >> 3, c8 = tid & 7;
    const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    // row R, chunk d lives at R * 128 + ((d ^ ((R >> 1) & 7)) * 16): fragment reads (16 rows of one chunk per lane group) and the
    // 8-lane write groups (one row) are both conflict-free
    const uint32_t wr_off = (uint32_t)(r0 * ROWB + ((c8 ^ ((r0 >> 1) & 7)) << 4));
    const uint32_t wrA = lds_u32 + wr_off, wrW = lds_u32 + 2 * PANEL + wr_off;
    const uint32_t voff = (uint32_t)(r0 * Kb + c8 * 16);      // the lane's part of every piece's address (M % 256 == 0: no row clamp)
    const long pstep = 32L * Kb;                              // piece i starts 32 rows further: scalar bases per piece
    // fragment read addresses: lane (col, hi) of chunk ks reads row (block row 0 + col), 16-byte chunk (2 ks + hi)
    const int swz = (col >> 1) & 7;
    uint32_t rdX[4], rdW[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t ko = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
        rdX[ks] = lds_u32 + (wm * 128 + col) * ROWB + ko;
        rdW[ks] = lds_u32 + 2 * PANEL + (wn * 128 + col) * ROWB + ko;
    }

    // ---- producer: the global-load stream, stage by stage across tile boundaries; saturates on the last stage of the last tile
    int ptile = tile_first, pk = 0, pleft = my_tiles;
    const char* pA = xg + tile_m0(ptile) * Kb;
    const char* pW = wg + (long)tile_n0(ptile) * Kb;
    auto producer_advance = [&]() {
        if (pk + 1 < nph) {
            ++pk;
            pA += ROWB;
            pW += ROWB;
        } else if (pleft > 1) {
            --pleft;
            pk = 0;
            ptile += tile_step;
            pA = xg + tile_m0(ptile) * Kb;
            pW = wg + (long)tile_n0(ptile) * Kb;
        }
    };
    u32x4 S[2][NP];                                           // the two register sets: stage s travels through set s & 1
    auto load_piece = [&](auto sc, auto ic) {
        constexpr int s = decltype(sc)::value, i = decltype(ic)::value;
        if constexpr (i < 8) g4::gload128(S[s][i], voff, pA + i * pstep);
        else g4::gload128(S[s][i], voff, pW + (i - 8) * pstep);
    };
    auto write_piece = [&](auto bc, auto sc, auto ic) {       // piece i of register set s -> stage buffer b
        constexpr int b = decltype(bc)::value, s = decltype(sc)::value, i = decltype(ic)::value;
        if constexpr (i < 8) g4::lds_write128<b * PANEL + i * 4096>(wrA, S[s][i]);
        else g4::lds_write128<b * PANEL + (i - 8) * 4096>(wrW, S[s][i]);
    };

    u32x4 F[2][8];                                            // fragment sets: [set][0..3] token blocks, [4..7] weight blocks
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) zero_acc_mfma<T>(acc[a][b]);
    auto read_frag = [&](auto bc, auto ksc, auto fsc, auto jc) {      // fragment j of chunk ks of stage buffer b -> set fs
        constexpr int b = decltype(bc)::value, ks = decltype(ksc)::value, fs = decltype(fsc)::value, j = decltype(jc)::value;
        if constexpr (j < 4) g4::lds_read128<b * PANEL + j * 4096>(F[fs][j], rdX[ks]);
        else g4::lds_read128<b * PANEL + (j - 4) * 4096>(F[fs][j], rdW[ks]);
    };

    // ---- prologue = the state a tile's last stage leaves behind: stages 0 and 1 in the two buffers, nothing in flight
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    static_for<NP>([&](auto ic) { load_piece(I0{}, ic); });
    producer_advance();
    static_for<NP>([&](auto ic) { load_piece(I1{}, ic); });
    producer_advance();
    g4::wait_vm<0>();
    static_for<NP>([&](auto ic) { write_piece(I0{}, I0{}, ic); });
    static_for<NP>([&](auto ic) { write_piece(I1{}, I1{}, ic); });
    g4::wait_lgkm<0>();
    asm volatile("s_barrier" ::: "memory");
    static_for<8>([&](auto jc) { read_frag(I0{}, I0{}, I0{}, jc); });
    g4::wait_lgkm<0>();

    // ---- one stage g (buffer BUF = g & 1) = four 16-channel chunks.  Chunk q runs its 16 MFMAs from fragment set q & 1; behind
    //      MFMA j < 8 it reads fragment j of the next chunk (chunk 3: of stage g + 1's first chunk) into the other set.  The stream:
    //        MID    behind MFMAs 8 .. of chunks 0 - 2: sixteen (write piece i of stage g + 1, request piece i of stage g + 3) pairs
    //        LAST   the same slots write stage g + 1 and request nothing; chunk 3 writes stage g + 2 into THIS buffer (free since the barrier)
    //        FIRST  the same slots request stages g + 2 and g + 3, two pieces per slot (both buffers were filled by the previous LAST)
    //      The barrier sits in front of chunk 3.
    // The stream's instructions take the 32 slots behind MFMAs 8 .. 15 of the four chunks, ONE per slot (a write + a request in one
    // 32-cycle MFMA gap cost ~ 25 cycles of matrix-pipe time per pair: tools/g4_stamps.py, 2500 against 2110 cycles per stage):
    // slot s = 8 q + j - 8.  MID: slots 0 .. 23 (in front of the barrier) carry W W G W W G ... = the sixteen writes of stage g + 1 and
    // the first eight requests of stage g + 3, slots 24 .. 31 the other eight requests.  Request i follows write i (same registers).
    auto chunk = [&](auto bufc, auto modec, auto qc) {
        constexpr int BUF = decltype(bufc)::value, MODE = decltype(modec)::value, q = decltype(qc)::value;
        constexpr int fs = q & 1;
        using BufC = std::integral_constant<int, BUF>;
        using OthC = std::integral_constant<int, BUF ^ 1>;
        static_for<16>([&](auto jc) {
            constexpr int j = decltype(jc)::value, a = j / 4, b = j % 4;
            if constexpr (!(ABL & 2)) g4::mfma_acc<T>(acc[a][b], F[fs][4 + a], F[fs][b]);
            if constexpr (j < 8 && !(ABL & 4)) {
                if constexpr (q < 3) read_frag(BufC{}, std::integral_constant<int, q + 1>{}, std::integral_constant<int, fs ^ 1>{}, jc);
                else read_frag(OthC{}, I0{}, std::integral_constant<int, fs ^ 1>{}, jc);
            }
            if constexpr (ABL & 1) {
            } else if constexpr (MODE == g4::LAST && q == 3) {
                g4::wait_vm<15 - j>();                        // (younger: the rest of stage g + 2's requests)
                write_piece(BufC{}, BufC{}, jc);
            } else if constexpr (j >= 8) {
                constexpr int sl = 8 * q + j - 8;             // stream slot 0 .. 31
                if constexpr (MODE == g4::FIRST) {
                    if constexpr (sl < 16) load_piece(BufC{}, std::integral_constant<int, sl>{});
                    else load_piece(OthC{}, std::integral_constant<int, sl - 16>{});
                    if constexpr (sl == 15 || sl == 31) producer_advance();
                } else if constexpr (sl < 24 && sl % 3 != 2) {
                    constexpr int i = 2 * (sl / 3) + sl % 3;  // write i of stage g + 1
                    // younger than its request: the rest of stage g + 1's, all of stage g + 2's, and this stage's i / 2 (MID)
                    g4::wait_vm<(MODE == g4::MID ? 31 - i + i / 2 : 31 - i)>();
                    write_piece(OthC{}, OthC{}, std::integral_constant<int, i>{});
                } else if constexpr (MODE == g4::MID) {
                    constexpr int i = sl < 24 ? sl / 3 : sl - 16;     // request i of stage g + 3
                    load_piece(OthC{}, std::integral_constant<int, i>{});
                    if constexpr (sl == 31) producer_advance();
                }
            }
        });
        // LDS writes issued behind the chunk's fragment reads (LDS returns in order): the W slots of this chunk
        constexpr int NWR = (MODE == g4::FIRST || (ABL & 1)) ? 0 : (q == 0 ? 6 : 5);
        if constexpr (q == 2) {
            g4::wait_lgkm<0>();                               // this wave's writes of stage g + 1 and its last fragments of stage g
            asm volatile("s_barrier" ::: "memory");
        } else if constexpr (q == 3) {
            g4::wait_lgkm<0>();
        } else {
            g4::wait_lgkm<NWR>();
        }
    };
    auto stage = [&](auto bufc, auto modec) {
        chunk(bufc, modec, std::integral_constant<int, 0>{});
        chunk(bufc, modec, std::integral_constant<int, 1>{});
        chunk(bufc, modec, std::integral_constant<int, 2>{});
        chunk(bufc, modec, std::integral_constant<int, 3>{});
    };
    using MFirst = std::integral_constant<int, g4::FIRST>;
    using MMid = std::integral_constant<int, g4::MID>;
    using MLast = std::integral_constant<int, g4::LAST>;

    // ---- a tile's epilogue: the queue is empty, no register is in flight; the accumulators are cleared by tile_epilogue (ZACC)
    int ctile = tile_first;
    int nstamp = 0;
    auto stamp = [&]() {                                      // (ablation builds only) shader-clock stamp of this point, wave 0 of workgroup 0
        if constexpr ((ABL & 16) != 0) {
            if (blockIdx.x == 0 && wid_s == 0 && nstamp < 256) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
                if (lane == 0) ((uint32_t*)(lds + STAMP_OFF))[nstamp] = t;
                ++nstamp;
            }
        }
    };
    auto tile_end = [&](auto nbufc) {                         // nbufc: the buffer that holds the next tile's first stage
        const long m0 = tile_m0(ctile);
        const int n0 = tile_n0(ctile);
        // hipcc does not know that the statements above are MFMAs: without this it copies / spills an accumulator one state behind
        // the MFMA that writes it (seen: scratch stores of a[0:15] right behind the tile's last MFMA).  Every accumulator is
        // "redefined" here, behind the wait states an 8-pass MFMA result needs, so no compiler access can sit above it.
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                       "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3])
                     :: "memory");
        // the epilogue's per-lane addressing is invariant across tiles: keep hipcc from hoisting ~40 registers of it out of the tile loop
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        const int col_e = lane_e & 31, wm_e = wid_e / WN, wn_e = wid_e % WN;
        float ln_pre[2 * TM];
        if constexpr (LNF) {
            // this tile's column vectors c1 | c2 -> LDS, the rows' mean / rstd from the producer's statistics
            const int tid_e = wid_e * 64 + lane_e;
            if (tid_e < BN / 2) {
                const int v = tid_e / (BN / 4), idx = (tid_e % (BN / 4)) * 4;
                *(f32x4*)(cvec + v * BN + idx) = *(const f32x4*)((v ? p.ln_c2 : p.ln_c1) + n0 + idx);
            }
            float mus[TM], rstds[TM];
            epi_ln_row_stats<TM>(p, m0, wm_e, col_e, mus, rstds);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                ln_pre[b] = mus[b];
                ln_pre[TM + b] = rstds[b];
            }
            __syncthreads();
        }
        if constexpr (ABL & 8) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) zero_acc_mfma<T>(acc[a][b]);
        } else
        tile_epilogue<T, NT, TM, TN, EPI, false, false, false, WN, RESM, true>(p, acc, lds + EPI_OFF, m0, n0, wm_e, wn_e, wid_e, lane_e, cvec, BN, LNF ? ln_pre : nullptr);
        ctile += tile_step;
        // the next stage's first fragments again (read once by the tile's last chunk): they need not live across the epilogue
        static_for<8>([&](auto jc) { read_frag(nbufc, I0{}, I0{}, jc); });
        g4::wait_lgkm<0>();
    };

    if constexpr (!NPH_ODD) {
        const int npair = (nph - 2) / 2;
        for (int t = 0; t < my_tiles; ++t) {
            stamp();
            stage(I0{}, MFirst{});
            stamp();
            for (int j = 0; j < npair; ++j) {
                stage(I1{}, MMid{});
                stamp();
                stage(I0{}, MMid{});
                stamp();
            }
            stage(I1{}, MLast{});
            stamp();
            tile_end(I0{});
        }
    } else {
        const int npair = (nph - 3) / 2;
        for (int t = 0; t < my_tiles; t += 2) {
            stamp();
            stage(I0{}, MFirst{});
            stamp();
            for (int j = 0; j < npair; ++j) {
                stage(I1{}, MMid{});
                stamp();
                stage(I0{}, MMid{});
                stamp();
            }
            stage(I1{}, MMid{});
            stamp();
            stage(I0{}, MLast{});
            stamp();
            tile_end(I1{});
            stamp();
            if (t + 1 >= my_tiles) break;
            stage(I1{}, MFirst{});
            for (int j = 0; j < npair; ++j) {
                stage(I0{}, MMid{});
                stage(I1{}, MMid{});
            }
            stage(I0{}, MMid{});
            stage(I1{}, MLast{});
            tile_end(I0{});
        }
    }
    if constexpr ((ABL & 16) != 0) {
        __syncthreads();
        if (blockIdx.x == 0 && wid_s == 0)
            for (int k = lane; k < 256; k += 64) ((uint32_t*)p.y)[k] = k < nstamp ? ((uint32_t*)(lds + STAMP_OFF))[k] : 0u;
    }
}

// ---- the same loop for TWO workgroups per CU (256 x 128 tile, 256 registers per wave): one workgroup's epilogue under the other's K loop
// tools/g4_stamps.py on the 256 x 256 kernel above: its K loop runs at 2460 cycles per 2048-cycle stage, but the fused GEGLU epilogue
// takes 13 400 cycles per tile -- as long as the five stages of a K = 320 tile -- with the matrix pipes idle, and a lone wave per SIMD
// issues one vector instruction per 4 cycles whatever it is (two interleaved waves pair their 2-cycle instructions).  Here a
// workgroup is 4 waves x 256 registers, wave = 128 tokens x 64 packed rows (8 accumulators = 128 AGPRs), and two workgroups share a
// CU: while one converts and stores its tile the other one's MFMAs have the matrix pipe, and both waves of a SIMD feed the vector
// unit.  57 KB of LDS per workgroup: ONE stage buffer (256 + 128 rows of 128 bytes) + 8 KB of epilogue staging + the column vectors.
//   stage g:  chunks 0 - 2 read their successor's fragments from the buffer; barrier A (every wave holds chunk 3's fragments);
//             chunk 3: per piece (wait, ds_write piece i of stage g + 1 from its register, request piece i of stage g + 2 into it);
//             barrier B; read stage g + 1's first fragments.  One register set: a request has one stage time (2 x 1024 MFMA cycles
//             with the other workgroup on the pipe) to arrive; what this wave's LDS round trip at the buffer turnover exposes, the
//             other workgroup's waves fill.
//   tile boundaries as above: the last stage requests nothing, the epilogue runs with an empty queue, the next tile's first stage
//   (already in the buffer) requests its second stage up front.
template <typename T, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_g4b_kernel(ConvParams p) {
    constexpr int NT = 256, WN = 2, TM = 4, TN = 2;
    constexpr int BM = 256, BN = 128, BK = 64;
    constexpr int ROWB = BK * 2;
    constexpr int PANEL_A = BM * ROWB, PANEL_W = BN * ROWB;   // 32 KB + 16 KB
    constexpr int NPA = 8, NPW = 4, NP = NPA + NPW;           // 16-byte pieces per lane and stage
    constexpr bool LNF = EPI == 4;
    constexpr int EPI_OFF = PANEL_A + PANEL_W;
    constexpr int EPI_BYTES = (NT / 64) * 32 * ((TN / 2) * 64);       // 8 KB
    constexpr int CVB = LNF ? 2 * BN * 4 : 0;
    constexpr int LDS_BYTES = EPI_OFF + EPI_BYTES + CVB;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    float* const cvec = (float*)(lds + EPI_OFF + EPI_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid_s / WN, wn = wid_s % WN;
    const char* xg = (const char*)p.x;
    const char* wg = (const char*)p.w;
    const int Kb = p.Cin * 2;
    const int nph = p.Cin / BK;                               // >= 2 (launcher)

    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / p.ngroups, tn_g = p.tiles_n / p.ngroups;
    const int grp = (blockIdx.x % 8) / xg_n;
    const int ntiles = (int)(p.nblocks / p.ngroups);
    const int tile_first = ((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const int tile_step = xg_n * per_xcd;
    if (tile_first >= ntiles) return;
    const int my_tiles = (ntiles - tile_first + tile_step - 1) / tile_step;
    auto tile_m0 = [&](int j) { return (long)((uint32_t)j / (uint32_t)tn_g) * BM; };
    auto tile_n0 = [&](int j) { return (grp * tn_g + (int)((uint32_t)j % (uint32_t)tn_g)) * BN; };

    const int r0 = tid >> 3, c8 = tid & 7;
    const uint32_t lds_u32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t wr_off = (uint32_t)(r0 * ROWB + ((c8 ^ ((r0 >> 1) & 7)) << 4));
    const uint32_t wrA = lds_u32 + wr_off, wrW = lds_u32 + PANEL_A + wr_off;
    const uint32_t voff = (uint32_t)(r0 * Kb + c8 * 16);
    const long pstep = 32L * Kb;
    const int swz = (col >> 1) & 7;
    uint32_t rdX[4], rdW[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t ko = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
        rdX[ks] = lds_u32 + (wm * 128 + col) * ROWB + ko;
        rdW[ks] = lds_u32 + PANEL_A + (wn * 64 + col) * ROWB + ko;
    }

    int ptile = tile_first, pk = 0, pleft = my_tiles;
    const char* pA = xg + tile_m0(ptile) * Kb;
    const char* pW = wg + (long)tile_n0(ptile) * Kb;
    auto producer_advance = [&]() {
        if (pk + 1 < nph) {
            ++pk;
            pA += ROWB;
            pW += ROWB;
        } else if (pleft > 1) {
            --pleft;
            pk = 0;
            ptile += tile_step;
            pA = xg + tile_m0(ptile) * Kb;
            pW = wg + (long)tile_n0(ptile) * Kb;
        }
    };
    u32x4 S[NP];
    auto load_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < NPA) g4::gload128(S[i], voff, pA + i * pstep);
        else g4::gload128(S[i], voff, pW + (i - NPA) * pstep);
    };
    auto write_piece = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < NPA) g4::lds_write128<i * 4096>(wrA, S[i]);
        else g4::lds_write128<(i - NPA) * 4096>(wrW, S[i]);
    };
    u32x4 F[2][6];                                            // [set][0..3] token blocks, [4..5] weight blocks
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) zero_acc_mfma<T>(acc[a][b]);
    auto read_frag = [&](auto ksc, auto fsc, auto jc) {
        constexpr int ks = decltype(ksc)::value, fs = decltype(fsc)::value, j = decltype(jc)::value;
        if constexpr (j < 4) g4::lds_read128<j * 4096>(F[fs][j], rdX[ks]);
        else g4::lds_read128<(j - 4) * 4096>(F[fs][j], rdW[ks]);
    };
    using I0 = std::integral_constant<int, 0>;

    // prologue: stage 0 in the buffer, nothing in flight (= what a tile's last stage leaves behind)
    static_for<NP>([&](auto ic) { load_piece(ic); });
    producer_advance();
    g4::wait_vm<0>();
    static_for<NP>([&](auto ic) { write_piece(ic); });
    g4::wait_lgkm<0>();
    asm volatile("s_barrier" ::: "memory");
    static_for<6>([&](auto jc) { read_frag(I0{}, I0{}, jc); });
    g4::wait_lgkm<0>();

    auto stage = [&](auto modec) {
        constexpr int MODE = decltype(modec)::value;
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value, fs = q & 1;
            static_for<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value, a = j / 4, b = j % 4;
                g4::mfma_acc<T>(acc[a][b], F[fs][4 + a], F[fs][b]);
                if constexpr (q < 3 && j < 6) read_frag(std::integral_constant<int, q + 1>{}, std::integral_constant<int, fs ^ 1>{}, jc);
                if constexpr (MODE == g4::FIRST && q == 0) {
                    // the tile's second stage, requested up front (the previous tile's last stage requested nothing)
                    if constexpr (j < 4) {
                        load_piece(std::integral_constant<int, 2 * j>{});
                        load_piece(std::integral_constant<int, 2 * j + 1>{});
                    } else {
                        load_piece(std::integral_constant<int, 4 + j>{});
                        if constexpr (j == 7) producer_advance();
                    }
                }
                if constexpr (q == 3) {
                    // buffer turnover: piece i of stage g + 1 out of its register, piece i of stage g + 2 requested into it
                    auto piece = [&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        g4::wait_vm<NP - 1 - i + (MODE == g4::LAST ? 0 : i)>();       // younger: the rest of stage g + 1's requests (+ this stage's i)
                        write_piece(ic);
                        if constexpr (MODE != g4::LAST) load_piece(ic);
                    };
                    if constexpr (j < 4) {
                        piece(std::integral_constant<int, 2 * j>{});
                        piece(std::integral_constant<int, 2 * j + 1>{});
                    } else {
                        piece(std::integral_constant<int, 4 + j>{});
                    }
                }
            });
            if constexpr (q < 2) {
                g4::wait_lgkm<0>();
            } else if constexpr (q == 2) {
                g4::wait_lgkm<0>();                           // chunk 3's fragments are in registers:
                asm volatile("s_barrier" ::: "memory");       // barrier A -- nobody reads the buffer any more
            } else {
                if constexpr (MODE != g4::LAST) producer_advance();
                g4::wait_lgkm<0>();
                asm volatile("s_barrier" ::: "memory");       // barrier B -- stage g + 1 is complete
                static_for<6>([&](auto jc) { read_frag(I0{}, I0{}, jc); });
                g4::wait_lgkm<0>();
            }
        });
    };
    using MFirst = std::integral_constant<int, g4::FIRST>;
    using MMid = std::integral_constant<int, g4::MID>;
    using MLast = std::integral_constant<int, g4::LAST>;

    int ctile = tile_first;
    for (int t = 0; t < my_tiles; ++t) {
        __builtin_amdgcn_s_setprio(1);            // the K loop's MFMAs go first; the other workgroup's epilogue fills the vector slots between them
                                                  // (a start skew between the CU's two workgroups changes nothing: profiles/r06_g4b_phase_skew.log)
        stage(MFirst{});
        for (int k = 2; k < nph; ++k) stage(MMid{});
        stage(MLast{});
        __builtin_amdgcn_s_setprio(0);
        // ---- epilogue: empty queue, no register in flight; the accumulators are cleared by tile_epilogue (ZACC)
        const long m0 = tile_m0(ctile);
        const int n0 = tile_n0(ctile);
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3])
                     :: "memory");
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        const int col_e = lane_e & 31, wm_e = wid_e / WN, wn_e = wid_e % WN;
        float ln_pre[2 * TM];
        if constexpr (LNF) {
            const int tid_e = wid_e * 64 + lane_e;
            if (tid_e < BN / 2) {
                const int v = tid_e / (BN / 4), idx = (tid_e % (BN / 4)) * 4;
                *(f32x4*)(cvec + v * BN + idx) = *(const f32x4*)((v ? p.ln_c2 : p.ln_c1) + n0 + idx);
            }
            float mus[TM], rstds[TM];
            epi_ln_row_stats<TM>(p, m0, wm_e, col_e, mus, rstds);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                ln_pre[b] = mus[b];
                ln_pre[TM + b] = rstds[b];
            }
            __syncthreads();
        }
        tile_epilogue<T, NT, TM, TN, EPI, false, false, false, WN, 0, true>(p, acc, lds + EPI_OFF, m0, n0, wm_e, wn_e, wid_e, lane_e, cvec, BN, LNF ? ln_pre : nullptr);
        ctile += tile_step;
        // (the first fragments of the next tile, read behind barrier B above, need not live across the epilogue)
        static_for<6>([&](auto jc) { read_frag(I0{}, I0{}, jc); });
        g4::wait_lgkm<0>();
    }
}

template <typename T, int EPI>
static int launch_g4b_t(ConvParams p, hipStream_t stream) {
    constexpr int BM = 256, BN = 128;
    p.tiles_n = p.Cout / BN;
    p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (p.nblocks > 0x7fffffffL || p.Cout % BN != 0 || p.Cin % 64 != 0 || p.Cin < 128 || p.M % BM != 0) {
        im360_set_error("gemm_g4b: unsupported shape");
        return IM360_ERR_ARG;
    }
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    const long want = (p.nblocks + 7) / 8 * 8;
    const unsigned grid = (unsigned)(want < 2L * ncu ? want : 2L * ncu);      // two workgroups per CU
    {
        const long wbytes = (long)p.tiles_n * BN * p.Cin * 2;
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
    hipLaunchKernelGGL((gemm_g4b_kernel<T, EPI>), dim3(grid), dim3(256), 0, stream, p);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

template <typename T, int EPI>
static int launch_g4_t(ConvParams p, hipStream_t stream) {
    constexpr int BM = 256, BN = 256;
    p.tiles_n = p.Cout / BN;
    p.nblocks = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (p.nblocks > 0x7fffffffL || p.Cout % BN != 0 || p.Cin % 64 != 0 || p.Cin < 128 || p.M % BM != 0) {
        im360_set_error("gemm_g4: unsupported shape");
        return IM360_ERR_ARG;
    }
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    const long want = (p.nblocks + 7) / 8 * 8;
    const unsigned grid = (unsigned)(want < ncu ? want : ncu);
    {
        const long wbytes = (long)p.tiles_n * BN * p.Cin * 2;
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
    const bool odd = ((p.Cin / 64) & 1) != 0;
    static_assert(EPI == 1 || EPI == 4, "built for the fused GEGLU epilogues (the plain epilogue on this tile was measured 7 - 33 % slower than the 256 x 320 loop: profiles/r06_g4_first_ab.log)");
#ifdef IM360_G4_ABL
    if constexpr (EPI == 1 && std::is_same<T, __bf16>::value) {
        // ablation / cycle-stamp builds (make CXXFLAGS+=-DIM360_G4_ABL; tools/g4_stamps.py, knob conv_dbg): 16 = stamps, +1 no stream, +8 no epilogue
        const int dbg = knob(KNOB_CONV_DBG);
#define IM360_G4_CASE(a) if (dbg == a) { if (odd) hipLaunchKernelGGL((gemm_g4_kernel<T, EPI, true, 0, a>), dim3(grid), dim3(256), 0, stream, p); else hipLaunchKernelGGL((gemm_g4_kernel<T, EPI, false, 0, a>), dim3(grid), dim3(256), 0, stream, p); IM360_CHECK_LAUNCH(); return IM360_OK; }
        IM360_G4_CASE(16) IM360_G4_CASE(17) IM360_G4_CASE(25)
#undef IM360_G4_CASE
    }
#endif
    if (odd) hipLaunchKernelGGL((gemm_g4_kernel<T, EPI, true, 0>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_g4_kernel<T, EPI, false, 0>), dim3(grid), dim3(256), 0, stream, p);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}
