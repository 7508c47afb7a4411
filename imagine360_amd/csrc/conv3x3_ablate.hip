// Ablation-only kernels of the convolution / GEMM family: compiled ONLY into `make ablate` builds (-DIM360_ABLATE), by inclusion
// from conv3x3.hip INSIDE its `namespace im360` (that file defines ConvParams, tile_epilogue, the LDS-DMA helpers and the launch
// plumbing used here).  Compiled on its own this file is an empty translation unit.  Both kernels are measured-and-rejected A/B
// variants: gemm_a3_kernel (identical results, 0.85 - 0.97 x the staggered loop, profiles/r04_gemm_a3_ab.log) and conv_halo_kernel
// (a different fp32 summation order, 4 - 30 % slower than the two-stage kernel, profiles/r02_ab_conv_halo.log); DESIGN.md section 3d.
#if defined(IM360_ABLATE) && defined(IM360_CONV3X3_INCLUDES_ABLATE)

// ---- token-major GEMM with TWO activation stages in flight ("A3", round 4, knob conv_ring 10) -----------------------------
// The staggered loop above holds two 64-channel stages of both operands, so ONE is in flight while the other is consumed, and a
// stage needs longer to arrive (1.1 - 2 us) than its MFMAs take (1.07 us): the loop runs at 53 - 65 % of the matrix rate
// (cycle stamps).  The activation rows are what comes from HBM / the Infinity Cache; the weight rows hit the L2.
// Here the activation tile gets THREE 64-channel buffers (two stages in flight) and the weight tile three HALF stages of 32
// channels (64-byte row segments: half the L2 path's efficiency on 5/9 of the bytes, still below the MFMA time) -- 156 KB:
//   LDS map: A0 | B0 | B1 | A1 | A2 | B2; the next tile's A0 / B0 / B1 are requested under the epilogue, which stages through A1 ..
// Intervals as in MODE 3 (R fetches the fragments of two 16-channel chunks, C runs their 20 MFMAs; the wave groups one interval
// apart).  In absolute intervals t (stage st: 4 st .. 4 st + 3):
//   t = 4 st      requests B half 2 st + 2 (slot of half 2 st - 1, last read at 4 st - 1), then A stage st + 2 (slot of st - 1)
//   t = 4 st + 2  requests B half 2 st + 3 (slot of half 2 st, last read at 4 st + 1)
//   end of 4 st + 1: B half 2 st + 1 has to be in LDS for every wave; end of 4 st + 3: B half 2 st + 2 and A stage st + 1
// (leading group: requests in its R intervals, waits at the end of its C intervals; trailing group: requests between the MFMAs of
// its C intervals one interval earlier in its own frame, waits at the end of its R intervals).  vmcnt retires in order, so each
// wait names how many YOUNGER requests may stay in flight: the B half always goes out in front of the A stage of the same
// interval, and A (st + 2) stays in flight across both waits of stage st.  Same accumulation order as every other loop.
// BCM (round 5, knob conv_ring 11): the weight operand arrives CHUNK-MAJOR -- [K / 32][rows][32 channels] instead of [rows][K] -- so that a
// 32-channel half stage of the tile is one contiguous 20 KB block and every LDS-DMA instruction of a wave covers whole 128-byte
// lines (two 64-byte rows per line) instead of 64-byte row segments: round 4 measured those at half the L2 -> LDS rate
// (profiles/r04_dma_seg_probe.txt) and blamed them for this kernel's 3 - 15 % deficit.  Same LDS image, same arithmetic.
template <typename T, int TN, int EPI, bool GNS = false, int RESM = 0, bool BCM = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_a3_kernel(ConvParams p) {
    constexpr int NT = 512, WN = 2, TM = 2, BM = 256, BN = WN * TN * 32;
    constexpr int TILE_A = BM * 128, TILE_BH = BN * 64;
    constexpr int A0_OFF = 0, B0_OFF = TILE_A, B1_OFF = B0_OFF + TILE_BH, A1_OFF = B1_OFF + TILE_BH, A2_OFF = A1_OFF + TILE_A, B2_OFF = A2_OFF + TILE_A;
    constexpr int RING_BYTES = B2_OFF + TILE_BH;
    constexpr int EPI_ROWB = (EPI == 1 || EPI == 4) ? (TN / 2) * 64 : TN * 64;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EPI_ROWB, EPI_OFF = A1_OFF;
    static_assert(EPI_OFF + EPI_BYTES <= RING_BYTES, "epilogue staging inside A1 | A2 | B2");
    constexpr bool LNF = EPI == 3 || EPI == 4, BIAS_LDS = EPI == 2 || EPI == 5;
    constexpr int CVB = LNF ? 2 * BN * 4 : (BIAS_LDS ? BN * 2 : 0);
    constexpr int LDS_BYTES = RING_BYTES + 2 * CVB;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    float* const cvec0 = (float*)(lds + RING_BYTES);
    int cpar = 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const int grp = wid_s >> 2;
    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w;
    const T* zero = (const T*)g_zero_chunk;
    // tile walk: as conv_ring_kernel
    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / p.ngroups, tn_g = p.tiles_n / p.ngroups;
    const int cgrp = (blockIdx.x % 8) / xg_n;
    const long ntiles = p.nblocks / p.ngroups;
    const long tile_first = (long)((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const long tile_step = (long)xg_n * per_xcd;
    auto tile_m0 = [&](long j) { return (long)((uint32_t)j / (uint32_t)tn_g) * BM; };
    auto tile_n0 = [&](long j) { return (cgrp * tn_g + (int)((uint32_t)j % (uint32_t)tn_g)) * BN; };
    const int K = p.Cin;
    const int nst = K / 64, nh = 2 * nst;
    const uint32_t lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
    auto a_off = [&](int s) { return s == 0 ? A0_OFF : (s == 1 ? A1_OFF : A2_OFF); };
    auto b_off = [&](int s) { return s == 0 ? B0_OFF : (s == 1 ? B1_OFF : B2_OFF); };

    // ---- producer: activations 8 chunks per 128-byte row (4 pieces per wave and stage), weights 4 chunks per 64-byte row
    //      (2 pieces per wave and half stage, a third for waves 0 - 3 when the tile is 320 rows)
    const int srowA = tid / 8, pdA = ((tid % 8) ^ ((srowA / 2) & 7)) * 8;
    const int srowB = tid / 4, pdB = ((tid % 4) ^ ((srowB / 4) & 3)) * 8;
    const T* aptr[4];
    uint32_t amask = 0;
    const T* bptr = wg;
    const long wrows = ((long)p.Cout + 127) / 128 * 128;              // rows of the packed weight (im360_pack_conv_weight pads to 128)
    const long bstride = BCM ? 128L * 32 : 128L * K;
    const long bhalf = BCM ? wrows * 32 : 32;                         // elements between consecutive 32-channel half stages
    auto init_tile = [&](long tile) {
        const long m0 = tile_m0(tile);
        const int n0 = tile_n0(tile);
        bptr = BCM ? wg + (long)(n0 + srowB) * 32 + pdB : wg + (long)(n0 + srowB) * K + pdB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + srowA + i * 64;
            const bool ok = m < p.M;
            aptr[i] = ok ? xg + m * K + pdA : zero;
            amask = ok ? (amask | (1u << i)) : (amask & ~(1u << i));
        }
    };
    constexpr bool B3 = TN == 5;                  // rows 256 .. 319 of the weight tile
    auto piece_a = [&](int slot, auto ic) {
        constexpr int i = decltype(ic)::value;
        lds_dma16_asm(aptr[i], lds_u32 + (uint32_t)(a_off(slot) + wid_s * 1024 + i * 8192));
        aptr[i] += ((amask >> i) & 1u) ? 64 : 0;
    };
    auto piece_b = [&](int slot, auto ic) {       // pieces 0, 1 every wave; piece 2 waves 0 - 3 (B3)
        constexpr int i = decltype(ic)::value;
        if (i < 2 || wid_s < 4) lds_dma16_asm(bptr + i * bstride, lds_u32 + (uint32_t)(b_off(slot) + wid_s * 1024 + i * 8192));
    };
    constexpr int NBP = B3 ? 3 : 2;               // piece slots of a half stage (the third is empty for waves 4 - 7)
    auto issue_a = [&](int slot) { static_for<4>([&](auto ic) { piece_a(slot, ic); }); };
    auto issue_b = [&](int slot) {
        static_for<NBP>([&](auto ic) { piece_b(slot, ic); });
        bptr += bhalf;
    };
    const int nb = B3 ? (wid_s < 4 ? 3 : 2) : 2;  // this wave's requests per half stage
    auto wait_vm = [&](int n) {                   // at most n of this wave's youngest requests may still be in flight
        switch (n) {
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    auto cv_fill = [&](long m0n, int n0n, int set) {      // as conv_ring_kernel's
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
    // the first requests of a tile: A0, B0, B1 (everything the epilogue's staging does not cover)
    auto issue_head = [&]() {
        issue_b(0);
        if (nh > 1) issue_b(1);
        issue_a(0);
    };

    // fragment byte offsets
    const int swzA = (col / 2) & 7, swzB = (col / 4) & 3;
    int koffA[4], koffB[2];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) koffA[kc] = ((kc * 2 + hi) ^ swzA) * 16;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) koffB[k2] = ((k2 * 2 + hi) ^ swzB) * 16;
    const int xrow = (wm * (TM * 32) + col) * 128, wrow = (wn * (TN * 32) + col) * 64;

    long tile = tile_first;
    if (tile >= ntiles) return;
    init_tile(tile);
    if constexpr (LNF || BIAS_LDS) cv_fill(tile_m0(tile), tile_n0(tile), 0);
    issue_head();
    bool prev_full = false;
    for (;;) {
        const long m0 = tile_m0(tile);
        const int n0 = tile_n0(tile);
        float* cvec = cvec0 + cpar * (CVB / 4);
        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        constexpr int NTAIL = (EPI == 1 || EPI == 4) ? 6 : 16;
        if (prev_full) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NTAIL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (grp) {
            // the trailing group skips one interval: what the leading group requests in its first R interval goes out here
            if (nst > 1) issue_a(1);
            if (nh > 2) issue_b(2);
            if (nst > 2) issue_a(2);
            asm volatile("s_barrier" ::: "memory");
        }
        for (int st = 0; st < nst; ++st) {
            const char* ax = lds + a_off(st % 3) + xrow;
            const char* bw0 = lds + b_off((2 * st) % 3) + wrow;
            const char* bw1 = lds + b_off((2 * st + 1) % 3) + wrow;
            const bool rb2 = 2 * st + 2 < nh, ra2 = st + 2 < nst, rb3 = 2 * st + 3 < nh;       // requests of this stage's schedule
            const bool rb4 = 2 * st + 4 < nh, ra3 = st + 3 < nst;                               // (trailing group: next stage's first)
            // younger requests that may stay in flight at the two waits (see the header)
            const int c0 = ((st == 0 && nst > 1) ? 4 : 0) + (rb2 ? nb : 0) + (ra2 ? 4 : 0);
            const int c1 = (ra2 ? 4 : 0) + (rb3 ? nb : 0);
            u32x4 xf[2][TM], wf[2][TN];
            // ---- R0
            if (!grp) {
                if (st == 0 && nst > 1) issue_a(1);
                if (rb2) issue_b((2 * st + 2) % 3);
                if (ra2) issue_a((st + 2) % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(ax + koffA[kc] + b * (32 * 128));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(bw0 + koffB[kc] + a * (32 * 64));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp) wait_vm(c0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- C0
            __builtin_amdgcn_s_setprio(1);
            static_for<2 * TN * TM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                if constexpr (m % 4 == 3 && m / 4 < NBP) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp && rb3) piece_b((2 * st + 3) % 3, std::integral_constant<int, m / 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (grp && rb3) bptr += bhalf;
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!grp) wait_vm(c0);
            asm volatile("s_barrier" ::: "memory");
            // ---- R1
            if (!grp && rb3) issue_b((2 * st + 3) % 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[kc][b] = *(const u32x4*)(ax + koffA[2 + kc] + b * (32 * 128));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[kc][a] = *(const u32x4*)(bw1 + koffB[kc] + a * (32 * 64));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp && rb2) wait_vm(c1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- C1
            __builtin_amdgcn_s_setprio(1);
            static_for<2 * TN * TM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, kc = m / (TN * TM), a = (m / TM) % TN, b = m % TM;
                acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                if constexpr (m % 2 == 1 && m / 2 < NBP + 4) {
                    constexpr int j = m / 2;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (j < NBP) {
                        if (grp && rb4) piece_b((2 * st + 4) % 3, std::integral_constant<int, (j < NBP ? j : 0)>{});
                    } else {
                        if (grp && ra3) piece_a((st + 3) % 3, std::integral_constant<int, (j >= NBP ? j - NBP : 0)>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (grp && rb4) bptr += bhalf;
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!grp && rb2) wait_vm(c1);
            asm volatile("s_barrier" ::: "memory");
        }
        if (!grp) asm volatile("s_barrier" ::: "memory");
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
        asm volatile("s_barrier" ::: "memory");          // every wave is done reading operand buffers
        const long next = tile + tile_step;
        if (next < ntiles) {
            init_tile(next);
            if constexpr (LNF || BIAS_LDS) cv_fill(tile_m0(next), tile_n0(next), cpar ^ 1);
            issue_head();
        }
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        tile_epilogue<T, NT, TM, TN, EPI, true, false, GNS, WN, RESM>(p, acc, lds + EPI_OFF, m0, n0, wid_e / WN, wid_e % WN, wid_e, lane_e, cvec, BN, (EPI == 3 || EPI == 4) ? ln_pre : nullptr);
        if (next >= ntiles) break;
        cpar ^= 1;
        prev_full = m0 + BM <= p.M;
        tile = next;
    }
}

// ---- 3x3 convolution from a halo'd pixel patch ----------------------------------------------------------------------
// The kernels above stream one shifted copy of the pixel tile per tap: nine LDS-DMA loads of every activation, 9x the
// input tensor through the L2 / fabric (profiles: fetch / input = 8.9 .. 14), and the CU's global -> LDS path (~21 B/clk)
// carries 36 KB per 32-channel phase.  Here a tile is a RECTANGLE of output pixels -- 256 / Wout whole rows of width
// Wout, inside one image or covering whole images -- and the K loop runs chunk-major: for each 32-channel chunk the
// (rows + 2) x (Wout + 2) patch of input pixels (zeros outside the image) is loaded ONCE into LDS, the nine taps read
// their shifted fragments out of it, and only the weights of (tap, chunk) stream per phase (20 KB).  Activations enter
// LDS 1.3 - 2.0x instead of 9x.  Same ring / counted-vmcnt / interleaved-request / persistent-tile machinery as
// conv_ring_kernel; the accumulation order over K is chunk-major here (tap-major there), so results agree with the other
// kernels to fp32 summation order, not bit for bit.
template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_halo_kernel(ConvParams p) {
    constexpr int NT = 512, WN = 2, TM = 2, TN = 5, EPI = 0;
    constexpr int BM = 256, BN = 320, BK = 32, ROWB = 64, KC = 2;
    constexpr int APATCH = 33792;                    // bytes of one patch slot: up to 528 pixels x 64 B
    constexpr int TILE_B = BN * ROWB;                // 20480
    constexpr int LDB = 2;                           // + half a round for waves 0..3 (320 rows)
    constexpr int A0 = 0, B0 = APATCH, B1 = B0 + TILE_B, A1 = B1 + TILE_B, B2 = A1 + APATCH, B3 = B2 + TILE_B;
    constexpr int EPI_OFF = A1, EPI_BYTES = (NT / 64) * 32 * TN * 64;
    constexpr int LDS_BYTES = EPI_OFF + EPI_BYTES > B3 + TILE_B ? EPI_OFF + EPI_BYTES : B3 + TILE_B;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w;
    const T* zero = (const T*)g_zero_chunk;
    const uint32_t lds_u32 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);

    const long ntiles = p.nblocks;
    const int per_xcd = gridDim.x / 8;
    const long tile_first = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const long tile_step = gridDim.x;

    const int R = p.halo_r, HS = p.halo_seg, PW = p.halo_pw, P = p.halo_p;
    const int LA = (P * 4 + NT - 1) / NT;
    const int nchunks = p.Cin / BK;
    const int nph = 9 * nchunks;
    const int nB = LDB + (wid_s < NT / 128 ? 1 : 0);       // weight pieces of this wave per phase

    // ---- weight stream: rows srow (+128 i) of the cout tile, chunk position swizzled by the row (as in conv_ring_kernel)
    const int srow = tid / 4;
    const int pd8 = ((tid % 4) ^ ((srow / 4) & 3)) * 8;
    const long bstride = (long)128 * 9 * p.Cin;
    const T* bbase_ptr = wg;                         // + (n0 + srow) * 9 * Cin + pd8, per tile
    int bq = 0;                                      // next phase whose weights are requested
    long bcol = 0;                                   // its column offset in the packed weights: tap * Cin + chunk * 32
    int btap = 0;
    // ---- patch stream: piece k of a chunk = patch chunks k * 512 + tid (16 bytes each); its source pixel is recomputed
    //      per piece (a few dozen scalar-ish instructions, <= 5 pieces per 9 phases) instead of held in registers
    int ac = 0;                                      // chunk whose patch pieces are being requested
    long grow0 = 0;                                  // first output row of the producer's tile in the (image, row) sequence
    // (the destination is wave-uniform by construction; readfirstlane makes that provable inside the lane-masked patch pieces)
    auto dma = [&](const T* src, int off) { lds_dma16_asm(src, __builtin_amdgcn_readfirstlane(lds_u32 + (uint32_t)off)); };
    auto bslot = [&](int q) { const int s = q & 3; return s == 0 ? B0 : (s == 1 ? B1 : (s == 2 ? B2 : B3)); };
    auto init_tile = [&](long tile) {
        const long tm = tile / p.tiles_n;
        const int n0 = (int)(tile % p.tiles_n) * BN;
        bbase_ptr = wg + (long)(n0 + srow) * 9 * p.Cin + pd8;
        bq = 0;
        bcol = 0;
        btap = 0;
        ac = 0;
        grow0 = tm * R;
    };
    auto issue_a = [&](int k) {                      // piece k of chunk `ac` -> patch slot ac & 1
        const int e = k * NT + tid;
        if (e < P * 4) {
            const int pp = e >> 2, pos = e & 3;
            const int prow = pp / PW, pcol = pp - prow * PW;
            const int seg = prow / (HS + 2), ry = prow - seg * (HS + 2);
            const long g = grow0 + (long)seg * HS;
            const long n = g / p.Hout;
            const int gy = (int)(g - n * p.Hout) + ry - 1, gx = pcol - 1 + p.x_off;
            const bool ok = gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
            const T* src = ok ? xg + (((n * p.Hin + gy) * p.Win + gx) * p.Cin + ((pos ^ ((pp >> 2) & 3)) << 3) + ac * BK) : zero;
            dma(src, ((ac & 1) ? A1 : A0) + k * (NT * 16) + wid_s * 1024);
        }
    };
    auto issue_b = [&](auto ic) {                    // piece i of phase `bq` -> weight slot bq & 3
        constexpr int i = decltype(ic)::value;
        const T* src = bbase_ptr + bcol + i * bstride;
        if (i < LDB || wid_s < NT / 128) dma(src, bslot(bq) + i * (NT * 16) + wid_s * 1024);
    };
    auto next_b = [&]() {                            // phase bq + 1: next tap of the chunk, or tap 0 of the next chunk
        ++bq;
        bcol += p.Cin;
        if (++btap == 9) {
            btap = 0;
            bcol += BK - 9L * p.Cin;
        }
    };
    auto wait_vm = [&](int n) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
    };
    // fragment addressing: weights as in conv_ring_kernel; pixels through the patch
    const int swz = (col / 4) & 3;
    int koff[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) koff[kc] = ((kc * 2 + hi) ^ swz) * 16;
    const int wrow = (wn * (TN * 32) + col) * ROWB;
    int pix0[TM];                                    // patch pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int l = wm * (TM * 32) + b * 32 + col;
        const int yl = l / p.Wout, x = l - yl * p.Wout;
        const int seg = yl / HS;
        pix0[b] = (seg * (HS + 2) + (yl - seg * HS)) * PW + x;
    }

    auto prologue = [&]() {                          // patch of chunk 0 and the weights of phases 0, 1
        for (int k = 0; k < LA; ++k) issue_a(k);
        ac = 1;
        static_for<LDB + 1>([&](auto ic) { issue_b(ic); });
        next_b();
        if (nph > 1) {
            static_for<LDB + 1>([&](auto ic) { issue_b(ic); });
            next_b();
        }
    };

    long tile = tile_first;
    if (tile >= ntiles) return;
    init_tile(tile);
    prologue();
    for (;;) {
        const long m0 = (tile / p.tiles_n) * BM;
        const int n0 = (int)(tile % p.tiles_n) * BN;
        asm volatile("s_barrier" ::: "memory");      // weight slot 2 / patch slot 1 overlap the previous epilogue's LDS
        if (nph > 2) {
            static_for<LDB + 1>([&](auto ic) { issue_b(ic); });
            next_b();
        }
        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        int t = 0, ch = 0;                           // tap and chunk of the phase being computed
        int tapoff = 0, tdx = 0;                     // tap offset inside the patch: (t / 3) * PW + t % 3, kept incrementally
        int g1 = nB, g2 = nB;                        // request-group sizes of phases ph - 1, ph - 2 (phase -1 requested phase 2)
        for (int ph = 0; ph < nph; ++ph) {
            // everything up to the request group of phase ph - 3 has to be in LDS: the groups of ph - 2 and ph - 1 may fly
            if (ph == 0) wait_vm(nph > 2 ? nB : 0);
            else wait_vm(g1 + g2);
            asm volatile("s_barrier" ::: "memory");
            const bool req_b = ph + 3 < nph, req_a = t < LA && ch + 1 < nchunks;
            const char* bt = lds + bslot(ph);
            const char* at = lds + ((ch & 1) ? A1 : A0);
            u32x4 xf[KC][TM], wf[KC][TN];
            int xa[TM], xs[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int q = pix0[b] + tapoff;
                xa[b] = q * ROWB;
                xs[b] = (q >> 2) & 3;
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[0][b] = *(const u32x4*)(at + xa[b] + ((hi ^ xs[b]) << 4));
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[0][a] = *(const u32x4*)(bt + wrow + koff[0] + a * (32 * ROWB));
#pragma unroll
            for (int b = 0; b < TM; ++b) xf[1][b] = *(const u32x4*)(at + xa[b] + (((2 + hi) ^ xs[b]) << 4));
#pragma unroll
            for (int a = 0; a < TN - 2; ++a) wf[1][a] = *(const u32x4*)(bt + wrow + koff[1] + a * (32 * ROWB));
            __builtin_amdgcn_sched_barrier(0);
            if (req_a) issue_a(t);                   // piece t of the next chunk's patch (t < LA <= 5)
            __builtin_amdgcn_sched_barrier(0);
            static_for<KC * TN>([&](auto jc) {
                constexpr int j = decltype(jc)::value, kc = j / TN, a = j % TN;
                if constexpr (kc == 1 && a == 0) {
#pragma unroll
                    for (int a2 = TN - 2; a2 < TN; ++a2) wf[1][a2] = *(const u32x4*)(bt + wrow + koff[1] + a2 * (32 * ROWB));
                }
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = Elem<T>::mfma32(__builtin_bit_cast(uint4, wf[kc][a]), __builtin_bit_cast(uint4, xf[kc][b]), acc[a][b]);
                constexpr int piece = (j % 2 == 1) ? j / 2 : -1;         // weight pieces after MFMA pairs 1, 3, 5
                if constexpr (piece >= 0 && piece <= LDB) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (req_b) issue_b(std::integral_constant<int, piece>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if (req_b) next_b();
            g2 = ph == 0 ? nB : g1;
            g1 = (req_b ? nB : 0) + ((req_a && t * NT + wid_s * 64 < P * 4) ? 1 : 0);
            ++tapoff;
            if (++tdx == 3) {
                tdx = 0;
                tapoff += PW - 3;
            }
            if (++t == 9) {
                t = 0;
                tapoff = 0;
                ++ch;
                ac = ch + 1;                          // the patch pieces requested during chunk ch belong to chunk ch + 1
            }
        }
        asm volatile("s_barrier" ::: "memory");      // every wave is done reading operand slots
        const long next = tile + tile_step;
        if (next < ntiles) {
            init_tile(next);
            prologue();
        }
        int lane_e = lane, wid_e = wid_s;
        asm volatile("" : "+v"(lane_e), "+s"(wid_e));
        tile_epilogue<T, NT, TM, TN, EPI, true>(p, acc, lds + EPI_OFF, m0, n0, wid_e / WN, wid_e % WN, wid_e, lane_e);
        if (next >= ntiles) break;
        tile = next;
    }
}

// the halo kernel's tile geometry, or false when the problem does not fit it (the streaming kernels take it then)
static bool halo_geometry(ConvParams& p) {
    if (p.ntaps != 9 || p.stride != 1 || p.up || p.wrap || p.y_off != 0 || p.x_off < 0) return false;
    if (p.Cout % 320 != 0 || p.Cin % 32 != 0 || p.Cin < 64 || p.Hin != p.Hout) return false;
    if (p.Wout > 256 || 256 % p.Wout != 0 || p.M % 256 != 0) return false;
    const int R = 256 / p.Wout;
    if (!(p.Hout % R == 0 || R % p.Hout == 0)) return false;
    const int hs = R < p.Hout ? R : p.Hout;
    const int P = (R / hs) * (hs + 2) * (p.Wout + 2);
    if (P > 528) return false;
    p.halo_r = R; p.halo_seg = hs; p.halo_pw = p.Wout + 2; p.halo_p = P;
    return true;
}

template <typename T>
static int launch_halo(ConvParams p, hipStream_t stream) {
    p.tiles_n = p.Cout / 320;
    p.nblocks = (p.M / 256) * p.tiles_n;
    p.dbg = 0;
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    const long want = (p.nblocks + 7) / 8 * 8;
    const unsigned grid = (unsigned)(want < ncu ? want : ncu);
    hipLaunchKernelGGL((conv_halo_kernel<T>), dim3(grid), dim3(512), 0, stream, p);
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

#endif  // IM360_ABLATE && IM360_CONV3X3_INCLUDES_ABLATE
