// Software-pipelined flash attention for head dim 64 without a bias (the UNets' spatial self-attention,
// diffusers/models/attention_processor.py:1195-1283 through animatediff/models/attention.py:230-297): the same mathematics and
// data layout as attn_fwd_kernel (attn_fwd.hip: S^T = K Q^T, running max as the MFMA C operand, deferred rescale, V through the
// transposing LDS read), with the instruction ORDER taken away from the compiler.
//
// Why (tools/mfma_valu_probe.hip, profiles/r04_mfma_valu_probe.txt): while a v_mfma_f32_32x32x16 occupies
// a SIMD's matrix pipe (32 cycles) that SIMD issues at most one other VALU instruction per 4 cycles (8 for v_exp_f32) -- from
// whichever wave -- so ~28 cycles of vector work hide under every MFMA *if it is issued there*; vector work issued while the
// pipe idles is pure loss, and a second or third wave per SIMD does not recover it (MFMA bursts followed by VALU bursts cost
// 60 cycles per MFMA at one wave per SIMD, 48 at two; the same instructions interleaved 38).  attn_fwd_kernel's loop is what
// hipcc makes of the source: 8 QK^T MFMAs back to back, 16 v_max3 with the pipe idle, then 80 VALU around 8 PV MFMAs.
// Here tile t's softmax (32 exp2, 32 row-sum adds, 16 pack conversions per lane) is issued between the QK^T MFMAs of tile
// t + 1 and its own PV MFMAs, and the row-max reduction of tile t + 1 under the last PV MFMAs -- every instruction of the loop
// is one `asm volatile` statement (hipcc keeps their order and still allocates the registers), LDS fragment reads are issued
// two MFMA slots ahead and waited for with counted lgkmcnt.  The schedule (vector ops per MFMA slot) is a template parameter.
//
// Differences from attn_fwd_kernel that change bits: none in the arithmetic (same MFMA order per accumulator, same exp2 / pack
// rounding); the row sums are accumulated in two interleaved fp32 chains instead of one, so results agree to fp32 rounding of
// the normaliser (tests/test_kernels_gpu.py::test_attention_pipelined_kernel compares both with the fp32 oracle).
#include "common.h"
#include "attn_params.h"

namespace im360 {
namespace {

constexpr int KVB = 64, D = 64, KP = D + 8, VP = 96, DC = D / 16, DV = D / 32;
constexpr int KT = KVB * KP, VT = KVB * VP;          // tile sizes (elements)
constexpr float PIPE_RESCALE_THR = 5.0f;             // log2 units, as attn_fwd_kernel

// ---- one instruction per statement ------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void mfma_new(f32x16& d, const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
template <typename T> __device__ __forceinline__ void mfma_acc(f32x16& d, const u32x4& a, const u32x4& b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
template <typename T> __device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
    uint32_t r;
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (macros: an ext-vector element cannot bind to a reference, but it can be an asm operand -- hipcc then uses the sub-register)
#define IM360_V_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define IM360_V_ADD(acc, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
#define IM360_V_MAX3_NEW(m, a, b, c) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c))
#define IM360_V_MAX3(m, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b))
template <int OFF> __device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read_tr64(u32x2& dst, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
// (the builtin, not an asm statement: hipcc pads an asm statement that reads a register DEFINED BY ANOTHER asm statement with no
//  compiler-visible instruction in between -- it cannot know what the first one was -- and the wait is such an instruction:
//  26 -> 12 s_nop per tile.  Encoding: vmcnt = 63 and expcnt = 7 (no wait), lgkmcnt = N.)
template <int N> __device__ __forceinline__ void wait_lgkm() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }

// ---- schedules: vector micro-ops issued after each of the 16 MFMA slots of a tile (softmax stream of 80, max stream of 16) ----
// Constraints (checked at compile time): chunk q's 20 softmax ops precede PV slot 8 + 2 q; no max op before slot 10 (the
// scores it reads left the matrix pipe with slot 7: >= 12 wait states).
struct Sched { int val[16]; int mx[16]; };
constexpr Sched SCHEDS[] = {
    {{7, 7, 6, 7, 7, 6, 7, 7, 6, 7, 7, 6, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 4, 4}},
    {{6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 2, 4, 4}},
    {{8, 8, 8, 8, 8, 8, 8, 8, 4, 4, 4, 4, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 3, 3, 3, 3}},
    {{10, 10, 10, 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 3, 3, 3, 3}},
};
constexpr int NSCHED = sizeof(SCHEDS) / sizeof(SCHEDS[0]);
constexpr bool sched_ok(const Sched& s) {
    int cum = 0, cm = 0;
    for (int i = 0; i < 16; ++i) {
        if (i >= 8 && (i - 8) % 2 == 0 && cum < 20 * ((i - 8) / 2 + 1)) return false;
        if (i < 10 && s.mx[i] != 0) return false;
        cum += s.val[i];
        cm += s.mx[i];
    }
    return cum == 80 && cm == 16;
}
constexpr int sched_prefix(const int* a, int n) { int c = 0; for (int i = 0; i < n; ++i) c += a[i]; return c; }
static_assert(sched_ok(SCHEDS[0]) && sched_ok(SCHEDS[1]) && sched_ok(SCHEDS[2]) && sched_ok(SCHEDS[3]), "schedule violates a dependence");

// order of the 20 softmax micro-ops of one 8-score chunk: E = exp2 in place, A = add into a row-sum chain, C = pack two
// weights.  No op directly follows the producer of one of its inputs (a transcendental result needs a wait state).
enum { OP_E = 0, OP_A = 1, OP_C = 2 };
struct MicroOp { int kind, arg; };
constexpr MicroOp CHUNK_OPS[20] = {{OP_E, 0}, {OP_E, 1}, {OP_E, 2}, {OP_A, 0}, {OP_E, 3}, {OP_A, 1}, {OP_E, 4}, {OP_A, 2}, {OP_C, 0}, {OP_E, 5},
                                   {OP_A, 3}, {OP_E, 6}, {OP_A, 4}, {OP_C, 1}, {OP_E, 7}, {OP_A, 5}, {OP_A, 6}, {OP_C, 2}, {OP_A, 7}, {OP_C, 3}};
// the 16 max micro-ops: three independent chains (a dependent v_max3 directly behind its producer costs a wait state) over the
// 32 scores v = sn[v / 16][v % 16]; op 15 joins the chains.  {chain, first value, values taken (3 = the chain's first op)}
struct MaxOp { int chain, v0, n; };
constexpr MaxOp MAX_OPS[15] = {{0, 0, 3}, {1, 3, 3}, {2, 6, 3}, {0, 9, 2}, {1, 11, 2}, {2, 13, 2}, {0, 15, 2}, {1, 17, 2}, {2, 19, 2}, {0, 21, 2},
                               {1, 23, 2}, {2, 25, 2}, {0, 27, 2}, {1, 29, 2}, {2, 30, 2}};        // (the last one takes value 30 again with 31)

template <typename T, int NW, int SCHED, int AHEAD = 2, int ABL = 0, int WPE = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_pipe_kernel(AttnParams p) {
    constexpr int NT = NW * 64;
    constexpr int CH = KVB * D / 8;        // 16-byte chunks in a K (or V) tile
    constexpr int NLT = NT < CH ? NT : CH; // threads that stage (all of them up to eight waves; the first eight of twelve)
    constexpr int CLD = CH / NLT;          // chunks per staging thread and operand
    constexpr int RSTEP = NLT / (D / 8);   // tile rows between a thread's consecutive chunks
    static_assert(CH % NLT == 0, "staging pattern");
    // ONE LDS object (a second one makes hipcc drain vmcnt in front of every LDS read): [K tile 0 | K tile 1 | V tile 0 | V tile 1]
    __shared__ __attribute__((aligned(16))) T lds[2 * KT + 2 * VT];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    // XCD-aware block order, (batch * head) major, query tile minor (as attn_fwd_kernel: one head's K / V stay in one XCD's L2)
    long lb = blockIdx.x;
    {
        const long nb = gridDim.x, qn = nb / 8, rn = nb % 8, xcd = lb % 8, idx = lb / 8;
        lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    }
    const int bh = (int)(lb / p.nqt);
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = (int)(lb % p.nqt) * (32 * NW) + wid * 32;
    const T* qb_ = (const T*)p.q + (long)b * p.q_bs + (long)h * D;
    const T* kb_ = (const T*)p.k + (long)(b / p.kv_group) * p.k_bs + (long)h * D;
    const T* vb_ = (const T*)p.v + (long)(b / p.kv_group) * p.v_bs + (long)h * D;
    const int ntiles = p.Nk / KVB;          // (Nk % 64 == 0: checked by the launcher)

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 dc + 8 hi .. + 7] * scale * log2(e)
    int qrow = q0 + col;
    const bool q_valid = qrow < p.Nq;
    if (!q_valid) qrow = p.Nq - 1;
    u32x4 qf[DC];
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
        float f[8];
        unpack8<T>(*(const uint4*)(qb_ + (long)qrow * p.q_rs + dc * 16 + hi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= p.scale_log2;
        qf[dc] = __builtin_bit_cast(u32x4, pack8<T>(f));
    }
    f32x16 o[DV], negm, sA[2], sB[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_sc = 0.f, lsum0 = 0.f, lsum1 = 0.f;

    // ---- staging: thread tid owns (row, 16-byte chunk) slots tid + i * NT of a tile
    // global -> registers through buffer loads: one wave-uniform descriptor per operand (base of this (batch, head)'s rows), the
    // tile as a scalar offset, the thread's (row, chunk) as ONE 32-bit VGPR offset -- no 64-bit address arithmetic in the loop
    const int srow = tid / (D / 8), sc8 = tid % (D / 8);
    const bool loader = NLT == NT || __builtin_amdgcn_readfirstlane(wid) < NLT / 64;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kb_, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vb_, 0, 0x7fffffff, 0x00020000);
    const uint32_t kvoff = (uint32_t)((long)srow * p.k_rs + sc8 * 8) * 2u, vvoff = (uint32_t)((long)srow * p.v_rs + sc8 * 8) * 2u;
    const uint32_t k_tile_b = (uint32_t)(p.k_rs * KVB * 2), v_tile_b = (uint32_t)(p.v_rs * KVB * 2);        // bytes per tile
    const uint32_t k_step_b = (uint32_t)(p.k_rs * RSTEP * 2), v_step_b = (uint32_t)(p.v_rs * RSTEP * 2);
    u32x4 kreg[CLD], vreg[CLD];
    auto load_k = [&](int t) {
        if (!loader) return;
        const uint32_t so = (uint32_t)(t < ntiles ? t : ntiles - 1) * k_tile_b;          // (past the end: the last tile again, never read)
#pragma unroll
        for (int i = 0; i < CLD; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff, so + i * k_step_b, 0);
    };
    auto load_v = [&](int t) {
        if (!loader) return;
        const uint32_t so = (uint32_t)(t < ntiles ? t : ntiles - 1) * v_tile_b;
#pragma unroll
        for (int i = 0; i < CLD; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff, so + i * v_step_b, 0);
    };
    T* const kdst = lds + srow * KP + sc8 * 8;
    T* const vdst = lds + 2 * KT + srow * VP + sc8 * 8;
    auto store_k = [&](int buf) {
        if (!loader) return;
#pragma unroll
        for (int i = 0; i < CLD; ++i) *(u32x4*)(kdst + buf * KT + i * RSTEP * KP) = kreg[i];
    };
    auto store_v = [&](int buf) {
        if (!loader) return;
#pragma unroll
        for (int i = 0; i < CLD; ++i) *(u32x4*)(vdst + buf * VT + i * RSTEP * VP) = vreg[i];
    };

    // fragment read bases (LDS byte addresses); everything else is an immediate offset
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int l16 = lane & 15, half = (lane >> 4) & 1;
    const uint32_t kaddr = (ABL & 1024) ? lds0 + lane * 16 : lds0 + (uint32_t)(col * KP + hi * 8) * 2u;
    const uint32_t vaddr = (ABL & 1024) ? lds0 + lane * 8 : lds0 + (uint32_t)(2 * KT) * 2u + (uint32_t)((4 * hi + (l16 >> 2)) * VP + 16 * half + 4 * (l16 & 3)) * 2u;

    // fragment rings: the read of slot s is issued AHEAD slots before its MFMA, into the buffer the MFMA of slot s - RING used
    // ABL (ablation builds only; results are garbage): 1 no exp2, 2 no QK^T MFMAs, 4 no PV MFMAs, 8 no fragment reads, 16 no barrier,
    // 32 no staging, 64 no add / pack / max
    constexpr int RING = AHEAD + 1;           // (the MFMA that last read a buffer was issued a slot before the buffer's next read is)
    u32x4 kfr[RING];
    u32x2 vlo[RING], vhi[RING];
    // (the packed weights of an 8-score chunk -- the B operand of its two PV MFMAs -- overwrite the chunk's first four score
    // registers: no separate P registers)

    using std::integral_constant;
    // QK^T MFMA i of a tile: k-step dc = i / 2 of half kb = i % 2 (the two score blocks' chains alternate)
    auto rd_k = [&](auto ic, auto bufc) {
        constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value;
        if constexpr ((ABL & (8 | 256)) == 0) lds_read128<(buf * KT + (i % 2) * 32 * KP + (i / 2) * 16) * 2>(kfr[i % RING], kaddr);
    };
    // PV MFMA j of a tile: chunk q = j / 2 (keys 32 (q / 2) + 16 (q % 2) + ...), output channel block dvb = j % 2
    auto rd_v = [&](auto jc, auto bufc) {
        constexpr int j = decltype(jc)::value, buf = decltype(bufc)::value;
        constexpr int q = j / 2, dvb = j % 2;
        constexpr int off = (buf * VT + ((q / 2) * 32 + 16 * (q % 2)) * VP + dvb * 32) * 2;
        if constexpr ((ABL & (8 | 512)) == 0) {
            lds_read_tr64<off>(vlo[j % RING], vaddr);
            lds_read_tr64<off + 8 * VP * 2>(vhi[j % RING], vaddr);
        }
    };
    auto mm_qk = [&](auto ic, f32x16 (&sn)[2]) {
        constexpr int i = decltype(ic)::value;
        if constexpr ((ABL & 2) != 0) return;
        if constexpr (i / 2 == 0) mfma_new<T>(sn[i % 2], kfr[i % RING], qf[0], negm);
        else mfma_acc<T>(sn[i % 2], kfr[i % RING], qf[i / 2]);
    };
    auto mm_pv = [&](auto jc, f32x16 (&sc)[2]) {
        constexpr int j = decltype(jc)::value, q = j / 2;
        if constexpr ((ABL & 4) != 0) return;
        const u32x4 a = {vlo[j % RING].x, vlo[j % RING].y, vhi[j % RING].x, vhi[j % RING].y};
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        f32x4v pw;
        if constexpr (q % 2 == 0) pw = __builtin_shufflevector(sc[q / 2], sc[q / 2], 0, 1, 2, 3);
        else pw = __builtin_shufflevector(sc[q / 2], sc[q / 2], 8, 9, 10, 11);
        mfma_acc<T>(o[j % 2], a, __builtin_bit_cast(u32x4, pw));
    };
    // softmax micro-op n (0 .. 79) on the current tile's scores (exponentiated in place)
    auto vop = [&](auto nc, f32x16 (&sc)[2]) {
        constexpr int n = decltype(nc)::value, q = n / 20, kind = CHUNK_OPS[n % 20].kind, arg = CHUNK_OPS[n % 20].arg;
        constexpr int kb = q / 2, r0 = 8 * (q % 2);
        if constexpr ((ABL & 1) != 0 && kind == OP_E) return;
        if constexpr ((ABL & 64) != 0 && kind != OP_E) return;
        if constexpr (kind == OP_E) {
            IM360_V_EXP(sc[kb][r0 + arg]);
        } else if constexpr (kind == OP_A) {
            if constexpr (arg % 2 == 0) IM360_V_ADD(lsum0, sc[kb][r0 + arg]);
            else IM360_V_ADD(lsum1, sc[kb][r0 + arg]);
        } else {
            // in place: packed word `arg` of the chunk goes into score register r0 + arg (whose own value has been added to the row
            // sum and packed by now: CHUNK_OPS order)
            if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(sc[kb][r0 + arg]) : "v"(sc[kb][r0 + 2 * arg]), "v"(sc[kb][r0 + 2 * arg + 1]));
            else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(sc[kb][r0 + arg]) : "v"(sc[kb][r0 + 2 * arg]), "v"(sc[kb][r0 + 2 * arg + 1]));
        }
    };
    // max micro-op n (0 .. 15) over the 32 scores of the next tile
    float mloc, mch[3];
    auto mop = [&](auto nc, f32x16 (&sn)[2]) {
        constexpr int n = decltype(nc)::value;
        if constexpr ((ABL & 64) != 0) { mloc = 0.f; return; }
        if constexpr (n == 15) {
            IM360_V_MAX3_NEW(mloc, mch[0], mch[1], mch[2]);
        } else {
            constexpr MaxOp M = MAX_OPS[n];
            if constexpr (M.n == 3) IM360_V_MAX3_NEW(mch[M.chain], sn[M.v0 / 16][M.v0 % 16], sn[(M.v0 + 1) / 16][(M.v0 + 1) % 16], sn[(M.v0 + 2) / 16][(M.v0 + 2) % 16]);
            else IM360_V_MAX3(mch[M.chain], sn[M.v0 / 16][M.v0 % 16], sn[(M.v0 + 1) / 16][(M.v0 + 1) % 16]);
        }
    };
    // the row max of the two half-waves, then the (rare) move of the running max; `first` sets it unconditionally
    auto decide = [&](f32x16 (&sn)[2], bool first) {
        float ma, mb;
        half_wave_pair(mloc, ma, mb);
        float mx;
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(ma), "v"(mb));         // (fmaxf would first canonicalise both inputs: two more v_max)
        if (first || __any(mx > PIPE_RESCALE_THR)) {
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // the last PV MFMAs' results: out of the matrix pipe before a VALU reads them
            const float delta = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            m_sc += delta;
            lsum0 *= alpha;
            lsum1 *= alpha;
#pragma unroll
            for (int i = 0; i < DV; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_sc;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sn[kb][r] -= delta;
        }
    };

    // One tile step.  PAR = t & 1.  On entry: sc = scores of tile t minus the running max, K(t + 1) in K buffer PAR ^ 1, V(t) in V
    // buffer PAR (visible after the barrier), kreg = K(t + 2), vreg = V(t + 1).  NEXT: tile t + 1 exists (its QK^T runs here).
    auto step = [&](int t, auto parc, auto nextc, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
        constexpr int PAR = decltype(parc)::value;
        constexpr bool NEXT = decltype(nextc)::value;
        using KB = integral_constant<int, PAR ^ 1>;
        using VB = integral_constant<int, PAR>;
        if constexpr ((ABL & 16) == 0) __syncthreads();
        // staging writes first (older than every fragment read of the step: the counted waits below need not know about them)
        if constexpr (NEXT && (ABL & 32) == 0) {
            store_k(PAR);            // K(t + 2) over K(t) (read during step t - 1)
            store_v(PAR ^ 1);        // V(t + 1) over V(t - 1) (read during step t - 1)
        }
        // fragment reads of the first AHEAD slots
        static_for<AHEAD>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            if constexpr (NEXT) {
                if constexpr (a < 8) rd_k(integral_constant<int, a>{}, KB{});
                else rd_v(integral_constant<int, a - 8>{}, VB{});
            } else {
                rd_v(integral_constant<int, a>{}, VB{});
            }
        });
        if constexpr (NEXT && (ABL & 32) == 0) {
            load_k(t + 3);
            load_v(t + 2);
        }
        if constexpr (NEXT) {
            constexpr const Sched& S = SCHEDS[SCHED];
            static_for<16>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                // fragments of slot s are in; younger LDS operations: the reads of slots s + 1 .. s + AHEAD - 1 (the
                // staging writes are older)
                constexpr int younger = [] { int c = 0; for (int k = 1; k < AHEAD; ++k) c += (s + k >= 16 ? 0 : (s + k < 8 ? 1 : 2)); return c; }();
                constexpr int v0 = sched_prefix(S.val, s), m0 = sched_prefix(S.mx, s);
                // slot 0's vector work goes IN FRONT of its MFMA: it needs nothing from LDS, and the step's first fragment reads
                // (issued right behind the barrier and the staging writes) are still in flight
                if constexpr (s == 0) static_for<S.val[s]>([&](auto k) { vop(integral_constant<int, v0 + decltype(k)::value>{}, sc); });
                if constexpr ((ABL & (8 | 128)) == 0) wait_lgkm<younger>();
                if constexpr (s < 8) mm_qk(integral_constant<int, s>{}, sn);
                else mm_pv(integral_constant<int, s - 8>{}, sc);
                if constexpr (s + AHEAD < 8) rd_k(integral_constant<int, s + AHEAD>{}, KB{});
                else if constexpr (s + AHEAD < 16) rd_v(integral_constant<int, s + AHEAD - 8>{}, VB{});
                if constexpr (s != 0) static_for<S.val[s]>([&](auto k) { vop(integral_constant<int, v0 + decltype(k)::value>{}, sc); });
                static_for<S.mx[s]>([&](auto k) { mop(integral_constant<int, m0 + decltype(k)::value>{}, sn); });
            });
            decide(sn, false);
        } else {
            // last tile: chunk 0's softmax first, then eight PV slots with the remaining chunks between them
            static_for<20>([&](auto k) { vop(k, sc); });
            static_for<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int younger = [] { int c = 0; for (int k = 1; k < AHEAD; ++k) c += (j + k >= 8 ? 0 : 2); return c; }();
                if constexpr ((ABL & (8 | 128)) == 0) wait_lgkm<younger>();
                mm_pv(jc, sc);
                if constexpr (j + AHEAD < 8) rd_v(integral_constant<int, j + AHEAD>{}, VB{});
                if constexpr (j < 6) static_for<10>([&](auto k) { vop(integral_constant<int, 20 + 10 * j + decltype(k)::value>{}, sc); });
            });
            // hipcc does not know these statements are MFMAs: whatever it places behind the last one (register moves where the
            // two instantiations of this step join, the epilogue) must find the results out of the matrix pipe
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        }
    };

    // ---- prologue: K(0), V(0), K(1) into LDS, K(2) / V(1) into the staging registers, scores of tile 0
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    load_k(1);
    store_k(1);
    load_k(2);
    load_v(1);
    __syncthreads();
    static_for<8>([&](auto ic) {
        rd_k(ic, integral_constant<int, 0>{});
        wait_lgkm<0>();
        mm_qk(ic, sA);
    });
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    static_for<16>([&](auto k) { mop(k, sA); });
    decide(sA, true);

    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        step(t, I0{}, std::true_type{}, sA, sB);
        step(t + 1, I1{}, std::true_type{}, sB, sA);
    }
    if (ntiles - t == 2) {
        step(t, I0{}, std::true_type{}, sA, sB);
        step(t + 1, I1{}, std::false_type{}, sB, sA);
    } else {
        step(t, I0{}, std::false_type{}, sA, sB);
    }

    // ---- epilogue: normalise, optional accumulate, store 4 consecutive channels per (lane, group)
    const float l_run = lsum0 + lsum1;
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

template <typename T, int NW>
int launch_nw(const AttnParams& p, hipStream_t stream, int sched, int ahead, int abl) {
    AttnParams q = p;
    q.nqt = (q.Nq + 32 * NW - 1) / (32 * NW);
    const long nblk = (long)q.B * q.H * q.nqt;
    if (nblk > 0x7fffffffL) {
        im360_set_error("attn_pipe: %ld workgroups exceed the grid limit", nblk);
        return IM360_ERR_ARG;
    }
    dim3 grid((unsigned)nblk, 1, 1), block(NW * 64);
#define IM360_PIPE_LAUNCH(S, A, B) hipLaunchKernelGGL((attn_pipe_kernel<T, NW, S, A, B>), grid, block, 0, stream, q)
#ifdef IM360_ABLATE
    if (abl && NW != 12) {
        switch (abl) {
            case 1: IM360_PIPE_LAUNCH(0, 2, 65 + 32 + 16 + 256); break;      // MFMA + V reads + waits only
            case 2: IM360_PIPE_LAUNCH(0, 2, 65 + 32 + 16 + 512); break;      // MFMA + K reads + waits only
            case 3: IM360_PIPE_LAUNCH(0, 2, 65 + 32 + 16 + 1024); break;     // MFMA + all reads at lane-linear addresses
            case 4: IM360_PIPE_LAUNCH(0, 2, 1024); break;                    // everything, reads at lane-linear addresses
            case 5: IM360_PIPE_LAUNCH(0, 2, 65 + 32 + 16 + 8); break;        // MFMA only
            case 6: IM360_PIPE_LAUNCH(0, 2, 6 + 32 + 16); break;      // VALU + reads + waits only
            case 7: IM360_PIPE_LAUNCH(0, 2, 128); break;              // reads issued, never waited for
            case 8: IM360_PIPE_LAUNCH(0, 2, 128 + 32); break;         // ... and no staging
            case 9: IM360_PIPE_LAUNCH(0, 2, 32 + 16); break;          // reads and waits, no staging, no barrier
            case 10: IM360_PIPE_LAUNCH(0, 2, 65); break;
            case 11: IM360_PIPE_LAUNCH(0, 2, 65 + 32 + 16); break;    // MFMA + reads + waits only
            default: IM360_PIPE_LAUNCH(0, 2, 65 + 128 + 32 + 16); break;   // MFMA + reads, no waits
        }
        IM360_CHECK_LAUNCH();
        return IM360_OK;
    }
#endif
    if constexpr (NW == 12) {
        // twelve waves = three per SIMD in ONE workgroup per CU (a K / V tile staged once for 384 query rows)
        if (ahead == 3) hipLaunchKernelGGL((attn_pipe_kernel<T, NW, 0, 3, 0, 3>), grid, block, 0, stream, q);
        else if (sched == 1) hipLaunchKernelGGL((attn_pipe_kernel<T, NW, 1, 2, 0, 3>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((attn_pipe_kernel<T, NW, 0, 2, 0, 3>), grid, block, 0, stream, q);
    } else if (ahead == 3) {
        switch (sched) {
            case 1: IM360_PIPE_LAUNCH(1, 3, 0); break;
            default: IM360_PIPE_LAUNCH(0, 3, 0); break;
        }
    } else if (ahead == 4) {
        IM360_PIPE_LAUNCH(0, 4, 0);
    } else {
        switch (sched) {
            case 1: IM360_PIPE_LAUNCH(1, 2, 0); break;
            case 2: IM360_PIPE_LAUNCH(2, 2, 0); break;
            case 3: IM360_PIPE_LAUNCH(3, 2, 0); break;
            default: IM360_PIPE_LAUNCH(0, 2, 0); break;
        }
    }
#undef IM360_PIPE_LAUNCH
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

}  // namespace

// Shapes taken: head dim 64 (the caller's template), no bias, one key / value set, Nk a multiple of 64 and >= 128, Nq >= 128.
// Knob attn_pipe: 0 = off; otherwise bits 0-2 = 1 + schedule, bits 3 / 6 = eight- / twelve-wave workgroups (default four),
// bits 4-5 = read-ahead - 2, bits 8-11 = ablation build (IM360_ABLATE only).
int launch_attn_pipe(const AttnParams& p, int dtype, hipStream_t stream) {
    int kb = knob(KNOB_ATTN_PIPE);
    // -1 (default): the rule measured on MI355X (profiles/r04_attn_pipe.log) -- eight-wave workgroups, schedule 1, for long
    // sequences (the panorama branch's levels 0 / 1: -8 % / -3 %); shorter ones (<= 1024 keys: 16 tiles, where a workgroup's
    // prologue and the last tile's drain weigh more) stay on attn_fwd_kernel's three waves per SIMD
    if (kb < 0) kb = (p.Nq >= 2048 && p.Nk >= 2048) ? 10 : 0;
    if (kb <= 0 || (kb & 7) == 0 || p.bias || p.k2 || (p.Nk % KVB) != 0 || p.Nk < 2 * KVB || p.Nq < 128) return 1;
    const int sched = (kb & 7) - 1, nw = (kb & 64) ? 12 : ((kb & 8) ? 8 : 4), ahead = 2 + ((kb >> 4) & 3), abl = (kb >> 8) & 15;
    if (sched >= NSCHED || ahead > 4) return 1;
    if (dtype == 0) {
        if (nw == 12) return launch_nw<__bf16, 12>(p, stream, sched, ahead, abl);
        return nw == 8 ? launch_nw<__bf16, 8>(p, stream, sched, ahead, abl) : launch_nw<__bf16, 4>(p, stream, sched, ahead, abl);
    }
    if (nw == 12) return launch_nw<_Float16, 12>(p, stream, sched, ahead, abl);
    return nw == 8 ? launch_nw<_Float16, 8>(p, stream, sched, ahead, abl) : launch_nw<_Float16, 4>(p, stream, sched, ahead, abl);
}

}  // namespace im360
