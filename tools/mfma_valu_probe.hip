// Does VALU / transcendental / conversion work hide under v_mfma on ONE gfx950 SIMD, and how much of it per MFMA gap?
//
// Round 3's tools/overlap_probe.hip left the instruction order to hipcc and measured MFMA bursts followed by SLP-packed
// v_pk_fma_f32 bursts.  Here every loop body is ONE inline-asm statement: 16 v_mfma_f32_32x32x16_bf16, each
// followed by exactly N filler instructions on registers no MFMA touches -- neither the scheduler nor the SLP vectoriser can
// move or merge anything.  The disassembly of every loop is checked in next to the numbers (profiles/r04_mfma_valu_probe_isa.txt,
// written by tools/probe_isa.py from the same source).
//
// Reported per variant: shader cycles (s_memtime) per MFMA *per SIMD*, i.e. wave cycles / (16 * iterations * waves per SIMD):
// 32.0 = the matrix pipe never waits.  Variants: filler kind x fillers per gap x waves per SIMD x accumulator rotation.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mfma_valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// operands: %0-%3 accumulators, %4-%11 filler registers f0..f7, %12 A, %13 B, %14 / %15 filler constants
#define M_(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %12, %13, %" #acc "\n\t"
// one filler on register operand r (4..11)
#define FMA_(r) "v_fma_f32 %" #r ", %" #r ", %14, %15\n\t"
#define ADD_(r) "v_add_f32 %" #r ", %" #r ", %14\n\t"
#define MAX_(r) "v_max3_f32 %" #r ", %" #r ", %14, %15\n\t"
#define EXP_(r) "v_exp_f32 %" #r ", %" #r "\n\t"
#define CVT_(r) "v_cvt_pk_bf16_f32 %" #r ", %" #r ", %14\n\t"
#define MOV_(r) "v_mov_b32 %" #r ", %14\n\t"
#define NOP_(r) "s_nop 0\n\t"
// fillers per gap, registers rotating f0..f7
#define G0(F)
#define G1(F) F(4)
#define G2(F) F(4) F(5)
#define G3(F) F(4) F(5) F(6)
#define G4(F) F(4) F(5) F(6) F(7)
#define G5(F) F(4) F(5) F(6) F(7) F(8)
#define G6(F) F(4) F(5) F(6) F(7) F(8) F(9)
#define G7(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10)
#define G8(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
#define G10(F) G8(F) F(4) F(5)
#define G12(F) G8(F) F(4) F(5) F(6) F(7)
#define G16(F) G8(F) G8(F)
// the softmax mix of a flash-attention gap: exp, exp, cvt_pk, add, add (+ max3 in the wider ones)
#define MIX5 EXP_(4) ADD_(5) EXP_(6) ADD_(7) CVT_(8)
#define MIX7 EXP_(4) ADD_(5) EXP_(6) ADD_(7) CVT_(8) MAX_(9) MAX_(10)
#define MIX10 EXP_(4) ADD_(5) EXP_(6) ADD_(7) CVT_(8) MAX_(9) MAX_(10) EXP_(11) ADD_(4) CVT_(5)
// 16 MFMAs, accumulators rotating over 4 / 2 / 1 registers blocks, gap G after each
#define BODY_A4(G) M_(0) G M_(1) G M_(2) G M_(3) G M_(0) G M_(1) G M_(2) G M_(3) G M_(0) G M_(1) G M_(2) G M_(3) G M_(0) G M_(1) G M_(2) G M_(3) G
#define BODY_A2(G) M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G
#define BODY_A1(G) M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G M_(0) G
// attention-shaped: 8 MFMAs in two 4-deep accumulate chains (QK^T: two score blocks), 8 in two alternating chains (PV)
#define BODY_ATT(G) M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(0) G M_(1) G M_(2) G M_(3) G M_(2) G M_(3) G M_(2) G M_(3) G M_(2) G M_(3) G
// burst: all fillers after the 16 MFMAs (what an un-pipelined loop does)
#define BODY_BURST(G) M_(0) M_(1) M_(2) M_(3) M_(0) M_(1) M_(2) M_(3) M_(0) M_(1) M_(2) M_(3) M_(0) M_(1) M_(2) M_(3) G G G G G G G G G G G G G G G G
#define BODY_NOMFMA(G) G G G G G G G G G G G G G G G G

struct Res { unsigned long long cyc; };

#define DEF_PROBE(NAME, BODY, ACC_CONSTRAINT, PRIO)                                                                               \
    __global__ __launch_bounds__(768) void NAME(unsigned long long* out, int iters) {                                              \
        bf16x8 a, b;                                                                                                               \
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (float)((threadIdx.x + j) & 31)); b[j] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + j) & 15)); } \
        f32x16 c0, c1, c2, c3;                                                                                                     \
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }                                       \
        float f0 = -0.1f - threadIdx.x * 1e-4f, f1 = -0.2f, f2 = -0.3f, f3 = -0.4f, f4 = -0.5f, f5 = -0.6f, f6 = -0.7f, f7 = -0.8f; \
        float k1 = 0.999f, k2 = -0.0005f;                                                                                          \
        if (PRIO && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);                             \
        __syncthreads();                                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                       \
            asm volatile(BODY                                                                                                      \
                         : ACC_CONSTRAINT(c0), ACC_CONSTRAINT(c1), ACC_CONSTRAINT(c2), ACC_CONSTRAINT(c3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), \
                           "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)                                                                  \
                         : "v"(a), "v"(b), "v"(k1), "v"(k2));                                                                      \
        }                                                                                                                          \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;                                                                           \
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];                                                           \
        if (s == 12345.678f) out[0] = 1;                                                                                           \
        if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                     \
    }
#define ACC_V(x) "+v"(x)
#define ACC_A(x) "+a"(x)

// ---- packed-f32 fillers need register pairs: their own kernel shape (4 pairs + 2 constant pairs)
#define PKF_(r) "v_pk_fma_f32 %" #r ", %" #r ", %10, %11\n\t"
#define PKA_(r) "v_pk_add_f32 %" #r ", %" #r ", %10\n\t"
#define MP_(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %8, %9, %" #acc "\n\t"
#define BODYP_A4(G) MP_(0) G MP_(1) G MP_(2) G MP_(3) G MP_(0) G MP_(1) G MP_(2) G MP_(3) G MP_(0) G MP_(1) G MP_(2) G MP_(3) G MP_(0) G MP_(1) G MP_(2) G MP_(3) G
#define DEF_PROBE_PK(NAME, BODY)                                                                                                   \
    __global__ __launch_bounds__(768) void NAME(unsigned long long* out, int iters) {                                              \
        bf16x8 a, b;                                                                                                               \
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (float)((threadIdx.x + j) & 31)); b[j] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + j) & 15)); } \
        f32x16 c0, c1, c2, c3;                                                                                                     \
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }                                       \
        f32x2 p0 = {-0.1f - threadIdx.x * 1e-4f, -0.2f}, p1 = {-0.3f, -0.4f}, p2 = {-0.5f, -0.6f}, p3 = {-0.7f, -0.8f};            \
        f32x2 k1 = {0.999f, 0.998f}, k2 = {-0.0005f, -0.0004f};                                                                    \
        __syncthreads();                                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                       \
            asm volatile(BODY : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)                     \
                         : "v"(a), "v"(b), "v"(k1), "v"(k2));                                                                      \
        }                                                                                                                          \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;                                                           \
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];                                                           \
        if (s == 12345.678f) out[0] = 1;                                                                                           \
        if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                     \
    }
#define GP1(F) F(4)
#define GP2(F) F(4) F(5)
#define GP4(F) F(4) F(5) F(6) F(7)

// ---- the kernels
DEF_PROBE(p_mfma_a4, BODY_A4(G0(FMA_)), ACC_V, 0)
DEF_PROBE(p_mfma_a2, BODY_A2(G0(FMA_)), ACC_V, 0)
DEF_PROBE(p_mfma_a1, BODY_A1(G0(FMA_)), ACC_V, 0)
DEF_PROBE(p_mfma_att, BODY_ATT(G0(FMA_)), ACC_V, 0)
DEF_PROBE(p_mfma_a4_agpr, BODY_A4(G0(FMA_)), ACC_A, 0)
#define FILLSET(K, F)                                      \
    DEF_PROBE(p_##K##_1, BODY_A4(G1(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_2, BODY_A4(G2(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_3, BODY_A4(G3(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_4, BODY_A4(G4(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_5, BODY_A4(G5(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_6, BODY_A4(G6(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_7, BODY_A4(G7(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_8, BODY_A4(G8(F)), ACC_V, 0)         \
    DEF_PROBE(p_##K##_12, BODY_A4(G12(F)), ACC_V, 0)       \
    DEF_PROBE(p_##K##_only8, BODY_NOMFMA(G8(F)), ACC_V, 0)
FILLSET(fma, FMA_)
FILLSET(add, ADD_)
FILLSET(exp, EXP_)
FILLSET(cvt, CVT_)
FILLSET(max3, MAX_)
FILLSET(mov, MOV_)
FILLSET(nop, NOP_)
DEF_PROBE(p_mix_5, BODY_A4(MIX5), ACC_V, 0)
DEF_PROBE(p_mix_7, BODY_A4(MIX7), ACC_V, 0)
DEF_PROBE(p_mix_10, BODY_A4(MIX10), ACC_V, 0)
DEF_PROBE(p_mix_5_att, BODY_ATT(MIX5), ACC_V, 0)
DEF_PROBE(p_mix_7_att, BODY_ATT(MIX7), ACC_V, 0)
DEF_PROBE(p_mix_10_att, BODY_ATT(MIX10), ACC_V, 0)
DEF_PROBE(p_mix_5_agpr, BODY_A4(MIX5), ACC_A, 0)
DEF_PROBE(p_mix_7_prio, BODY_A4(MIX7), ACC_V, 1)
DEF_PROBE(p_mix_10_prio, BODY_A4(MIX10), ACC_V, 1)
DEF_PROBE(p_mix_5_burst, BODY_BURST(MIX5), ACC_V, 0)
DEF_PROBE(p_mix_10_burst, BODY_BURST(MIX10), ACC_V, 0)
DEF_PROBE(p_mix_5_only, BODY_NOMFMA(MIX5), ACC_V, 0)
DEF_PROBE(p_mix_10_only, BODY_NOMFMA(MIX10), ACC_V, 0)
DEF_PROBE(p_fma_5_a2, BODY_A2(G5(FMA_)), ACC_V, 0)
DEF_PROBE(p_fma_5_a1, BODY_A1(G5(FMA_)), ACC_V, 0)
DEF_PROBE(p_fma_8_burst, BODY_BURST(G8(FMA_)), ACC_V, 0)
DEF_PROBE_PK(p_pkfma_1, BODYP_A4(GP1(PKF_)))
DEF_PROBE_PK(p_pkfma_2, BODYP_A4(GP2(PKF_)))
DEF_PROBE_PK(p_pkfma_4, BODYP_A4(GP4(PKF_)))
DEF_PROBE_PK(p_pkadd_2, BODYP_A4(GP2(PKA_)))
DEF_PROBE_PK(p_pkadd_4, BODYP_A4(GP4(PKA_)))

typedef void (*kern_t)(unsigned long long*, int);
struct Variant { const char* name; kern_t k; int nfill; bool has_mfma; };
#define V(name, n, m) {#name, name, n, m}
#define VSET(K) V(p_##K##_1, 1, true), V(p_##K##_2, 2, true), V(p_##K##_3, 3, true), V(p_##K##_4, 4, true), V(p_##K##_5, 5, true), V(p_##K##_6, 6, true), \
                V(p_##K##_7, 7, true), V(p_##K##_8, 8, true), V(p_##K##_12, 12, true), V(p_##K##_only8, 8, false)
static Variant variants[] = {
    V(p_mfma_a4, 0, true), V(p_mfma_a2, 0, true), V(p_mfma_a1, 0, true), V(p_mfma_att, 0, true), V(p_mfma_a4_agpr, 0, true),
    VSET(fma), VSET(add), VSET(exp), VSET(cvt), VSET(max3), VSET(mov), VSET(nop),
    V(p_mix_5, 5, true), V(p_mix_7, 7, true), V(p_mix_10, 10, true), V(p_mix_5_att, 5, true), V(p_mix_7_att, 7, true), V(p_mix_10_att, 10, true),
    V(p_mix_5_agpr, 5, true), V(p_mix_7_prio, 7, true), V(p_mix_10_prio, 10, true), V(p_mix_5_burst, 5, true), V(p_mix_10_burst, 10, true),
    V(p_mix_5_only, 5, false), V(p_mix_10_only, 10, false),
    V(p_fma_5_a2, 5, true), V(p_fma_5_a1, 5, true), V(p_fma_8_burst, 8, true),
    V(p_pkfma_1, 1, true), V(p_pkfma_2, 2, true), V(p_pkfma_4, 4, true), V(p_pkadd_2, 2, true), V(p_pkadd_4, 4, true),
};

int main(int argc, char** argv) {
    const int iters = 2000;
    const int grids[2] = {16, 256};
    unsigned long long* d;
    hipMalloc(&d, (1 + 256 * 16) * sizeof(unsigned long long));
    std::vector<unsigned long long> h(1 + 256 * 16);
    printf("# cycles per MFMA per SIMD (s_memtime; 32.0 = matrix pipe saturated); wall ns per 16-MFMA iteration in brackets\n");
    printf("# columns: 1 / 2 / 3 waves per SIMD on 16 workgroups (no power cap), then 1 / 2 waves per SIMD on 256 workgroups\n");
    printf("%-18s %5s | %-16s %-16s %-16s | %-16s %-16s\n", "variant", "fill", "16wg x1", "16wg x2", "16wg x3", "256wg x1", "256wg x2");
    for (const Variant& v : variants) {
        printf("%-18s %5d |", v.name, v.nfill);
        for (int gi = 0; gi < 2; ++gi) {
            for (int wps = 1; wps <= (gi == 0 ? 3 : 2); ++wps) {
                const int grid = grids[gi], threads = 256 * wps;
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(threads), 0, 0, d, 50);
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::vector<double> c;
                for (int b = 0; b < grid; ++b)
                    for (int w = 0; w < 4 * wps; ++w) c.push_back((double)h[1 + b * 16 + w]);
                std::sort(c.begin(), c.end());
                const double med = c[c.size() / 2];
                printf(" %6.1f [%6.0f]  ", med / (16.0 * iters * wps), ms * 1e6 / iters);
                if (gi == 0 && wps == 3) printf("|");
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
