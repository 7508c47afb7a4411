// What does an LDS fragment read cost next to MFMAs on a gfx950 SIMD?  Companion of tools/mfma_valu_probe.hip: every loop body
// is one inline-asm statement -- 16 v_mfma_f32_32x32x16_bf16, each followed by R LDS reads (ds_read_b128 or ds_read_b64_tr_b16,
// conflict-free lane-linear addresses) and optionally the 5-op softmax mix; one s_waitcnt lgkmcnt(0) per 4 MFMAs (reads are
// consumed a quarter of a loop later, as in a pipelined kernel).  Reports wall ns per 16-MFMA iteration and shader cycles per
// MFMA per SIMD for 1 / 2 / 3 waves per SIMD, on 16 and on 256 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// operands: %0-%3 accumulators, %4-%7 b128 destinations, %8-%11 b64 destinations, %12-%16 VALU registers, %17 A, %18 B, %19 LDS address, %20 / %21 constants
#define M_(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %17, %18, %" #acc "\n\t"
#define R128_(d, off) "ds_read_b128 %" #d ", %19 offset:" #off "\n\t"
#define RTR_(d, off) "ds_read_b64_tr_b16 %" #d ", %22 offset:" #off "\n\t"
#define R64_(d, off) "ds_read_b64 %" #d ", %22 offset:" #off "\n\t"
#define MIX5 "v_exp_f32 %12, %12\n\tv_add_f32 %13, %13, %20\n\tv_exp_f32 %14, %14\n\tv_add_f32 %15, %15, %20\n\tv_cvt_pk_bf16_f32 %16, %16, %20\n\t"
#define WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define NONE
// groups of 4 MFMAs; G = what follows each MFMA
#define Q4(G0, G1, G2, G3) M_(0) G0 M_(1) G1 M_(2) G2 M_(3) G3
#define BODY(G0, G1, G2, G3) Q4(G0, G1, G2, G3) WAIT Q4(G0, G1, G2, G3) WAIT Q4(G0, G1, G2, G3) WAIT Q4(G0, G1, G2, G3) WAIT

#define DEF_PROBE(NAME, BODYSTR)                                                                                                   \
    __global__ __launch_bounds__(768) void NAME(unsigned long long* out, int iters) {                                              \
        __shared__ __attribute__((aligned(16))) unsigned lds[16384];                                                               \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;                                            \
        bf16x8 a, b;                                                                                                               \
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (float)((threadIdx.x + j) & 31)); b[j] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + j) & 15)); } \
        f32x16 c0, c1, c2, c3;                                                                                                     \
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }                                       \
        u32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;                                                                        \
        u32x2 e0 = {0, 0}, e1 = e0, e2 = e0, e3 = e0;                                                                              \
        float f0 = -0.1f, f1 = -0.2f, f2 = -0.3f, f3 = -0.4f, f4 = -0.5f, k1 = 0.999f, k2 = -0.0005f;                              \
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024; \
        const unsigned addr8 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 512; \
        __syncthreads();                                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                       \
            asm volatile(BODYSTR                                                                                                   \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), \
                           "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4)                                                        \
                         : "v"(a), "v"(b), "v"(addr), "v"(k1), "v"(k2), "v"(addr8) : "memory");                                                \
        }                                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s = f0 + f1 + f2 + f3 + f4 + (float)(d0.x + d1.y + d2.z + d3.w + e0.x + e1.y + e2.x + e3.y);                        \
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];                                                           \
        if (s == 12345.678f) out[0] = 1;                                                                                           \
        if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                     \
    }
DEF_PROBE(q_none, BODY(NONE, NONE, NONE, NONE))
DEF_PROBE(q_mix, BODY(MIX5, MIX5, MIX5, MIX5))
DEF_PROBE(q_r128_h, BODY(R128_(4, 0), NONE, R128_(5, 4096), NONE))                                   // one b128 per two MFMAs (the QK^T rate at two query blocks)
DEF_PROBE(q_r128_1, BODY(R128_(4, 0), R128_(5, 4096), R128_(6, 8192), R128_(7, 12288)))            // one per MFMA (QK^T at one query block per wave)
DEF_PROBE(q_r128_2, BODY(R128_(4, 0) R128_(5, 4096), R128_(6, 8192) R128_(7, 12288), R128_(4, 16384) R128_(5, 20480), R128_(6, 24576) R128_(7, 28672)))
DEF_PROBE(q_r128_1_mix, BODY(R128_(4, 0) MIX5, R128_(5, 4096) MIX5, R128_(6, 8192) MIX5, R128_(7, 12288) MIX5))
DEF_PROBE(q_r128_h_mix, BODY(R128_(4, 0) MIX5, MIX5, R128_(5, 4096) MIX5, MIX5))
DEF_PROBE(q_tr_1, BODY(RTR_(8, 0), RTR_(9, 4096), RTR_(10, 8192), RTR_(11, 12288)))
DEF_PROBE(q_tr_2, BODY(RTR_(8, 0) RTR_(9, 4096), RTR_(10, 8192) RTR_(11, 12288), RTR_(8, 16384) RTR_(9, 20480), RTR_(10, 24576) RTR_(11, 28672)))   // PV at one query block
DEF_PROBE(q_tr_2_mix, BODY(RTR_(8, 0) RTR_(9, 4096) MIX5, RTR_(10, 8192) RTR_(11, 12288) MIX5, RTR_(8, 16384) RTR_(9, 20480) MIX5, RTR_(10, 24576) RTR_(11, 28672) MIX5))
DEF_PROBE(q_tr_1_mix, BODY(RTR_(8, 0) MIX5, RTR_(9, 4096) MIX5, RTR_(10, 8192) MIX5, RTR_(11, 12288) MIX5))
DEF_PROBE(q_r64_2, BODY(R64_(8, 0) R64_(9, 4096), R64_(10, 8192) R64_(11, 12288), R64_(8, 16384) R64_(9, 20480), R64_(10, 24576) R64_(11, 28672)))
DEF_PROBE(q_r64_2_mix, BODY(R64_(8, 0) R64_(9, 4096) MIX5, R64_(10, 8192) R64_(11, 12288) MIX5, R64_(8, 16384) R64_(9, 20480) MIX5, R64_(10, 24576) R64_(11, 28672) MIX5))
// the attention tile's own ratio: 8 b128 + 16 tr per 16 MFMAs
DEF_PROBE(q_attn, Q4(R128_(4, 0), R128_(5, 4096), R128_(6, 8192), R128_(7, 12288)) WAIT Q4(R128_(4, 0), R128_(5, 4096), R128_(6, 8192), R128_(7, 12288)) WAIT
                  Q4(RTR_(8, 0) RTR_(9, 4096), RTR_(10, 8192) RTR_(11, 12288), RTR_(8, 16384) RTR_(9, 20480), RTR_(10, 24576) RTR_(11, 28672)) WAIT
                  Q4(RTR_(8, 0) RTR_(9, 4096), RTR_(10, 8192) RTR_(11, 12288), RTR_(8, 16384) RTR_(9, 20480), RTR_(10, 24576) RTR_(11, 28672)) WAIT)
DEF_PROBE(q_attn_mix, Q4(R128_(4, 0) MIX5, R128_(5, 4096) MIX5, R128_(6, 8192) MIX5, R128_(7, 12288) MIX5) WAIT Q4(R128_(4, 0) MIX5, R128_(5, 4096) MIX5, R128_(6, 8192) MIX5, R128_(7, 12288) MIX5) WAIT
                  Q4(RTR_(8, 0) RTR_(9, 4096) MIX5, RTR_(10, 8192) RTR_(11, 12288) MIX5, RTR_(8, 16384) RTR_(9, 20480) MIX5, RTR_(10, 24576) RTR_(11, 28672) MIX5) WAIT
                  Q4(RTR_(8, 0) RTR_(9, 4096) MIX5, RTR_(10, 8192) RTR_(11, 12288) MIX5, RTR_(8, 16384) RTR_(9, 20480) MIX5, RTR_(10, 24576) RTR_(11, 28672) MIX5) WAIT)

// dependent forms (what a real kernel does): the MFMA of slot i takes the fragment requested two slots earlier as its A operand,
// with a counted wait in front of it; AHEAD = 2 over four (b128) / four pairs of (tr64) destination registers
#define W1 "s_waitcnt lgkmcnt(1)\n\t"
#define W2 "s_waitcnt lgkmcnt(2)\n\t"
#define W3 "s_waitcnt lgkmcnt(3)\n\t"
#define MD_(acc, d) "v_mfma_f32_32x32x16_bf16 %" #acc ", %" #d ", %18, %" #acc "\n\t"
#define DEPQ(F) W1 MD_(0, 4) R128_(6, 0) F W1 MD_(1, 5) R128_(7, 4096) F W1 MD_(2, 6) R128_(4, 8192) F W1 MD_(3, 7) R128_(5, 12288) F
DEF_PROBE(q_dep128, DEPQ(NONE) DEPQ(NONE) DEPQ(NONE) DEPQ(NONE))
DEF_PROBE(q_dep128_mix, DEPQ(MIX5) DEPQ(MIX5) DEPQ(MIX5) DEPQ(MIX5))
// tr64 pairs: destinations e0..e3 (%8-%11) are 64-bit; an MFMA A operand is 128-bit, so consume d-registers but WAIT on the tr reads
#define DEPT(F) W2 MD_(0, 4) RTR_(8, 0) RTR_(9, 4096) F W2 MD_(1, 5) RTR_(10, 8192) RTR_(11, 12288) F W2 MD_(2, 6) RTR_(8, 16384) RTR_(9, 20480) F W2 MD_(3, 7) RTR_(10, 24576) RTR_(11, 28672) F
DEF_PROBE(q_deptr, DEPT(NONE) DEPT(NONE) DEPT(NONE) DEPT(NONE))
DEF_PROBE(q_deptr_mix, DEPT(MIX5) DEPT(MIX5) DEPT(MIX5) DEPT(MIX5))
#define DEPQ3(F) W2 MD_(0, 4) R128_(7, 0) F W2 MD_(1, 5) R128_(4, 4096) F W2 MD_(2, 6) R128_(5, 8192) F W2 MD_(3, 7) R128_(6, 12288) F
DEF_PROBE(q_dep128_a3_mix, DEPQ3(MIX5) DEPQ3(MIX5) DEPQ3(MIX5) DEPQ3(MIX5))

typedef void (*kern_t)(unsigned long long*, int);
struct Variant { const char* name; kern_t k; };
#define V(n) {#n, n}
static Variant variants[] = {V(q_none), V(q_mix), V(q_r128_h), V(q_r128_1), V(q_r128_2), V(q_r128_h_mix), V(q_r128_1_mix), V(q_tr_1), V(q_tr_2), V(q_tr_1_mix), V(q_tr_2_mix),
                             V(q_r64_2), V(q_r64_2_mix), V(q_attn), V(q_attn_mix), V(q_dep128), V(q_dep128_mix), V(q_dep128_a3_mix), V(q_deptr), V(q_deptr_mix)};

int main() {
    const int iters = 2000;
    const int grids[2] = {16, 256};
    unsigned long long* d;
    (void)hipMalloc(&d, (1 + 256 * 16) * sizeof(unsigned long long));
    std::vector<unsigned long long> h(1 + 256 * 16);
    printf("# wall ns per 16-MFMA iteration (512 matrix-pipe cycles per wave); in brackets: slowest wave's shader cycles per MFMA per SIMD\n");
    printf("%-16s | %-18s %-18s %-18s | %-18s %-18s\n", "variant", "16wg x1", "16wg x2", "16wg x3", "256wg x1", "256wg x2");
    for (const Variant& v : variants) {
        printf("%-16s |", v.name);
        for (int gi = 0; gi < 2; ++gi)
            for (int wps = 1; wps <= (gi == 0 ? 3 : 2); ++wps) {
                const int grid = grids[gi], threads = 256 * wps;
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(threads), 0, 0, d, 50);
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(threads), 0, 0, d, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                double mx = 0;
                for (int b = 0; b < grid; ++b)
                    for (int w = 0; w < 4 * wps; ++w) mx = std::max(mx, (double)h[1 + b * 16 + w]);
                printf(" %7.0f [%6.1f]   ", ms * 1e6 / iters, mx / (16.0 * iters * wps));
                if (gi == 0 && wps == 3) printf("|");
            }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
