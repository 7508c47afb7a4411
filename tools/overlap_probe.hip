// Do v_mfma and VALU / transcendental work overlap on one gfx950 SIMD?  Loops of 16 MFMAs (four independent accumulators),
// 128 v_fma (eight independent chains) and 32 v_exp per iteration, alone and together, in ONE wave per SIMD (same wave issues
// both) and in TWO waves per SIMD (one wave issues the MFMAs, its partner the VALU work).  Prints ns per iteration per config.
// Build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o /tmp/overlap_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, bool SPLIT>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wid = threadIdx.x >> 6;
    bool do_m = MODE & 1, do_v = MODE & 2, do_e = MODE & 4;
    if (SPLIT) {        // waves 0-3 (one per SIMD) take the matrix work, waves 4-7 the vector work
        if (wid < 4) { do_v = false; do_e = false; } else { do_m = false; }
    }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8], e[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + threadIdx.x * 1e-3f + i; e[i] = -0.1f * i - threadIdx.x * 1e-4f; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 16 || MODE == 17 || (MODE == 19 && wid < 4)) {      // accumulators in AGPRs (inline asm), optionally 8 fma after each MFMA
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                    if (MODE == 17) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
                    }
                }
            continue;
        }
        if (MODE == 19) {        // partner wave of the AGPR-MFMA wave: fma only
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
            continue;
        }
        if (MODE == 8) {         // hand-interleaved in ONE wave: each MFMA followed by eight independent fma
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            continue;
        }
        if (do_m) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        }
        if (do_e) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.5f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_grid = 256;
template <int MODE, bool SPLIT>
static void run(const char* name, int threads, float* d) {
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, SPLIT>), dim3(g_grid), dim3(threads), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, SPLIT>), dim3(g_grid), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %8.1f ns / iteration\n", name, ms * 1e6 / iters);
}

int main(int argc, char** argv) {
    float* d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    for (int pass = 0; pass < 2; ++pass) {
    g_grid = pass ? 16 : 256;
    printf("---- %d workgroups (%s)\n", g_grid, pass ? "a sixteenth of the chip: no power cap" : "every CU");
    printf("per iteration: 16 MFMA 32x32x16 (= 512 matrix-pipe cycles), 128 v_fma, 32 v_exp; one workgroup per CU\n");
    run<1, false>("1 wave/SIMD: MFMA only", 256, d);
    run<2, false>("1 wave/SIMD: 128 fma only", 256, d);
    run<4, false>("1 wave/SIMD: 32 exp only", 256, d);
    run<6, false>("1 wave/SIMD: fma + exp", 256, d);
    run<3, false>("1 wave/SIMD: MFMA + fma (same wave)", 256, d);
    run<7, false>("1 wave/SIMD: MFMA + fma + exp (same wave)", 256, d);
    run<1, false>("2 waves/SIMD: MFMA only (both)", 512, d);
    run<6, false>("2 waves/SIMD: fma + exp (both)", 512, d);
    run<7, false>("2 waves/SIMD: MFMA + fma + exp (both, same wave)", 512, d);
    run<3, true>("2 waves/SIMD: one MFMA, partner fma", 512, d);
    run<7, true>("2 waves/SIMD: one MFMA, partner fma + exp", 512, d);
    run<8, false>("1 wave/SIMD: MFMA, 8 fma, MFMA, ... hand-interleaved", 256, d);
    run<16, false>("1 wave/SIMD: MFMA only, accumulators in AGPRs", 256, d);
    run<17, false>("1 wave/SIMD: AGPR MFMA, 8 fma, ... interleaved", 256, d);
    run<19, false>("2 waves/SIMD: one AGPR MFMA, partner fma", 512, d);
    }
    return 0;
}
