// Empirical check of the v_mfma_f32_32x32x16_bf16 fragment layouts assumed by imagine360_amd/csrc:
//   A: lane l holds A[l & 31][8 * (l >> 5) + j],  B: lane l holds B[8 * (l >> 5) + j][l & 31],
//   C/D: lane l reg r holds C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {   // A [32][16], B [16][32], C [32][32]
    const int l = threadIdx.x, hi = l >> 5, c = l & 31;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[c * 16 + 8 * hi + j];
        b[j] = (__bf16)B[(8 * hi + j) * 32 + c];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + c] = acc[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hC[32 * 32], ref[32 * 32];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 5 - 2); }
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + n]; ref[i * 32 + n] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
    printf("mfma_f32_32x32x16_bf16 layout probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
