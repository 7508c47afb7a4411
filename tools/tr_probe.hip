// Probe of gfx950's ds_read_b64_tr_b16 (run on the MI355X): prints, for every lane, which 16-bit LDS elements it
// received when lane l addresses the 8-byte chunk l (elements 4l .. 4l+3).  attn_fwd.hip's PV operand gather relies
// on: within a 16-lane group, lane i gets element (i & 3) of the chunks of lanes (i >> 2) + 4j, j = 0..3.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(s16x4* out) {
    __shared__ short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + threadIdx.x * 4));
}
int main() {
    s16x4* d; s16x4 h[64];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int i = l & 15, grp = l >> 4;
            const int expect = (grp * 16 + (i >> 2) + 4 * j) * 4 + (i & 3);
            printf(" %3d%s", h[l][j], h[l][j] == expect ? "" : "!");
            bad += h[l][j] != expect;
        }
        printf("\n");
    }
    printf("mismatches vs the assumed gather: %d\n", bad);
    return bad != 0;
}
