// What does it cost a CU to move operand tiles global -> LDS, by path?  (is the 21 B/clk/CU of the conv /
// GEMM K loop a property of LDS-DMA or of any global -> LDS stream?)
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction), what conv_igemm_kernel / conv_ring_kernel use
//   mode 1: global_load_dwordx4 into VGPRs + ds_write_b128 (register staging, what hipBLASLt's kernels do)
//   mode 2: like 1, but the loads of step s+1 are issued before the ds_writes of step s (one stage of registers in flight)
// One 512-thread workgroup per CU streams PIECES KiB per wave per step out of an L2-resident buffer (each workgroup re-reads
// its own 1 MB window), STEPS steps, one barrier per step like the K loop.  Prints bytes / clock / CU at the measured time
// and a 2.4 GHz nominal clock, for 1 and 2 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PIECES, int LDSB>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, unsigned* __restrict__ sink, int steps, long window) {
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (long)blockIdx.x * window;
    const long wmask = window - 1;
    unsigned acc = 0;
    u32x4 r[PIECES], r2[PIECES];
    long off = (long)wid * PIECES * 1024 + lane * 16;
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) r[i] = *(const u32x4*)(base + ((off + i * 1024) & wmask));
        off += 8 * PIECES * 1024;
    }
    for (int s = 0; s < steps; ++s) {
        char* stage = lds + (s & 1) * (LDSB / 2) + wid * PIECES * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((off + i * 1024) & wmask)),
                                                 (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) r[i] = *(const u32x4*)(base + ((off + i * 1024) & wmask));
#pragma unroll
            for (int i = 0; i < PIECES; ++i) *(u32x4*)(stage + i * 1024 + lane * 16) = r[i];
        } else {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) r2[i] = *(const u32x4*)(base + ((off + i * 1024) & wmask));
#pragma unroll
            for (int i = 0; i < PIECES; ++i) *(u32x4*)(stage + i * 1024 + lane * 16) = r[i];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) r[i] = r2[i];
        }
        off += 8 * PIECES * 1024;
        __syncthreads();
        acc += *(const unsigned*)(lds + (s & 1) * (LDSB / 2) + ((tid * 4 + s * 64) & (LDSB / 2 - 1)));      // keep the LDS data alive
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int MODE, int PIECES, int LDSB>
static void run(const char* name, const char* buf, unsigned* sink, int grid, int steps, long window) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<MODE, PIECES, LDSB>), dim3(grid), dim3(512), 0, 0, buf, sink, steps, window);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, PIECES, LDSB>), dim3(grid), dim3(512), 0, 0, buf, sink, steps, window);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)grid * steps * 8.0 * PIECES * 1024.0;
    const double per_cu = bytes / 256.0 / (ms * 1e-3);              // B/s per CU
    printf("%-34s grid %4d  %d KiB/wave/step  LDS %3d KB: %7.3f ms  %6.2f TB/s chip  %6.1f GB/s/CU = %5.1f B/clk/CU @2.4GHz\n", name, grid,
           PIECES, LDSB / 1024, ms, bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / 2.4e9);
}

int main() {
    char* buf;
    unsigned* sink;
    hipMalloc(&buf, 512L << 20);
    hipMemset(buf, 1, 512L << 20);
    hipMalloc(&sink, 4096);
    const int steps = 400;
    for (long window : {65536L, 1L << 20}) {      // 64 KB per workgroup: the XCD's L2 holds every window; 1 MB: Infinity Cache
        printf("-- source window %ld KB per workgroup (%s)\n", window >> 10, window == 65536 ? "L2 resident" : "beyond L2");
        // one workgroup per CU: 144 KB of LDS (two 72 KB stages = the conv tile), 9 pieces per wave per step
        run<0, 9, 147456>("LDS-DMA                1 WG/CU", buf, sink, 256, steps, window);
        run<1, 9, 147456>("VGPR + ds_write        1 WG/CU", buf, sink, 256, steps, window);
        run<2, 9, 147456>("VGPR, loads run ahead  1 WG/CU", buf, sink, 256, steps, window);
        // two workgroups per CU: 64 KB of LDS each, 4 pieces per wave per step
        run<0, 4, 65536>("LDS-DMA                2 WG/CU", buf, sink, 512, steps, window);
        run<1, 4, 65536>("VGPR + ds_write        2 WG/CU", buf, sink, 512, steps, window);
        run<2, 4, 65536>("VGPR, loads run ahead  2 WG/CU", buf, sink, 512, steps, window);
    }
    return 0;
}
