// Does the length of the contiguous row segment a wave's LDS-DMA instruction covers change what a CU gets out of its L2?
// (the conv / GEMM K loops move 21 B/clk/CU, tools/dma_probe.hip streams fully contiguous KiBs at 43.)
// Every workgroup (512 threads, one per CU) reads the same R x PITCH byte matrix (L2 resident) the way a K loop does: per
// step 32 KB = (32 KB / SEG) rows x SEG contiguous bytes, K blocks fastest; global_load_lds_dwordx4, one barrier per step.
//   SEG = 64: the ring kernel's 32-channel phases; 128: the two-stage kernel's 64-channel steps; 256 / 512 / whole rows.
// `shared` = 1: all workgroups read ONE matrix (weights); 0: each workgroup its own (activations, still L2 / MALL resident).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_seg_probe.hip -o tools/dma_seg_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SEG>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, unsigned* __restrict__ sink, int steps, int R, int PITCH, long wg_stride) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 32768];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (long)blockIdx.x * wg_stride;
    constexpr int ROWS = 32768 / SEG;
    const int kblocks = PITCH / SEG, rblocks = R / ROWS;
    unsigned acc = 0;
    int off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = ((wid * 4 + i) * 64 + lane) * 16;
        off[i] = (q / SEG) * PITCH + (q % SEG);
    }
    int kb = 0, rb = 0;
    for (int s = 0; s < steps; ++s) {
        char* stage = lds + (s & 3) * 32768 + wid * 4096;
        const char* b = base + (long)rb * ROWS * PITCH + kb * SEG;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + off[i]),
                                             (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
        if (++kb == kblocks) { kb = 0; if (++rb == rblocks) rb = 0; }
        if (s >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *(const unsigned*)(lds + ((s + 2) & 3) * 32768 + ((tid * 4 + s * 64) & 32767));
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int SEG>
static void run(const char* buf, unsigned* sink, int R, int PITCH, int shared) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int steps = 800, grid = 256;
    const long stride = shared ? 0 : (long)R * PITCH;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<SEG>), dim3(grid), dim3(512), 0, 0, buf, sink, steps, R, PITCH, stride);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<SEG>), dim3(grid), dim3(512), 0, 0, buf, sink, steps, R, PITCH, stride);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)grid * steps * 32768.0, per_cu = bytes / 256.0 / (ms * 1e-3);
    printf("SEG %5d B  R %5d rows  pitch %5d B  %s: %7.3f ms  %6.2f TB/s chip  %5.1f B/clk/CU @2.4GHz\n", SEG, R, PITCH,
           shared ? "one matrix for all " : "a matrix per workgroup", ms, bytes / (ms * 1e-3) / 1e12, per_cu / 2.4e9);
}

int main() {
    char* buf;
    unsigned* sink;
    hipMalloc(&buf, 1L << 30);
    hipMemset(buf, 1, 1L << 30);
    hipMalloc(&sink, 4096);
    for (int shared = 1; shared >= 0; --shared)
        for (int pitch : {640, 2560}) {
            const int R = shared ? 512 : (pitch == 640 ? 512 : 512);
            run<64>(buf, sink, R, pitch, shared);
            run<128>(buf, sink, R, pitch, shared);
            if (pitch % 256 == 0) run<256>(buf, sink, R, pitch, shared);
            if (pitch % 512 == 0) run<512>(buf, sink, R, pitch, shared);
            if (pitch == 640) run<640>(buf, sink, 1024, pitch, shared);      // whole rows: SEG = pitch -> fully contiguous stream
            if (pitch == 2560) run<2560>(buf, sink, 1024, pitch, shared);
        }
    return 0;
}
