// What rate do the operand tiles of a token-major GEMM reach a CU at, as a function of how many 64-channel stages are in flight?
//   The ring kernel's tile walk and staging pattern without its consumers: 256 persistent workgroups
// of 512 threads walk the (M/256) x (N/320) tiles of  x[M, K] @ w[N, K]^T  in the XCD-aware order of conv_ring_kernel
// (cout groups as the launcher picks them); per tile and 64-channel stage every wave issues its 9 LDS-DMA pieces (whole
// 128-byte lines: 4 of the 256 activation rows, 5 of the 320 weight rows) and waits until at most D - 1 older stages are
// still in flight, then a barrier.  No MFMA, no fragment reads, no epilogue: the time is what the memory system needs to
// deliver the operand stream with D stages (72 KB each) of requests outstanding per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gemm_stream_probe.hip -o tools/gemm_stream_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int D, int T>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ xg, const char* __restrict__ wg, unsigned* __restrict__ sink,
                                             long M, int K, int N, int ngroups) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 73728];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / 320;
    const long nblocks = (M / 256) * tiles_n;
    const int per_xcd = gridDim.x / 8;
    const int xg_n = 8 / ngroups, tn_g = tiles_n / ngroups;
    const int grp = (blockIdx.x % 8) / xg_n;
    const long ntiles = nblocks / ngroups;
    const long tile_first = (long)((blockIdx.x % 8) % xg_n) * per_xcd + blockIdx.x / 8;
    const long tile_step = (long)xg_n * per_xcd;
    const int srow = tid / 8, ch = (tid % 8) * 16;
    const int nst = K / 64;
    const long pitch = (long)K * 2;
    unsigned acc = 0;
    int s_glob = 0;
    // T > 0: L2 touch-prefetch T stages ahead of the request: ONE dword per 128-byte line, each line by one of the CUs that
    // share the tile in this round (activation tile: the tn_g CUs on its cout tiles; weight tile: the 32 / tn_g CUs on its pixel
    // tiles) -- wave 0 touches activation lines, wave 1 weight lines; the touch is a plain load into a dead register
    const int lane = tid & 63;
    const int sa = tn_g < 32 ? tn_g : 32, sb = 32 / sa > 0 ? 32 / sa : 1;
    unsigned dead = 0;
    for (long tile = tile_first; tile < ntiles; tile += tile_step) {
        const long m0 = (tile / tn_g) * 256;
        const int n0 = (grp * tn_g + (int)(tile % tn_g)) * 320;
        const char* a = xg + (m0 + srow) * pitch + ch;
        const char* b = wg + (long)(n0 + srow) * pitch + ch;
        const long nxt = tile + tile_step < ntiles ? tile + tile_step : tile;
        const long m1 = (nxt / tn_g) * 256;
        const int n1 = (grp * tn_g + (int)(nxt % tn_g)) * 320;
        const int nloc = (int)(tile % tn_g), mloc = (int)((tile / tn_g) % sb);
        for (int st = 0; st < nst; ++st, ++s_glob) {
            if (T > 0 && wid < 2) {
                int ts = st + T;
                const bool cross = ts >= nst;
                ts = cross ? ts - nst : ts;
                if (ts < nst) {
                    if (wid == 0) {
                        for (int r = lane * sa + (nloc % sa); r < 256; r += 64 * sa)
                            dead += *(const volatile unsigned*)(xg + ((cross ? m1 : m0) + r) * pitch + ts * 128);
                    } else {
                        for (int r = lane * sb + (mloc % sb); r < 320; r += 64 * sb)
                            dead += *(const volatile unsigned*)(wg + (long)((cross ? n1 : n0) + r) * pitch + ts * 128);
                    }
                }
            }
            char* stage = lds + (s_glob & 1) * 73728 + wid * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + i * 64 * pitch + st * 128),
                                                 (__attribute__((address_space(3))) void*)(stage + i * 8192), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + i * 64 * pitch + st * 128),
                                                 (__attribute__((address_space(3))) void*)(stage + 32768 + i * 8192), 16, 0, 0);
            if (T > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the touches were issued BEFORE this stage's pieces: in-order vmcnt)
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 * (D - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += *(const unsigned*)(lds + tid * 4) + dead;
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int D, int T>
static void run(const char* x, const char* w, unsigned* sink, long M, int K, int N, int ng) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<D, T>), dim3(256), dim3(512), 0, 0, x, w, sink, M, K, N, ng);
    (void)hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<D, T>), dim3(256), dim3(512), 0, 0, x, w, sink, M, K, N, ng);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)(M / 256) * (N / 320) * (K / 64) * 73728.0;
    const double ideal = 2.0 * M * K * N / 2.5e15 * 1e3;
    printf("  D=%d stages in flight, touch %d ahead: %7.3f ms  %6.2f TB/s into LDS  %5.1f B/clk/CU @2.4GHz   (MFMA time of this GEMM at peak %.3f ms)\n", D, T, ms,
           bytes / (ms * 1e-3) / 1e12, bytes / 256.0 / (ms * 1e-3) / 2.4e9, ideal);
}

int main() {
    char *x, *w;
    unsigned* sink;
    (void)hipMalloc(&x, 2L << 30);
    (void)hipMalloc(&w, 256L << 20);
    (void)hipMemset(x, 1, 2L << 30);
    (void)hipMemset(w, 1, 256L << 20);
    (void)hipMalloc(&sink, 4096);
    struct { const char* name; long M; int K, N, ng; } shapes[] = {
        {"pers L2 ff-in  40960 x 1280 -> 10240, 8 cout groups", 40960, 1280, 10240, 8}, {"same, 1 group", 40960, 1280, 10240, 1},
        {"pers L1 ff-out 163840 x 2560 -> 640", 163840, 2560, 640, 1}, {"pers L0 ff-out 655360 x 1280 -> 320", 655360, 1280, 320, 1},
        {"pers L1 qkv 163840 x 640 -> 1920 (2 groups)", 163840, 640, 1920, 2}, {"pers L0 proj 655360 x 320 -> 320", 655360, 320, 320, 1}};
    for (auto& s : shapes) {
        printf("%s\n", s.name);
        run<1, 0>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<2, 0>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<3, 0>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<1, 1>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<1, 2>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<1, 3>(x, w, sink, s.M, s.K, s.N, s.ng);
        run<1, 4>(x, w, sink, s.M, s.K, s.N, s.ng);
    }
    return 0;
}
