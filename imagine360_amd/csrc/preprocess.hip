// Host-preprocessing geometry of the inference script on the MI355X, gfx950.
//
// The reference warps every input frame on the CPU with cv2.remap(uint8 image, float32 maps, INTER_CUBIC, BORDER_WRAP):
// 20 perspective views per panorama frame (process_equi, inference_dual_p2e.py:113-144; Equirec2Perspec.py:61) and one
// equirectangular canvas per input frame (pers2pano_vid, :291-304; Perspec2Equirec.py:65).  The sampling maps depend on
// the camera only, so one launch warps all frames through all maps: out[n][m] = remap(img[n], map[m]).
//
// Arithmetic = OpenCV's fixed-point bicubic remap, restated from the published imgwarp.cpp (OpenCV is not in this image:
// parity unpinned, see oracle/im360_oracle/preprocess.py): coordinates quantised to 1/32 pixel with round-half-even, a
// 1024 x 16 table of int16 weights (2^15 scale) built by the host (imagine360_amd/preprocess.py), modulo border handling,
// (sum + 2^14) >> 15 saturated to uint8.  HBM-bound by construction: every output byte is written once, the 16 taps of
// neighbouring pixels hit the same cache lines.
#include "common.h"

namespace im360 {

__device__ __forceinline__ int wrap_index(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

template <int C>
__global__ __launch_bounds__(256) void remap_cubic_wrap_u8_kernel(const uint8_t* __restrict__ img, const float* __restrict__ map_x,
                                                                   const float* __restrict__ map_y, const short* __restrict__ wtab,
                                                                   uint8_t* __restrict__ out, long N, int M, int H, int W, int h,
                                                                   int w) {
    const long per_map = (long)h * w;
    const long total = N * M * per_map;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % per_map;
        const long t = i / per_map;
        const int m = (int)(t % M);
        const long n = t / M;
        const int sx = __float2int_rn(map_x[m * per_map + pix] * 32.0f);          // cvRound: round half to even
        const int sy = __float2int_rn(map_y[m * per_map + pix] * 32.0f);
        const short* wt = wtab + (((sy & 31) << 5) + (sx & 31)) * 16;
        int ix = sx >> 5, iy = sy >> 5;
        ix = (ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix)) - 1;              // saturate_cast<short>, then the first tap
        iy = (iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy)) - 1;
        int xs[4], acc[C];
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k] = wrap_index(ix + k, W) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0;
        const uint8_t* base = img + n * (long)H * W * C;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const uint8_t* row = base + (long)wrap_index(iy + k1, H) * W * C;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int wgt = wt[k1 * 4 + k2];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += (int)row[xs[k2] + c] * wgt;
            }
        }
        uint8_t* o = out + i * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int v = (acc[c] + (1 << 14)) >> 15;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

}  // namespace im360

// out[n][m] = cv2.remap(img[n], map_x[m], map_y[m], INTER_CUBIC, borderMode=BORDER_WRAP) for uint8 images [N, H, W, C]
// (C = 1, 3 or 4), float32 maps [M, h, w] in source pixels, wtab = the 1024 x 16 int16 bicubic weight table.
extern "C" __attribute__((visibility("default"))) int im360_remap_cubic_wrap_u8(const void* img, const void* map_x, const void* map_y, const void* wtab, void* out,
                                         int64_t N, int64_t M, int64_t H, int64_t W, int64_t C, int64_t h, int64_t w,
                                         void* stream) {
    using namespace im360;
    IM360_CHECK_ARG(img && map_x && map_y && wtab && out, "remap_cubic_wrap_u8: null pointer");
    IM360_CHECK_ARG(N > 0 && M > 0 && H > 0 && W > 0 && h > 0 && w > 0, "remap_cubic_wrap_u8: empty problem");
    IM360_CHECK_ARG(H <= 32767 && W <= 32767, "remap_cubic_wrap_u8: source larger than 32767 pixels per side");
    IM360_CHECK_ARG(C == 1 || C == 3 || C == 4, "remap_cubic_wrap_u8: %ld channels unsupported (1, 3, 4)", (long)C);
    IM360_CHECK_ARG(((uintptr_t)map_x % 4) == 0 && ((uintptr_t)map_y % 4) == 0 && ((uintptr_t)wtab % 2) == 0, "remap_cubic_wrap_u8: misaligned pointer");
    const long total = N * M * h * w;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define IM360_REMAP(CC)                                                                                              \
    hipLaunchKernelGGL((remap_cubic_wrap_u8_kernel<CC>), dim3(blocks), dim3(256), 0, s, (const uint8_t*)img,          \
                       (const float*)map_x, (const float*)map_y, (const short*)wtab, (uint8_t*)out, (long)N, (int)M,   \
                       (int)H, (int)W, (int)h, (int)w)
    if (C == 1) IM360_REMAP(1);
    else if (C == 3) IM360_REMAP(3);
    else IM360_REMAP(4);
#undef IM360_REMAP
    IM360_CHECK_LAUNCH();
    return IM360_OK;
}

// Largest all-ones rectangle of a HOST mask [H, W] (uint8, 1 = inside): rect = (top, left, width, height), with the scan
// order and tie-breaking of the reference's pure-Python get_maxrec_cord (src/modules/utils.py:39-73: column heights, one
// monotone stack per row, the first strictly larger area wins).  Host code on purpose: it is a sequential O(H W) scan of a
// mask the host already holds (the reference spends ~0.1 s per frame in Python loops on it).
extern "C" __attribute__((visibility("default"))) int im360_max_rect(const uint8_t* mask, int64_t H, int64_t W, int64_t* rect) {
    IM360_CHECK_ARG(mask && rect && H > 0 && W > 0, "max_rect: null pointer / empty mask");
    IM360_CHECK_ARG(H * W <= (1L << 31), "max_rect: mask too large");
    int* dp = (int*)malloc(sizeof(int) * (size_t)W);
    int* stack = (int*)malloc(sizeof(int) * (size_t)(W + 1));
    if (!dp || !stack) {
        free(dp);
        free(stack);
        im360_set_error("max_rect: out of host memory");
        return IM360_ERR_ARG;
    }
    for (int64_t j = 0; j < W; ++j) dp[j] = 0;
    long best = 0;
    rect[0] = rect[1] = rect[2] = rect[3] = 0;
    for (int64_t i = 0; i < H; ++i) {
        for (int64_t j = 0; j < W; ++j) dp[j] = mask[i * W + j] == 1 ? dp[j] + 1 : 0;
        int top = 0;                                   // stack size
        for (int64_t j = 0; j <= W; ++j) {
            const int hcur = j < W ? dp[j] : 0;
            while (top > 0 && hcur < dp[stack[top - 1]]) {
                const int idx = stack[--top];
                const long hv = dp[idx];
                const long wv = top == 0 ? j : j - stack[top - 1] - 1;
                if (hv * wv > best) {
                    best = hv * wv;
                    rect[0] = i - hv + 1;
                    rect[1] = top == 0 ? 0 : stack[top - 1] + 1;
                    rect[2] = wv;
                    rect[3] = hv;
                }
            }
            stack[top++] = (int)j;
        }
    }
    free(dp);
    free(stack);
    return IM360_OK;
}
