// OpenEDS validation metric on the device: the reference's Tester generates an image, resizes it to
// 400 x 640 with cv2.INTER_LINEAR, maps [-1, 1] to 0..255 with an int TRUNCATION and scores sqrt(sum d^2) / (H W) per image
// (util/tester.py:44-47,93-97; data/postprocessor.py:58-73,92-107; models/networks/loss.py:102-155).
//
//   to255(x)        = (int)(((x + 1) * 255) / 2)         fp32 operations in exactly this order, truncation toward zero
//   resize          : bilinear with half-pixel centres and edge clamping (cv2.INTER_LINEAR on float images and
//                     torch's align_corners=False agree on this; no antialiasing when shrinking), fp32
//   err[n]          = sqrtf((float)sum_pixels (a - b)^2) / (float)(H * W)      a, b integers 0..255; the sum is exact (u64)
//
// HBM-bound, tiny (one pass over two images): one workgroup per image, no workspace, no atomics, deterministic.
#include "common.h"

namespace {

__device__ __forceinline__ int to255(float x) { return (int)(((x + 1.f) * 255.f) / 2.f); }

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;                                                // valid in thread 0
}

template <typename T>
__global__ __launch_bounds__(1024) void openeds_error_kernel(const T* __restrict__ a, const T* __restrict__ b, long HW, float* __restrict__ err) {
    __shared__ unsigned long long red[16];
    const T* pa = a + (size_t)blockIdx.x * HW;
    const T* pb = b + (size_t)blockIdx.x * HW;
    unsigned long long s = 0;
    for (long i = threadIdx.x; i < HW; i += blockDim.x) {
        const int d = to255(load1<T>(pa + i)) - to255(load1<T>(pb + i));
        s += (unsigned long long)(d * d);
    }
    const unsigned long long t = block_sum_u64(s, red);
    if (threadIdx.x == 0) err[blockIdx.x] = sqrtf((float)t) / (float)HW;
}

__global__ __launch_bounds__(1024) void openeds_error_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, long HW, float* __restrict__ err) {
    __shared__ unsigned long long red[16];
    const uint8_t* pa = a + (size_t)blockIdx.x * HW;
    const uint8_t* pb = b + (size_t)blockIdx.x * HW;
    unsigned long long s = 0;
    for (long i = threadIdx.x; i < HW; i += blockDim.x) {
        const int d = (int)pa[i] - (int)pb[i];
        s += (unsigned long long)(d * d);
    }
    const unsigned long long t = block_sum_u64(s, red);
    if (threadIdx.x == 0) err[blockIdx.x] = sqrtf((float)t) / (float)HW;
}

// out[n][oy][ox] = to255(bilinear(x[n], oy, ox)); one thread per output pixel
template <typename T>
__global__ __launch_bounds__(256) void resize_to255_kernel(const T* __restrict__ x, uint8_t* __restrict__ out, int H, int W, int Ho, int Wo,
                                                           float sy, float sx) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
    if (ox >= Wo) return;
    // source coordinate of the pixel centre, clamped at the edges (cv2: fx < 0 -> 0, sx >= W - 1 -> last pixel, weight 0)
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    float wy = fy - (float)y0, wx = fx - (float)x0;
    if (y0 >= H - 1) { y0 = H - 1; wy = 0.f; }
    if (x0 >= W - 1) { x0 = W - 1; wx = 0.f; }
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const T* p = x + (size_t)n * H * W;
    const float v00 = load1<T>(p + (size_t)y0 * W + x0), v01 = load1<T>(p + (size_t)y0 * W + x1);
    const float v10 = load1<T>(p + (size_t)y1 * W + x0), v11 = load1<T>(p + (size_t)y1 * W + x1);
    // horizontal first, then vertical (the order of torch's upsample_bilinear2d and of cv2's two-pass resize)
    const float top = v00 + wx * (v01 - v00), bot = v10 + wx * (v11 - v10);
    float v = top + wy * (bot - top);
    int q = to255(v);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    out[((size_t)n * Ho + oy) * Wo + ox] = (uint8_t)q;
}

}  // namespace

extern "C" int s2e_openeds_error(int dtype, const void* fake, const void* target, int N, int H, int W, float* err, void* stream) {
    if (!fake || !target || !err || N <= 0 || H <= 0 || W <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_openeds_error: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    if (dtype == S2E_BF16) openeds_error_kernel<bf16_t><<<N, 1024, 0, st>>>((const bf16_t*)fake, (const bf16_t*)target, HW, err);
    else if (dtype == S2E_F32) openeds_error_kernel<float><<<N, 1024, 0, st>>>((const float*)fake, (const float*)target, HW, err);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_openeds_error: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("openeds_error_kernel");
    return S2E_OK;
}

extern "C" int s2e_openeds_error_u8(const uint8_t* produced, const uint8_t* target, int N, int H, int W, float* err, void* stream) {
    if (!produced || !target || !err || N <= 0 || H <= 0 || W <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_openeds_error_u8: bad argument");
    openeds_error_u8_kernel<<<N, 1024, 0, (hipStream_t)stream>>>(produced, target, (long)H * W, err);
    S2E_CHECK_LAUNCH("openeds_error_u8_kernel");
    return S2E_OK;
}

extern "C" int s2e_resize_to255(int dtype, const void* x, int N, int H, int W, uint8_t* out, int Ho, int Wo, void* stream) {
    if (!x || !out || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_resize_to255: bad argument");
    if (Ho > 65535 || N > 65535) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_resize_to255: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ceil_div(Wo, 256), Ho, N);
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    if (dtype == S2E_BF16) resize_to255_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, out, H, W, Ho, Wo, sy, sx);
    else if (dtype == S2E_F32) resize_to255_kernel<float><<<grid, 256, 0, st>>>((const float*)x, out, H, W, Ho, Wo, sy, sx);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_resize_to255: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("resize_to255_kernel");
    return S2E_OK;
}
