// OpenEDS validation metric on the device: the reference's Tester generates an image, resizes it to
// 400 x 640 with cv2.INTER_LINEAR, maps [-1, 1] to 0..255 with an int TRUNCATION and scores sqrt(sum d^2) / (H W) per image
// (util/tester.py:44-47,93-97; data/postprocessor.py:58-73,92-107; models/networks/loss.py:102-155).
//
//   to255(x)        = (int)(((x + 1) * 255) / 2)         fp32 operations in exactly this order, truncation toward zero
//   resize          : cv2.INTER_LINEAR on a float64 image (half-pixel centres, edge clamp, float tap weights, double sums; no
//                     antialiasing when shrinking), then ((v + 1) * 255) / 2 in double and the truncation
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

// out[n][oy][ox] = trunc(((bilinear(x[n], oy, ox) + 1) * 255) / 2); one thread per output pixel.
// Arithmetic = OpenCV's INTER_LINEAR on a float64 image followed by the reference's float64 unnormalize
// (data/postprocessor.py:108-114,58-73), step by step: tap position f = (float)((d + 0.5) * scale - 0.5) with scale a double,
// s = floor(f), f -= s, clamped to the image at both edges with weight 0; FLOAT weights (1 - f, f); a horizontal pass then a
// vertical pass S0 * w0 + S1 * w1 in DOUBLE; ((v + 1) * 255) / 2 in double; truncation toward zero.  No fused multiply-adds
// (cv2 and numpy round every product), so the value agrees with the oracle's fp64 restatement to the last bit.
struct LinTap { int s0, s1; float w0, w1; };
__device__ __forceinline__ LinTap lin_tap(int d, double scale, int n_src) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    LinTap t;
    t.s0 = s; t.s1 = s + 1 < n_src ? s + 1 : n_src - 1;
    t.w0 = 1.f - f; t.w1 = f;
    return t;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_to255_kernel(const T* __restrict__ x, uint8_t* __restrict__ out, int H, int W, int Ho, int Wo,
                                                           double sy, double sx) {
#pragma clang fp contract(off)
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
    if (ox >= Wo) return;
    const LinTap tx = lin_tap(ox, sx, W), ty = lin_tap(oy, sy, H);
    const T* p = x + (size_t)n * H * W;
    const double v00 = (double)load1<T>(p + (size_t)ty.s0 * W + tx.s0), v01 = (double)load1<T>(p + (size_t)ty.s0 * W + tx.s1);
    const double v10 = (double)load1<T>(p + (size_t)ty.s1 * W + tx.s0), v11 = (double)load1<T>(p + (size_t)ty.s1 * W + tx.s1);
    const double top = __dadd_rn(__dmul_rn(v00, (double)tx.w0), __dmul_rn(v01, (double)tx.w1));
    const double bot = __dadd_rn(__dmul_rn(v10, (double)tx.w0), __dmul_rn(v11, (double)tx.w1));
    const double v = __dadd_rn(__dmul_rn(top, (double)ty.w0), __dmul_rn(bot, (double)ty.w1));
    const double u = __ddiv_rn(__dmul_rn(__dadd_rn(v, 1.0), 255.0), 2.0);
    int q = (int)u;                                          // truncation toward zero (`.int()`)
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
    // cv2: inv_scale = dst / (double)src; scale = 1. / inv_scale
    const double sy = 1.0 / ((double)Ho / (double)H), sx = 1.0 / ((double)Wo / (double)W);
    if (dtype == S2E_BF16) resize_to255_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, out, H, W, Ho, Wo, sy, sx);
    else if (dtype == S2E_F32) resize_to255_kernel<float><<<grid, 256, 0, st>>>((const float*)x, out, H, W, Ho, Wo, sy, sx);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_resize_to255: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("resize_to255_kernel");
    return S2E_OK;
}
