// Loss reductions (hinge GAN, feature-matching L1), their element-wise gradients, the flat-arena
// Adam step, and the library's error/version entry points.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------ error state
static thread_local char g_err[512] = "";
void s2e_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* s2e_last_error(void) { return g_err; }
int s2e_deterministic(void) {
    static const int v = [] { const char* e = getenv("S2E_DETERMINISTIC"); return e ? atoi(e) != 0 : 0; }();
    return v;
}
extern "C" int s2e_version(void) { return 1; }

// ------------------------------------------------------------------------------------ zero fill (see common.h)
__global__ void s2e_zero_kernel(uint32_t* __restrict__ p, size_t n_words) {
    const size_t nv = n_words / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x)
        ((u32x4_t*)p)[i] = u32x4_t{0, 0, 0, 0};
    for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}
int s2e_zero_async(void* ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return S2E_OK;
    if (!ptr || (bytes & 3) || ((uintptr_t)ptr & 15))
        S2E_FAIL(S2E_ERR_ARG, "s2e_zero_async: need a 16-byte aligned pointer and a size that is a multiple of 4");
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256 + 1;
    if (blocks > 4096) blocks = 4096;
    s2e_zero_kernel<<<(int)blocks, 256, 0, st>>>((uint32_t*)ptr, words);
    S2E_CHECK_LAUNCH("s2e_zero_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ loss reduce
template <int MODE> __device__ __forceinline__ float loss_term(float a, float b) {
    if (MODE == S2E_LOSS_NEG_MEAN) return -a;
    if (MODE == S2E_LOSS_HINGE_REAL) return -fminf(a - 1.f, 0.f);
    if (MODE == S2E_LOSS_HINGE_FAKE) return -fminf(-a - 1.f, 0.f);
    return fabsf(a - b);
}
template <int MODE> __device__ __forceinline__ float loss_dterm(float a, float b) {
    if (MODE == S2E_LOSS_NEG_MEAN) return -1.f;
    // torch.min(x, 0) (loss.py:68,71) splits the gradient 0.5/0.5 at an exact tie x == 0
    if (MODE == S2E_LOSS_HINGE_REAL) { const float x = a - 1.f; return x < 0.f ? -1.f : (x == 0.f ? -0.5f : 0.f); }
    if (MODE == S2E_LOSS_HINGE_FAKE) { const float x = -a - 1.f; return x < 0.f ? 1.f : (x == 0.f ? 0.5f : 0.f); }
    const float d = a - b;
    return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);       // torch l1_loss backward = sign(a-b)
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b, long n, float scale,
                                                          float* __restrict__ out) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float red[4];
    float s = 0.f;
    // 16-B vector path only when both pointers are 16-B aligned (a fake/real half of an odd-sized map is not)
    const bool al = ((((uintptr_t)a) | (MODE == S2E_LOSS_L1 ? (uintptr_t)b : 0)) & 15) == 0;
    const long nv = al ? n / VEC : 0;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
        float fa[VEC], fb[VEC];
        unpack16<T>(*(const u32x4_t*)(a + v * VEC), fa);
        if (MODE == S2E_LOSS_L1) unpack16<T>(*(const u32x4_t*)(b + v * VEC), fb);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += loss_term<MODE>(fa[j], MODE == S2E_LOSS_L1 ? fb[j] : 0.f);
    }
    for (long i = nv * VEC + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        s += loss_term<MODE>(load1<T>(a + i), MODE == S2E_LOSS_L1 ? load1<T>(b + i) : 0.f);   // tail / unaligned
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, scale * (red[0] + red[1] + red[2] + red[3]));
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_grad_kernel(const T* __restrict__ a, const T* __restrict__ b, long n, float scale,
                                 const float* __restrict__ gscale, T* __restrict__ da, int accumulate) {
    constexpr int VEC = Vec<T>::N;
    if (gscale) scale *= *gscale;
    // 16-byte vectors when every pointer is 16-byte aligned (round 5: element by element -- 2-byte loads and stores -- the eight
    // feature-matching gradients of a G step took 12 us each for 2 ... 17 MB)
    const bool al = ((((uintptr_t)a) | ((uintptr_t)da) | (MODE == S2E_LOSS_L1 ? (uintptr_t)b : 0)) & 15) == 0;
    const long nv = al ? n / VEC : 0;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
        float fa[VEC], fb[VEC], g[VEC];
        unpack16<T>(*(const u32x4_t*)(a + v * VEC), fa);
        if (MODE == S2E_LOSS_L1) unpack16<T>(*(const u32x4_t*)(b + v * VEC), fb);
        if (accumulate) unpack16<T>(*(const u32x4_t*)(da + v * VEC), g);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float d = scale * loss_dterm<MODE>(fa[j], MODE == S2E_LOSS_L1 ? fb[j] : 0.f);
            g[j] = accumulate ? d + g[j] : d;
        }
        *(u32x4_t*)(da + v * VEC) = pack16<T>(g);
    }
    for (long i = nv * VEC + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float g = scale * loss_dterm<MODE>(load1<T>(a + i), MODE == S2E_LOSS_L1 ? load1<T>(b + i) : 0.f);
        if (accumulate) g += load1<T>(da + i);
        store1<T>(da + i, g);
    }
}

template <typename T>
static int loss_dispatch(bool grad, int mode, const T* a, const T* b, long n, float scale, const float* gscale, void* out,
                         int accumulate, hipStream_t st) {
    // one 16-byte vector per thread and trip.  The reduction ends with ONE float atomic per block on a single address, and
    // same-address atomics serialise at ~12 ns each: 1024 blocks cost the feature-matching sums more (13 us) than their 2 ... 17 MB
    // of reads -- at most 256 blocks
    const long work = n / Vec<T>::N + 1;
    const int cap = grad ? 2048 : 256;
    const int grid = (int)((work + 255) / 256 < cap ? (work + 255) / 256 : cap);
#define S2E_L(MM) do { if (grad) loss_grad_kernel<T, MM><<<grid, 256, 0, st>>>(a, b, n, scale, gscale, (T*)out, accumulate); \
                       else loss_reduce_kernel<T, MM><<<grid, 256, 0, st>>>(a, b, n, scale, (float*)out); } while (0)
    switch (mode) {
        case S2E_LOSS_NEG_MEAN: S2E_L(S2E_LOSS_NEG_MEAN); break;
        case S2E_LOSS_HINGE_REAL: S2E_L(S2E_LOSS_HINGE_REAL); break;
        case S2E_LOSS_HINGE_FAKE: S2E_L(S2E_LOSS_HINGE_FAKE); break;
        case S2E_LOSS_L1: S2E_L(S2E_LOSS_L1); break;
        default: S2E_FAIL(S2E_ERR_ARG, "loss: bad mode %d", mode);
    }
#undef S2E_L
    S2E_CHECK_LAUNCH("loss kernel");
    return S2E_OK;
}

extern "C" int s2e_loss_reduce(int dtype, int mode, const void* a, const void* b, long n, float scale, float* out, void* stream) {
    if (!a || !out || n <= 0 || (mode == S2E_LOSS_L1 && !b)) S2E_FAIL(S2E_ERR_ARG, "s2e_loss_reduce: bad argument");
    if (dtype == S2E_BF16) return loss_dispatch<bf16_t>(false, mode, (const bf16_t*)a, (const bf16_t*)b, n, scale, nullptr, out, 0, (hipStream_t)stream);
    if (dtype == S2E_F32) return loss_dispatch<float>(false, mode, (const float*)a, (const float*)b, n, scale, nullptr, out, 0, (hipStream_t)stream);
    S2E_FAIL(S2E_ERR_ARG, "s2e_loss_reduce: bad dtype %d", dtype);
}
extern "C" int s2e_loss_grad(int dtype, int mode, const void* a, const void* b, long n, float scale, const float* gscale,
                             void* da, int accumulate, void* stream) {
    if (!a || !da || n <= 0 || (mode == S2E_LOSS_L1 && !b)) S2E_FAIL(S2E_ERR_ARG, "s2e_loss_grad: bad argument");
    if (dtype == S2E_BF16) return loss_dispatch<bf16_t>(true, mode, (const bf16_t*)a, (const bf16_t*)b, n, scale, gscale, da, accumulate, (hipStream_t)stream);
    if (dtype == S2E_F32) return loss_dispatch<float>(true, mode, (const float*)a, (const float*)b, n, scale, gscale, da, accumulate, (hipStream_t)stream);
    S2E_FAIL(S2E_ERR_ARG, "s2e_loss_grad: bad dtype %d", dtype);
}

// ------------------------------------------------------------------------------------ Adam over a flat arena
// Hyper-parameters live in DEVICE memory (hyper[0..5] = lr, beta1, beta2, eps, completed steps, grad_scale)
// so that a captured hipGraph replays with the current learning rate and bias corrections.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, long n, const float* __restrict__ hyper) {
    const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], t = hyper[4] + 1.f, grad_scale = hyper[5];
    const float wd = hyper[6];                              // torch.optim.Adam's L2 term: g += weight_decay * p
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float lr_bc1 = lr / bc1, rsqrt_bc2 = 1.f / sqrtf(bc2);
    const long nv = n / 4;
    // beta1 == 0 (the reference's TTUR setting, pix2pix_model.py:98-108) and no weight decay: m_t = g_t * grad_scale EXACTLY, whatever m_{t-1}
    // was (0 * m + 1 * g), and bc1 = 1 -- so the first moment is neither read nor written: 20 instead of 28 bytes per parameter, the
    // same bits in p and v.  (The caller can form m from g when it wants to save it: optim.FlatAdam.state_dict.)
    if (beta1 == 0.f && wd == 0.f) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
            f32x4_t pp = ((f32x4_t*)p)[i], gg = ((const f32x4_t*)g)[i], vv = ((f32x4_t*)v)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gr = gg[j] * grad_scale + wd * pp[j];
                const float mj = beta1 * 0.f + (1.f - beta1) * gr;
                vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
                pp[j] -= lr_bc1 * mj / (sqrtf(vv[j]) * rsqrt_bc2 + eps);
            }
            ((f32x4_t*)p)[i] = pp; ((f32x4_t*)v)[i] = vv;
        }
        if (blockIdx.x == 0)
            for (long i = nv * 4 + threadIdx.x; i < n; i += blockDim.x) {
                const float gr = g[i] * grad_scale + wd * p[i];
                const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
                v[i] = vi;
                p[i] -= lr_bc1 * ((1.f - beta1) * gr) / (sqrtf(vi) * rsqrt_bc2 + eps);
            }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        f32x4_t pp = ((f32x4_t*)p)[i], gg = ((const f32x4_t*)g)[i], mm = ((f32x4_t*)m)[i], vv = ((f32x4_t*)v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * grad_scale + wd * pp[j];
            mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
            vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
            pp[j] -= lr_bc1 * mm[j] / (sqrtf(vv[j]) * rsqrt_bc2 + eps);
        }
        ((f32x4_t*)p)[i] = pp; ((f32x4_t*)m)[i] = mm; ((f32x4_t*)v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gr = g[i] * grad_scale + wd * p[i];
            const float mi = beta1 * m[i] + (1.f - beta1) * gr;
            const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
            m[i] = mi; v[i] = vi;
            p[i] -= lr_bc1 * mi / (sqrtf(vi) * rsqrt_bc2 + eps);
        }
}
__global__ void adam_tick_kernel(float* hyper) { hyper[4] += 1.f; }

extern "C" int s2e_adam_flat(float* p, const float* g, float* m, float* v, long n, float* hyper, void* stream) {
    if (!p || !g || !m || !v || !hyper || n <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_adam_flat: bad argument");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) S2E_FAIL(S2E_ERR_ARG, "s2e_adam_flat: arenas must be 16-byte aligned");
    const long nv = n / 4 + 1;
    const int grid = (int)((nv + 255) / 256 < 4096 ? (nv + 255) / 256 : 4096);
    adam_flat_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper);
    S2E_CHECK_LAUNCH("adam_flat_kernel");
    adam_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>(hyper);      // after every block has read hyper[4]
    S2E_CHECK_LAUNCH("adam_tick_kernel");
    return S2E_OK;
}

// ---- data-parallel gradient exchange, 'direct' form (seg2eye_amd/distributed.py): the owner of a bucket shard adds the P copies the
// all-to-all delivered, in rank order, fp32 accumulation, ONE rounding to the payload dtype -- every replica gets the same bits from the
// all-gather that follows.  recv: [world][shard] elements; out: [shard].  One 16-byte vector per thread.
template <typename T>
__global__ __launch_bounds__(256) void shard_sum_kernel(const T* __restrict__ recv, T* __restrict__ out, int world, long shard) {
    constexpr int VEC = Vec<T>::N;
    const long nv = shard / VEC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        float a[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = 0.f;
        for (int r = 0; r < world; ++r) {
            float f[VEC];
            unpack16<T>(*(const u32x4_t*)(recv + (size_t)r * shard + i * VEC), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] += f[j];
        }
        *(u32x4_t*)(out + i * VEC) = pack16<T>(a);
    }
    if (blockIdx.x == 0)
        for (long i = nv * VEC + threadIdx.x; i < shard; i += blockDim.x) {
            float a = 0.f;
            for (int r = 0; r < world; ++r) a += load1<T>(recv + (size_t)r * shard + i);
            store1<T>(out + i, a);
        }
}

extern "C" int s2e_shard_sum(int dtype, const void* recv, void* out, int world, long shard, void* stream) {
    if (!recv || !out || world <= 0 || shard <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_shard_sum: bad argument");
    if ((((uintptr_t)recv | (uintptr_t)out) & 15) || (shard * (dtype == S2E_BF16 ? 2 : 4)) % 16 != 0)
        S2E_FAIL(S2E_ERR_ARG, "s2e_shard_sum: buffers and the shard size must be 16-byte multiples");
    const long nv = shard / (dtype == S2E_BF16 ? 8 : 4) + 1;
    const int grid = (int)((nv + 255) / 256 < 2048 ? (nv + 255) / 256 : 2048);
    if (dtype == S2E_BF16) shard_sum_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)recv, (bf16_t*)out, world, shard);
    else if (dtype == S2E_F32) shard_sum_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)recv, (float*)out, world, shard);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_shard_sum: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("shard_sum_kernel");
    return S2E_OK;
}
