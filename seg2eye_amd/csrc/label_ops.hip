// Label-map and resampling kernels (all HBM-bound, NHWC, 16-byte channel groups per thread).
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------ label conv3x3
// conv3x3(one_hot(nearest_down(label))) == sum over the 9 taps of one table row selected by the
// neighbour's class: zero MACs, 9 LDS row reads per 16-B output vector.  The [9*ncls][<=128] slice of
// the table this block needs sits in LDS (<= 18 KiB for ncls = 4).
// bx of nbx: this block's place among the blocks that walk the pixels of this (layer, channel chunk `by`)
template <typename T>
__device__ __forceinline__ void label_conv3x3_body(const uint8_t* __restrict__ label, const float* __restrict__ weight,
        const float* __restrict__ bias, T* __restrict__ out, int N, int H, int W, int h, int w, int ncls, int Cout, int relu,
        int bx, int nbx, int by, float* tab, int fast_min = 65536) {
    constexpr int VEC = Vec<T>::N;
    constexpr int CT = 128, CTP = CT + 4;                 // channels per block; LDS row pitch in floats (see the table fill)
    const int cbase = by * CT;
    const int cw = min(CT, Cout - cbase);
    const int rows = 9 * ncls;
    // the table straight from the OIHW conv weight: tab[(tap*ncls + cls)][cc] = weight[cbase+cc][cls][tap].  Lanes walk the SOURCE (the
    // block's [cw][rows] slice is one contiguous run: coalesced 4-byte loads); the rows of the LDS table are CTP = CT + 4 floats apart, so
    // consecutive lanes (r, r + 1, ...: 132 floats further each) spread over eight banks -- with a pitch of CT they all met on one, which
    // is why rounds 1-5 walked the channel instead and paid for it with one cache line per lane: that preamble, not the pixel loop, was the
    // cost of the batched launch (155 us at up to 1024 blocks a layer; fewer blocks for the small layers alone: 124 us)
    for (int i = threadIdx.x; i < cw * rows; i += blockDim.x) {
        const int cc = i / rows, r = i - cc * rows;       // r = cls*9 + tap in the OIHW weight
        const int cls = r / 9, tap = r - cls * 9;
        tab[(tap * ncls + cls) * CTP + cc] = weight[(size_t)cbase * rows + i];
    }
    if (cw < CT)
        for (int i = threadIdx.x; i < rows * (CT - cw); i += blockDim.x) {
            const int r = i / (CT - cw), cc = cw + i - r * (CT - cw);
            tab[r * CTP + cc] = 0.f;
        }
    for (int i = threadIdx.x; i < CT; i += blockDim.x) tab[rows * CTP + i] = (bias && i < cw) ? bias[cbase + i] : 0.f;
    __syncthreads();
    // Label maps are piecewise constant: for an interior pixel whose 3x3 neighbourhood is ONE class the sum is a
    // per-class constant.  uni[cls][cc] = bias + sum_tap tab[tap][cls] (same order as the general path below, so
    // both paths give the same bits); such pixels cost one LDS vector read instead of nine.
    float* uni = tab + (rows + 1) * CTP;                  // [ncls][CT]
    for (int i = threadIdx.x; i < ncls * CT; i += blockDim.x) {
        const int cl = i / CT, cc = i - cl * CT;
        float a = tab[rows * CTP + cc];
        for (int t = 0; t < 9; ++t) a += tab[(t * ncls + cl) * CTP + cc];
        uni[i] = a;
    }
    __syncthreads();
    // ... and for bf16 outputs that constant AS IT IS STORED (ReLU applied, rounded): nine pixels in ten of a label map are
    // interior to a class, and the general path below spends ~70 instructions per 16-byte vector on them (two LDS reads, eight
    // max, the conversions, the address).  With the stored form the uniform path is one 16-byte LDS read and one store (150 -> 148 us a
    // launch: the launch was not bound there -- ablations: 125 us without its stores, 133 without its label loads; what paid was fewer
    // blocks for the small layers, label_conv_blocks).
    uint16_t* unib = (uint16_t*)(uni + ncls * CT);         // [ncls][CT] bf16 bits
    if constexpr (std::is_same<T, bf16_t>::value) {
        for (int i = threadIdx.x; i < ncls * CT; i += blockDim.x) {
            const float a = relu ? fmaxf(uni[i], 0.f) : uni[i];
            unib[i] = (uint16_t)f32_to_bf16_bits(a);
        }
        __syncthreads();
    }

    const int cgb = (cw + VEC - 1) / VEC;                 // lanes per pixel
    const int sy = H / h, sx = W / w;
    const int hw = h * w, npix = N * hw;                  // < 2^31 (checked by the launcher)
    // word of a pixel: its 9 neighbour classes, 3 bits each (7 = padding); bit 27 = "all nine are the same class"
    auto load_word = [&](int pix) __attribute__((always_inline)) -> unsigned {
        const int n = pix / hw, rem = pix - n * hw;
        const int y = rem / w, x = rem - y * w;
        const uint8_t* lb = label + (size_t)n * H * W;
        unsigned wd = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
            const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy), xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
            const unsigned v = lb[(size_t)yc * sy * W + (size_t)xc * sx];
            wd |= (ok ? (v & 7u) : 7u) << (3 * t);
        }
        const unsigned c0 = wd & 7u;
        if (c0 != 7u && wd == c0 * 0x1249249u) wd |= 1u << 27;      // 0x1249249 = sum of 8^t, t < 9
        return wd;
    };
    // emit one pixel's 16-byte channel group given its word
    auto emit = [&](unsigned word, int pix, int tx) __attribute__((always_inline)) {
        if constexpr (std::is_same<T, bf16_t>::value) {
            if ((word & (1u << 27)) && (Cout % VEC) == 0) {
                *(u32x4_t*)(out + (size_t)pix * Cout + cbase + tx * VEC) = *(const u32x4_t*)(unib + (word & 7u) * CT + tx * VEC);
                return;
            }
        }
        float acc[VEC];
        if (word & (1u << 27)) {
            const float* u = uni + (word & 7u) * CT + tx * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = u[j];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = tab[rows * CTP + tx * VEC + j];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned c = (word >> (3 * t)) & 7u;
                if (c != 7u) {
                    const float* trow = tab + (t * ncls + c) * CTP + tx * VEC;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += trow[j];
                }
            }
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaxf(acc[j], 0.f);
        }
        T* o = out + (size_t)pix * Cout + cbase + tx * VEC;
        if ((Cout % VEC) == 0) *(u32x4_t*)o = pack16<T>(acc);
        else for (int j = 0; j < VEC && cbase + tx * VEC + j < Cout; ++j) store1<T>(o + j, acc[j]);
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((64 % cgb) == 0 && npix >= fast_min) {            // (small maps of a launch of their own: too few 256-pixel blocks to fill the chip)
        // A wave owns 64 consecutive pixels per pass: every lane fetches the word of ITS pixel (nine byte loads, no
        // redundancy across the pixel's lanes), then the wave emits the 64 pixels in cgb sub-steps of 64/cgb pixels,
        // each lane pulling the word it needs with one shuffle.  The coordinate arithmetic and the label loads are
        // thus paid once per 64 pixels instead of once per 64/cgb: the loop was VALU-issue-bound (~150 wave
        // instructions per KB stored), not HBM-bound.
        // (tried: the one-in-ten pixels of the nine-tap path queued per wave and emitted densely behind the uniform ones, so that they do not
        //  drag the other pixels of their sub-step along: 119 -> 121 us per launch, not kept)
        const int ppw = 64 / cgb;
        const int tx = lane % cgb, sub = lane / cgb;
        for (int base = (bx * 4 + wave) * 64; base < npix; base += nbx * 256) {
            const int mine = base + lane;
            const unsigned wmine = mine < npix ? load_word(mine) : 0u;
            for (int k = 0; k < cgb; ++k) {
                const int src = k * ppw + sub;
                const unsigned word = __shfl(wmine, src, 64);
                if (base + src < npix) emit(word, base + src, tx);
            }
        }
    } else {
        const int ppb = 256 / cgb;
        const int tx = threadIdx.x % cgb, ty = threadIdx.x / cgb;
        if (ty >= ppb) return;
        for (int pix = bx * ppb + ty; pix < npix; pix += nbx * ppb) emit(load_word(pix), pix, tx);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void label_conv3x3_kernel(const uint8_t* __restrict__ label, const float* __restrict__ weight,
        const float* __restrict__ bias, T* __restrict__ out, int N, int H, int W, int h, int w, int ncls, int Cout, int relu) {
    extern __shared__ __attribute__((aligned(16))) float tab[];   // [9*ncls][CT], bias[CT], uni[ncls][CT]
    label_conv3x3_body<T>(label, weight, bias, out, N, H, W, h, w, ncls, Cout, relu, blockIdx.x, gridDim.x, blockIdx.y, tab);
}

constexpr int LABEL_FAST_MIN_BATCH = 2048;             // (see label_conv_blocks)
// all label convs of a generator forward in ONE launch (19 mlp_shared convs, most of them a few microseconds of work behind a
// ~5-us launch): block_map[b] = {job, bx, nbx}; outputs at out_base + job.out_off (one buffer per forward)
template <typename T>
__global__ __launch_bounds__(256) void label_conv3x3_batch_kernel(const uint8_t* __restrict__ label, const s2e_label_conv_job* __restrict__ jobs,
        const int* __restrict__ block_map, char* __restrict__ out_base, int N, int H, int W, int ncls) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_label_conv_job J = jobs[bm[0]];
    label_conv3x3_body<T>(label, J.weight, J.bias, (T*)(out_base + J.out_off), N, H, W, J.h, J.w, ncls, J.cout, J.relu, bm[1], bm[2], 0, tab, LABEL_FAST_MIN_BATCH);
}

// blocks along the pixels of one layer (the single launch's grid.x)
// fast_min: the wave-per-64-pixels path from this many pixels on -- 65536 for a layer launched alone (fewer 256-pixel blocks would not
// fill the chip), 2048 inside the batched launch, where the other layers' blocks do: a 64^2 x 8 layer then has 128 blocks instead of 1024,
// each of which builds the layer's 18-KB table first
static long label_conv_blocks(int dtype, long npix, int Cout, int fast_min = 65536) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    const int cw = Cout < 128 ? Cout : 128;
    const int cgb_h = (cw + vec - 1) / vec;
    const int ppb = ((64 % cgb_h) == 0 && npix >= fast_min) ? 256 : 256 / cgb_h;       // pixels per block per pass
    const long gx = (npix + ppb - 1) / ppb;
    static const long cap = [] { const char* e = getenv("S2E_LABEL_CONV_CAP"); return e ? atol(e) : 1024L; }();   // (the optimum of 256 ... 2048, measured)
    return gx > cap ? cap : gx;
}

extern "C" int s2e_label_conv3x3(int dtype, const uint8_t* label, const float* weight, const float* bias, void* out,
                                 int N, int H, int W, int h, int w, int ncls, int Cout, int relu, void* stream) {
    if (!label || !weight || !out || N <= 0 || h <= 0 || w <= 0 || Cout <= 0 || ncls <= 0 || ncls > 7)
        S2E_FAIL(S2E_ERR_ARG, "s2e_label_conv3x3: bad argument");
    if (H % h || W % w) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_label_conv3x3: %dx%d is not an integer multiple of %dx%d", H, W, h, w);
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_label_conv3x3: bad dtype %d", dtype);
    const long npix = (long)N * h * w;
    if (npix >= (1L << 31)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_label_conv3x3: too many pixels for 32-bit indices");
    const long gx = label_conv_blocks(dtype, npix, Cout);
    dim3 grid((unsigned)gx, ceil_div(Cout, 128));
    const size_t lds = ((size_t)(9 * ncls + 1) * 132 + (size_t)ncls * 128) * sizeof(float) + (size_t)ncls * 128 * 2;     // table rows at a pitch of 132 floats, uniform rows, the same as stored bf16
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) label_conv3x3_kernel<bf16_t><<<grid, 256, lds, st>>>(label, weight, bias, (bf16_t*)out, N, H, W, h, w, ncls, Cout, relu);
    else label_conv3x3_kernel<float><<<grid, 256, lds, st>>>(label, weight, bias, (float*)out, N, H, W, h, w, ncls, Cout, relu);
    S2E_CHECK_LAUNCH("label_conv3x3_kernel");
    return S2E_OK;
}

// Batched form: jobs (DEVICE array; every cout <= 128) share the label batch; block_map (DEVICE int32 triples) from
// s2e_label_conv_block_map, which also serves to count (block_map_host NULL).
extern "C" long s2e_label_conv_block_map(int dtype, const s2e_label_conv_job* jobs_host, int n_jobs, int N, int* block_map_host) {
    if (!jobs_host || n_jobs < 0 || N <= 0) return S2E_ERR_ARG;
    long nb = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const s2e_label_conv_job& J = jobs_host[j];
        if (J.cout <= 0 || J.cout > 128 || J.h <= 0 || J.w <= 0) return S2E_ERR_ARG;
        const long gx = label_conv_blocks(dtype, (long)N * J.h * J.w, J.cout, LABEL_FAST_MIN_BATCH);
        for (long b = 0; b < gx; ++b, ++nb)
            if (block_map_host) { int* e = block_map_host + 3 * nb; e[0] = j; e[1] = (int)b; e[2] = (int)gx; }
    }
    return nb;
}
extern "C" int s2e_label_conv3x3_batch(int dtype, const uint8_t* label, const s2e_label_conv_job* jobs, const int* block_map,
                                       int n_blocks, void* out_base, int N, int H, int W, int ncls, void* stream) {
    if (!label || !jobs || !block_map || !out_base || n_blocks <= 0 || N <= 0 || ncls <= 0 || ncls > 7)
        S2E_FAIL(S2E_ERR_ARG, "s2e_label_conv3x3_batch: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_label_conv3x3_batch: bad dtype %d", dtype);
    const size_t lds = ((size_t)(9 * ncls + 1) * 132 + (size_t)ncls * 128) * sizeof(float) + (size_t)ncls * 128 * 2;     // table rows at a pitch of 132 floats, uniform rows, the same as stored bf16
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) label_conv3x3_batch_kernel<bf16_t><<<n_blocks, 256, lds, st>>>(label, jobs, block_map, (char*)out_base, N, H, W, ncls);
    else label_conv3x3_batch_kernel<float><<<n_blocks, 256, lds, st>>>(label, jobs, block_map, (char*)out_base, N, H, W, ncls);
    S2E_CHECK_LAUNCH("label_conv3x3_batch_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ one-hot (+ image) NHWC
template <typename T>
__global__ void onehot_nhwc_kernel(const uint8_t* __restrict__ label, const T* __restrict__ img, T* __restrict__ out,
                                   int N, int H, int W, int h, int w, int ncls, int cpad) {
    const int sy = H / h, sx = W / w;
    const long total = (long)N * h * w * cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cpad;
        const int c = (int)(i - pix * cpad);
        float v = 0.f;
        if (c < ncls) {
            const int n = (int)(pix / (h * w));
            const int rem = (int)(pix - (long)n * h * w);
            const int y = rem / w, x = rem - y * w;
            v = label[((size_t)n * H + (size_t)y * sy) * W + (size_t)x * sx] == c ? 1.f : 0.f;
        } else if (c == ncls && img) {
            v = load1<T>(img + pix);
        }
        store1<T>(out + i, v);
    }
}

// cpad == 8 (every caller): one thread per PIXEL -- one label byte in, the 8 channels out as one (bf16) or two (fp32) 16-byte stores.
// (The element-per-thread form above: eight 2-byte stores and three 64-bit divisions per pixel, ~10 us per call, ten calls per step.)
template <typename T>
__global__ __launch_bounds__(256) void onehot_nhwc8_kernel(const uint8_t* __restrict__ label, const T* __restrict__ img, T* __restrict__ out,
                                                           int N, int H, int W, int h, int w, int ncls) {
    const int sy = H / h, sx = W / w, total = N * h * w;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += gridDim.x * blockDim.x) {
        const int n = pix / (h * w), rem = pix - n * h * w;
        const int y = rem / w, x = rem - y * w;
        const int cls = label[((size_t)n * H + (size_t)y * sy) * W + (size_t)x * sx];
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (c < ncls && c == cls) ? 1.f : 0.f;
        if (img && ncls < 8) v[ncls] = load1<T>(img + pix);
        if constexpr (Vec<T>::N == 8) {
            *(u32x4_t*)(out + (size_t)pix * 8) = pack16<T>(v);
        } else {
            *(u32x4_t*)(out + (size_t)pix * 8) = pack16<T>(v);
            *(u32x4_t*)(out + (size_t)pix * 8 + 4) = pack16<T>(v + 4);
        }
    }
}

extern "C" int s2e_onehot_nhwc(int dtype, const uint8_t* label, const void* img, void* out,
                               int N, int H, int W, int h, int w, int ncls, int cpad, void* stream) {
    if (!label || !out || N <= 0 || h <= 0 || w <= 0 || ncls <= 0 || cpad < ncls + (img ? 1 : 0))
        S2E_FAIL(S2E_ERR_ARG, "s2e_onehot_nhwc: bad argument");
    if (H % h || W % w) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_onehot_nhwc: non-integer downsampling ratio");
    const long total = (long)N * h * w * cpad;
    hipStream_t st = (hipStream_t)stream;
    if (cpad == 8 && ((uintptr_t)out & 15) == 0 && (long)N * h * w < (1L << 31) && (dtype == S2E_BF16 || dtype == S2E_F32)) {
        const long px = (long)N * h * w;
        const int gridp = (int)((px + 255) / 256 < 4096 ? (px + 255) / 256 : 4096);
        if (dtype == S2E_BF16) onehot_nhwc8_kernel<bf16_t><<<gridp, 256, 0, st>>>(label, (const bf16_t*)img, (bf16_t*)out, N, H, W, h, w, ncls);
        else onehot_nhwc8_kernel<float><<<gridp, 256, 0, st>>>(label, (const float*)img, (float*)out, N, H, W, h, w, ncls);
        S2E_CHECK_LAUNCH("onehot_nhwc8_kernel");
        return S2E_OK;
    }
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == S2E_BF16) onehot_nhwc_kernel<bf16_t><<<grid, 256, 0, st>>>(label, (const bf16_t*)img, (bf16_t*)out, N, H, W, h, w, ncls, cpad);
    else if (dtype == S2E_F32) onehot_nhwc_kernel<float><<<grid, 256, 0, st>>>(label, (const float*)img, (float*)out, N, H, W, h, w, ncls, cpad);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_onehot_nhwc: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("onehot_nhwc_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ nearest x2 upsample
template <typename T>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int h, int w, int cg) {
    constexpr int VEC = Vec<T>::N;
    const int H2 = 2 * h, W2 = 2 * w;
    const long nvec = (long)N * H2 * W2 * cg;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long pix = v / cg;
        const int g = (int)(v - pix * cg);
        const int n = (int)(pix / ((long)H2 * W2));
        const int rem = (int)(pix - (long)n * H2 * W2);
        const int oy = rem / W2, ox = rem - oy * W2;
        const size_t src = (((size_t)n * h + (oy >> 1)) * w + (ox >> 1)) * cg + g;
        *(u32x4_t*)(y + (size_t)v * VEC) = *(const u32x4_t*)(x + src * VEC);
    }
}
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int N, int h, int w, int cg) {
    constexpr int VEC = Vec<T>::N;
    const int W2 = 2 * w;
    const long nvec = (long)N * h * w * cg;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long pix = v / cg;
        const int g = (int)(v - pix * cg);
        const int n = (int)(pix / ((long)h * w));
        const int rem = (int)(pix - (long)n * h * w);
        const int y = rem / w, x = rem - y * w;
        const size_t b = (((size_t)n * 2 * h + 2 * y) * W2 + 2 * x) * cg + g;
        float a[VEC], t[VEC];
        unpack16<T>(*(const u32x4_t*)(gy + b * VEC), a);
        unpack16<T>(*(const u32x4_t*)(gy + (b + cg) * VEC), t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] += t[j];
        unpack16<T>(*(const u32x4_t*)(gy + (b + (size_t)W2 * cg) * VEC), t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] += t[j];
        unpack16<T>(*(const u32x4_t*)(gy + (b + (size_t)W2 * cg + cg) * VEC), t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] += t[j];
        *(u32x4_t*)(gx + (size_t)v * VEC) = pack16<T>(a);
    }
}

static int ups_launch(int dtype, const void* a, void* b, int N, int h, int w, int C, void* stream, bool fwd, const char* name) {
    if (!a || !b || N <= 0 || h <= 0 || w <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "%s: bad argument", name);
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "%s: bad dtype %d", name, dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "%s: C=%d not a multiple of %d", name, C, vec);
    const int cg = C / vec;
    const long nvec = (long)N * h * w * cg * (fwd ? 4 : 1);
    const int grid = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (fwd) {
        if (dtype == S2E_BF16) upsample2x_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, h, w, cg);
        else upsample2x_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, N, h, w, cg);
    } else {
        if (dtype == S2E_BF16) upsample2x_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, h, w, cg);
        else upsample2x_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, N, h, w, cg);
    }
    S2E_CHECK_LAUNCH(name);
    return S2E_OK;
}
extern "C" int s2e_upsample2x_fwd(int dtype, const void* x, void* y, int N, int h, int w, int C, void* stream) {
    return ups_launch(dtype, x, y, N, h, w, C, stream, true, "s2e_upsample2x_fwd");
}
extern "C" int s2e_upsample2x_bwd(int dtype, const void* gy, void* gx, int N, int h, int w, int C, void* stream) {
    return ups_launch(dtype, gy, gx, N, h, w, C, stream, false, "s2e_upsample2x_bwd");
}

// ------------------------------------------------------------------------------------ avg_pool 3x3 s2 p1
// count_include_pad=False: divide by the number of in-bounds taps.
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        const int n = (int)(pix / ((long)Ho * Wo));
        const int rem = (int)(pix - (long)n * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        float s = 0.f; int cnt = 0;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                s += load1<T>(x + (((size_t)n * H + iy) * W + ix) * C + c);
                ++cnt;
            }
        }
        store1<T>(y + i, s / (float)cnt);
    }
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int iy = rem / W, ix = rem - iy * W;
        float s = 0.f;
        // outputs oy with 2*oy-1 <= iy <= 2*oy+1
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= Ho) continue;
            const int cy = min(2 * oy + 1, H - 1) - max(2 * oy - 1, 0) + 1;
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const int cx = min(2 * ox + 1, W - 1) - max(2 * ox - 1, 0) + 1;
                s += load1<T>(gy + (((size_t)n * Ho + oy) * Wo + ox) * C + c) / (float)(cy * cx);
            }
        }
        store1<T>(gx + i, s);
    }
}
// One thread per (pixel, 16-byte channel group) when C is a multiple of the vector width (the discriminator's 8-channel input:
// one vector per pixel); 32-bit index arithmetic.  The element-per-thread forms below (64-bit divisions, 2-byte accesses) took 12 /
// 45 us on the (16, 256, 256, 8) input for 8 / 17 MB of traffic; they remain for ragged channel counts.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    constexpr int VEC = Vec<T>::N;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, cg = C / VEC;
    const int total = N * Ho * Wo * cg;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int pix = i / cg, g = i - pix * cg;
        const int n = pix / (Ho * Wo), rem = pix - n * Ho * Wo;
        const int oy = rem / Wo, ox = rem - oy * Wo;
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        int cnt = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);          // (clamped address, weight 0: no branch per tap)
                float f[VEC];
                unpack16<T>(*(const u32x4_t*)(x + (((size_t)n * H + cy) * W + cx) * C + g * VEC), f);
                if (ok) {
                    ++cnt;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) s[j] += f[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = s[j] / (float)cnt;
        *(u32x4_t*)(y + (size_t)i * VEC) = pack16<T>(s);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_vec_kernel(const T* __restrict__ gy, T* __restrict__ gx, int N, int H, int W, int C) {
    constexpr int VEC = Vec<T>::N;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, cg = C / VEC;
    const int total = N * H * W * cg;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int pix = i / cg, g = i - pix * cg;
        const int n = pix / (H * W), rem = pix - n * H * W;
        const int iy = rem / W, ix = rem - iy * W;
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        // outputs oy with 2*oy-1 <= iy <= 2*oy+1: oy in {iy/2, (iy+1)/2} (one or two), likewise ox
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a ? (iy + 1) / 2 : iy / 2;
            if ((a && oy == iy / 2) || oy >= Ho) continue;
            const int cy = min(2 * oy + 1, H - 1) - max(2 * oy - 1, 0) + 1;
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int ox = bq ? (ix + 1) / 2 : ix / 2;
                if ((bq && ox == ix / 2) || ox >= Wo) continue;
                const int cx = min(2 * ox + 1, W - 1) - max(2 * ox - 1, 0) + 1;
                float f[VEC];
                unpack16<T>(*(const u32x4_t*)(gy + (((size_t)n * Ho + oy) * Wo + ox) * C + g * VEC), f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) s[j] += f[j] / (float)(cy * cx);
            }
        }
        *(u32x4_t*)(gx + (size_t)i * VEC) = pack16<T>(s);
    }
}
static int pool_launch(int dtype, const void* a, void* b, int N, int H, int W, int C, void* stream, bool fwd, const char* name) {
    if (!a || !b || N <= 0 || H <= 0 || W <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "%s: bad argument", name);
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "%s: bad dtype %d", name, dtype);
    const long total = fwd ? (long)N * ((H + 1) / 2) * ((W + 1) / 2) * C : (long)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec == 0 && total / vec < (1L << 31) && (long)N * H * W * C < (1L << 31)) {      // one 16-byte vector per thread
        const long tv = total / vec;
        const int gridv = (int)((tv + 255) / 256 < 8192 ? (tv + 255) / 256 : 8192);
        if (fwd) {
            if (dtype == S2E_BF16) avgpool_fwd_vec_kernel<bf16_t><<<gridv, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, H, W, C);
            else avgpool_fwd_vec_kernel<float><<<gridv, 256, 0, st>>>((const float*)a, (float*)b, N, H, W, C);
        } else {
            if (dtype == S2E_BF16) avgpool_bwd_vec_kernel<bf16_t><<<gridv, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, H, W, C);
            else avgpool_bwd_vec_kernel<float><<<gridv, 256, 0, st>>>((const float*)a, (float*)b, N, H, W, C);
        }
        S2E_CHECK_LAUNCH(name);
        return S2E_OK;
    }
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (fwd) {
        if (dtype == S2E_BF16) avgpool_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, H, W, C);
        else avgpool_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, N, H, W, C);
    } else {
        if (dtype == S2E_BF16) avgpool_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, (bf16_t*)b, N, H, W, C);
        else avgpool_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, N, H, W, C);
    }
    S2E_CHECK_LAUNCH(name);
    return S2E_OK;
}
extern "C" int s2e_avgpool3x3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
    return pool_launch(dtype, x, y, N, H, W, C, stream, true, "s2e_avgpool3x3s2_fwd");
}
extern "C" int s2e_avgpool3x3s2_bwd(int dtype, const void* gy, void* gx, int N, int H, int W, int C, void* stream) {
    return pool_launch(dtype, gy, gx, N, H, W, C, stream, false, "s2e_avgpool3x3s2_bwd");
}

// ------------------------------------------------------------------------------------ tanh backward
template <typename T>
__global__ void tanh_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ gx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float yy = load1<T>(y + i);
        store1<T>(gx + i, load1<T>(gy + i) * (1.f - yy * yy));
    }
}
extern "C" int s2e_tanh_bwd(int dtype, const void* gy, const void* y, void* gx, long n, void* stream) {
    if (!gy || !y || !gx || n <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_tanh_bwd: bad argument");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) tanh_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)gy, (const bf16_t*)y, (bf16_t*)gx, n);
    else if (dtype == S2E_F32) tanh_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)gy, (const float*)y, (float*)gx, n);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_tanh_bwd: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("tanh_bwd_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ LeakyReLU backward
template <typename T>
__global__ void lrelu_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ gx, long n) {
    constexpr int VEC = Vec<T>::N;
    const long nv = n / VEC;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
        float g[VEC], yy[VEC];
        unpack16<T>(*(const u32x4_t*)(gy + v * VEC), g);
        unpack16<T>(*(const u32x4_t*)(y + v * VEC), yy);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] *= (yy[j] > 0.f ? 1.f : 0.2f);
        *(u32x4_t*)(gx + v * VEC) = pack16<T>(g);
    }
    for (long i = nv * VEC + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        store1<T>(gx + i, load1<T>(gy + i) * (load1<T>(y + i) > 0.f ? 1.f : 0.2f));
}
extern "C" int s2e_lrelu_bwd(int dtype, const void* gy, const void* y, void* gx, long n, void* stream) {
    if (!gy || !y || !gx || n <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_lrelu_bwd: bad argument");
    if (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gx) & 15) S2E_FAIL(S2E_ERR_ARG, "s2e_lrelu_bwd: pointers must be 16-byte aligned");
    const long nv = n / 4 + 1;
    const int grid = (int)((nv + 255) / 256 < 8192 ? (nv + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) lrelu_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)gy, (const bf16_t*)y, (bf16_t*)gx, n);
    else if (dtype == S2E_F32) lrelu_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)gy, (const float*)y, (float*)gx, n);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_lrelu_bwd: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("lrelu_bwd_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ bilinear resize (encoder front end)
// F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False) of single-channel images (encoder.py:55: every style
// image is resized to 256 x 256 before the first conv): source coordinate (d + 0.5) * (in / out) - 0.5 clamped at 0, taps
// i0 = floor, i1 = min(i0 + 1, in - 1), weights (1 - l, l) -- torch's area_pixel_compute_source_index rule, fp32 (opmath).
// x: (N, H, W) fp32; y: (N, Ho, Wo) of T (the compute dtype the first conv consumes; (N,Ho,Wo,1) NHWC is the same memory).
struct BilTap { int i0, i1; float l; };
__device__ __forceinline__ BilTap bil_tap(int d, float scale, int n_in) {
    float f = ((float)d + 0.5f) * scale - 0.5f;
    f = f < 0.f ? 0.f : f;
    BilTap t;
    t.i0 = (int)f;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l = f - (float)t.i0;
    return t;
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, int H, int W, int Ho, int Wo,
                                                           float sy, float sx) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
    if (ox >= Wo) return;
    const BilTap ty = bil_tap(oy, sy, H), tx = bil_tap(ox, sx, W);
    const float* p = x + (size_t)n * H * W;
    const float v = (1.f - ty.l) * ((1.f - tx.l) * p[(size_t)ty.i0 * W + tx.i0] + tx.l * p[(size_t)ty.i0 * W + tx.i1])
                  + ty.l * ((1.f - tx.l) * p[(size_t)ty.i1 * W + tx.i0] + tx.l * p[(size_t)ty.i1 * W + tx.i1]);
    store1<T>(y + ((size_t)n * Ho + oy) * Wo + ox, v);
}
// gx (N, H, W) fp32, ZERO-FILLED by the caller, += the four taps of every output pixel's gradient (fp32 atomics: the images
// are tiny -- 65536 outputs each -- and a gather form would need the inverse tap table of a non-integer scale)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ gy, float* __restrict__ gx, int H, int W, int Ho, int Wo,
                                                           float sy, float sx) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
    if (ox >= Wo) return;
    const BilTap ty = bil_tap(oy, sy, H), tx = bil_tap(ox, sx, W);
    const float g = load1<T>(gy + ((size_t)n * Ho + oy) * Wo + ox);
    float* p = gx + (size_t)n * H * W;
    atomicAdd(p + (size_t)ty.i0 * W + tx.i0, g * (1.f - ty.l) * (1.f - tx.l));
    atomicAdd(p + (size_t)ty.i0 * W + tx.i1, g * (1.f - ty.l) * tx.l);
    atomicAdd(p + (size_t)ty.i1 * W + tx.i0, g * ty.l * (1.f - tx.l));
    atomicAdd(p + (size_t)ty.i1 * W + tx.i1, g * ty.l * tx.l);
}

static int bilinear_launch(int dtype, const void* a, void* b, int N, int H, int W, int Ho, int Wo, void* stream, bool fwd, const char* name) {
    if (!a || !b || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) S2E_FAIL(S2E_ERR_ARG, "%s: bad argument", name);
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "%s: bad dtype %d", name, dtype);
    if (Ho > 65535 || N > 65535) S2E_FAIL(S2E_ERR_UNSUPPORTED, "%s: grid too large", name);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ceil_div(Wo, 256), Ho, N);
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    if (fwd) {
        if (dtype == S2E_BF16) bilinear_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const float*)a, (bf16_t*)b, H, W, Ho, Wo, sy, sx);
        else bilinear_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, H, W, Ho, Wo, sy, sx);
    } else {
        if (dtype == S2E_BF16) bilinear_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, (float*)b, H, W, Ho, Wo, sy, sx);
        else bilinear_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, (float*)b, H, W, Ho, Wo, sy, sx);
    }
    S2E_CHECK_LAUNCH(name);
    return S2E_OK;
}
extern "C" int s2e_bilinear_resize_fwd(int dtype, const float* x, void* y, int N, int H, int W, int Ho, int Wo, void* stream) {
    return bilinear_launch(dtype, x, y, N, H, W, Ho, Wo, stream, true, "s2e_bilinear_resize_fwd");
}
extern "C" int s2e_bilinear_resize_bwd(int dtype, const void* gy, float* gx, int N, int H, int W, int Ho, int Wo, void* stream) {
    return bilinear_launch(dtype, gy, gx, N, H, W, Ho, Wo, stream, false, "s2e_bilinear_resize_bwd");
}

// ------------------------------------------------------------------------------------ label-uniform rectangles (SPADE sparsity)
// gamma and beta at a pixel depend on the labels of its 5x5 neighbourhood only (one-hot -> conv3x3 -> ReLU -> conv3x3,
// normalization.py:97-101).  A rectangle of th x tw pixels of the (nearest-downsampled) h x w label map whose pixels AND the
// in-image pixels of its 2-pixel halo all carry one class c takes gamma / beta from a per-class table instead of the conv
// (s2e_spade_modulate_uniform); the others are listed for the dense launch (s2e_spade_conv_modulate_sparse).
// Two launches: (1) one WAVE per rectangle reads the rectangle's pixels and in-image 2-pixel halo (<= ~550 labels, a few
// independent byte loads per lane; a thread-per-rectangle loop with an early exit serialised 400 global-load latencies per
// launch) and reduces "all equal to the first"; (2) one block compacts the flags into the two lists in rectangle order
// (rectangle r = (n * tiles_y + ty) * tiles_x + tx, the fused launch's order) -- deterministic, and neighbouring dense
// rectangles stay neighbours.  cls[r] = class, or 255 for a dense rectangle; counts = {dense, uniform}.
__global__ __launch_bounds__(256) void label_rect_classify_kernel(const uint8_t* __restrict__ label, int N, int H, int W, int h, int w, int tw, int th,
        int tiles_x, int tiles_y, uint8_t* __restrict__ cls) {
    const int per = tiles_x * tiles_y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= N * per) return;
    const int n = r / per, rr = r - n * per;
    const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
    const int sy = H / h, sx = W / w;
    const uint8_t* lb = label + (size_t)n * H * W;
    const int y0 = max(0, ty * th - 2), y1 = min(h, ty * th + th + 2), x0 = max(0, tx * tw - 2), x1 = min(w, tx * tw + tw + 2);
    const int rw = x1 - x0, cnt = rw * (y1 - y0);
    const int first = lb[(size_t)y0 * sy * W + (size_t)x0 * sx];
    int diff = 0;
    for (int i = lane; i < cnt; i += 64) {
        const int yy = i / rw, xx = i - yy * rw;
        diff |= (lb[(size_t)(y0 + yy) * sy * W + (size_t)(x0 + xx) * sx] != first);
    }
    const bool any = __ballot(diff != 0) != 0ull;
    if (lane == 0) cls[r] = any ? (uint8_t)255 : (uint8_t)first;
}

__global__ __launch_bounds__(1024) void label_rect_compact_kernel(const uint8_t* __restrict__ cls, int total, int* __restrict__ dense_list,
        int* __restrict__ uni_list, int* __restrict__ counts) {
    __shared__ int wsum[2][16];
    __shared__ int base[2];
    if (threadIdx.x == 0) { base[0] = 0; base[1] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = 0; r0 < total; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        const bool in = r < total;
        const bool isd = in && cls[r] == 255, isu = in && cls[r] != 255;
        const unsigned long long bd = __ballot(isd), bu = __ballot(isu);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int pd = __popcll(bd & below), pu = __popcll(bu & below);
        if (lane == 0) { wsum[0][wave] = __popcll(bd); wsum[1][wave] = __popcll(bu); }
        __syncthreads();
        int od = base[0], ou = base[1];
        for (int k = 0; k < wave; ++k) { od += wsum[0][k]; ou += wsum[1][k]; }
        if (isd) dense_list[od + pd] = r;
        if (isu) uni_list[ou + pu] = r;
        __syncthreads();
        if (threadIdx.x == 0) { for (int k = 0; k < 16; ++k) { base[0] += wsum[0][k]; base[1] += wsum[1][k]; } }
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = base[0]; counts[1] = base[1]; }
}

extern "C" int s2e_label_rect_classify(const uint8_t* label, int N, int H, int W, int h, int w, int tw, int th,
                                       uint8_t* cls, int* dense_list, int* uni_list, int* counts, void* stream) {
    if (!label || !cls || !dense_list || !uni_list || !counts || N <= 0 || h <= 0 || w <= 0 || tw <= 0 || th <= 0)
        S2E_FAIL(S2E_ERR_ARG, "s2e_label_rect_classify: bad argument");
    if (H % h || W % w) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_label_rect_classify: %dx%d is not an integer multiple of %dx%d", H, W, h, w);
    const int tiles_x = ceil_div(w, tw), tiles_y = ceil_div(h, th);
    const int total = N * tiles_x * tiles_y;
    hipStream_t st = (hipStream_t)stream;
    label_rect_classify_kernel<<<ceil_div(total, 4), 256, 0, st>>>(label, N, H, W, h, w, tw, th, tiles_x, tiles_y, cls);
    label_rect_compact_kernel<<<1, 1024, 0, st>>>(cls, total, dense_list, uni_list, counts);
    S2E_CHECK_LAUNCH("label_rect_classify kernels");
    return S2E_OK;
}

// The per-class table of a SPADE: gamma | beta (with their bias) of a map that is ONE class c everywhere, for every position
// class (cy, cx) in {0, 1, interior, H-2, H-1}^2 -- 4 x 25 vectors of 2C floats.  With A[m] = ReLU(b_sh + sum of the in-image taps
// of w_sh[:, c]) the activation at a pixel whose 3x3 window is cut by the image border as m = (my, mx) in {low, none, high}^2
// (rounded to the activation dtype, as the dense path stores it),
//     T[c][cy][cx][co] = b[co] + sum_{taps (ty,tx) inside the image at (cy,cx)}  W[co][ty][tx][:] . A[m(cy,ty)][m(cx,tx)][:]
// with W the PACKED weight the conv launches use (same bf16 values); fp32 out.
// Work split: a block = one class x 4 output rows; a WAVE takes one row with its lanes along the 128 hidden channels
// (coalesced 128-B reads of the packed row, A from LDS), accumulates the 25 position classes per lane and folds the 64 lanes with
// DPP adds at the end.  (One thread per output row walking its row alone -- 384 dependent 2-byte loads -- took 42 us per launch,
// 0.5 ms per step, for 60 MFLOP.)
template <typename T>
__device__ __forceinline__ void spade_class_table_body(const float* __restrict__ w_sh, const float* __restrict__ b_sh,
        const T* __restrict__ wq, const float* __restrict__ bias, float* __restrict__ table, int ncls, int nh, int C2, int kpad,
        int bx, int c, float (*A)[128]) {
    constexpr int ROWS = 4;                                  // one row per wave: its 18 weight loads are one latency round
    const int co0 = bx * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 9 * 128; i += 256) if ((i & 127) >= nh) A[i >> 7][i & 127] = 0.f;       // nh < 128: the unused columns
    for (int ci = tid; ci < nh; ci += 256) {                 // A[m][ci]: the nine taps of (ci, c) loaded once, all in flight
        const float* wr = w_sh + ((size_t)ci * ncls + c) * 9;
        float w9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[t] = wr[t];
        const float b = b_sh[ci];
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const int my = m / 3, mx = m - my * 3;
            float a = b;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t - ty * 3;
                const bool ok = !((my == 0 && ty == 0) || (my == 2 && ty == 2) || (mx == 0 && tx == 0) || (mx == 2 && tx == 2));
                if (ok) a += w9[t];
            }
            A[m][ci] = (float)(T)fmaxf(a, 0.f);               // the activation as the dense path stores it
        }
    }
    __syncthreads();
    for (int r = wave; r < ROWS; r += 4) {
        const int co = co0 + r;
        if (co >= C2) break;                                 // wave-uniform
        float t25[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) t25[k] = 0.f;
        const T* wrow = wq + (size_t)co * kpad;
        float wv[9][2];                                      // this lane's channels lane, lane + 64 (nh <= 128) of the nine taps
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 2; ++j) wv[tap][j] = lane + 64 * j < nh ? load1<T>(wrow + tap * nh + lane + 64 * j) : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float v[9];
#pragma unroll
            for (int m = 0; m < 9; ++m) v[m] = wv[tap][0] * A[m][lane] + wv[tap][1] * A[m][lane + 64];
            const int ty = tap / 3, tx = tap - ty * 3;       // tap offset (ty - 1, tx - 1); compile-time under the unroll
#pragma unroll
            for (int cy = 0; cy < 5; ++cy) {
                if ((cy == 0 && ty == 0) || (cy == 4 && ty == 2)) continue;                       // neighbour row outside the image
                const int my = ((cy == 0 && ty == 1) || (cy == 1 && ty == 0)) ? 0 : (((cy == 4 && ty == 1) || (cy == 3 && ty == 2)) ? 2 : 1);
#pragma unroll
                for (int cx = 0; cx < 5; ++cx) {
                    if ((cx == 0 && tx == 0) || (cx == 4 && tx == 2)) continue;
                    const int mx = ((cx == 0 && tx == 1) || (cx == 1 && tx == 0)) ? 0 : (((cx == 4 && tx == 1) || (cx == 3 && tx == 2)) ? 2 : 1);
                    t25[cy * 5 + cx] += v[my * 3 + mx];
                }
            }
        }
        const float b = bias ? bias[co] : 0.f;
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const float tot = wave_sum_last(t25[k]);         // valid in lane 63
            if (lane == 63) table[((size_t)c * 25 + k) * C2 + co] = tot + b;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void spade_class_table_kernel(const float* __restrict__ w_sh, const float* __restrict__ b_sh,
        const T* __restrict__ wq, const float* __restrict__ bias, float* __restrict__ table, int ncls, int nh, int C2, int kpad) {
    __shared__ float A[9][128];
    spade_class_table_body<T>(w_sh, b_sh, wq, bias, table, ncls, nh, C2, kpad, blockIdx.x, blockIdx.y, A);
}
// the tables of all label-sparse SPADE layers of a forward in one launch: block_map[b] = {job, bx}; grid.y = class
template <typename T>
__global__ __launch_bounds__(256) void spade_class_table_batch_kernel(const s2e_class_table_job* __restrict__ jobs,
        const int* __restrict__ block_map, char* __restrict__ table_base, int ncls, int bk) {
    __shared__ float A[9][128];
    const int* bm = block_map + 2 * blockIdx.x;
    const s2e_class_table_job J = jobs[bm[0]];
    const int kpad = (9 * J.nh + bk - 1) / bk * bk;
    spade_class_table_body<T>(J.w_sh, J.b_sh, (const T*)J.w_packed, J.bias, (float*)(table_base + J.table_off), ncls, J.nh, 2 * J.C, kpad,
                              bm[1], blockIdx.y, A);
}

extern "C" int s2e_spade_class_table(int dtype, const float* w_sh, const float* b_sh, const void* w_packed, const float* bias,
                                     float* table, int ncls, int nh, int C, void* stream) {
    if (!w_sh || !b_sh || !w_packed || !table || ncls <= 0 || ncls > 8 || nh <= 0 || nh > 128 || C <= 0)
        S2E_FAIL(S2E_ERR_ARG, "s2e_spade_class_table: bad argument (nh <= 128)");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_class_table: bad dtype %d", dtype);
    const int bk = dtype == S2E_BF16 ? 64 : 32;
    const int kpad = ceil_div(9 * nh, bk) * bk;
    const dim3 grid(ceil_div(2 * C, 4), ncls);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) spade_class_table_kernel<bf16_t><<<grid, 256, 0, st>>>(w_sh, b_sh, (const bf16_t*)w_packed, bias, table, ncls, nh, 2 * C, kpad);
    else spade_class_table_kernel<float><<<grid, 256, 0, st>>>(w_sh, b_sh, (const float*)w_packed, bias, table, ncls, nh, 2 * C, kpad);
    S2E_CHECK_LAUNCH("spade_class_table_kernel");
    return S2E_OK;
}

extern "C" long s2e_class_table_block_map(const s2e_class_table_job* jobs_host, int n_jobs, int* block_map_host) {
    if (!jobs_host || n_jobs < 0) return S2E_ERR_ARG;
    long nb = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (jobs_host[j].C <= 0 || jobs_host[j].nh <= 0 || jobs_host[j].nh > 128) return S2E_ERR_ARG;
        const int gx = ceil_div(2 * jobs_host[j].C, 4);
        for (int b = 0; b < gx; ++b, ++nb)
            if (block_map_host) { block_map_host[2 * nb] = j; block_map_host[2 * nb + 1] = b; }
    }
    return nb;
}
extern "C" int s2e_spade_class_table_batch(int dtype, const s2e_class_table_job* jobs, const int* block_map, int n_blocks,
                                           void* table_base, int ncls, void* stream) {
    if (!jobs || !block_map || !table_base || n_blocks <= 0 || ncls <= 0 || ncls > 8)
        S2E_FAIL(S2E_ERR_ARG, "s2e_spade_class_table_batch: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_class_table_batch: bad dtype %d", dtype);
    const int bk = dtype == S2E_BF16 ? 64 : 32;
    const dim3 grid(n_blocks, ncls);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) spade_class_table_batch_kernel<bf16_t><<<grid, 256, 0, st>>>(jobs, block_map, (char*)table_base, ncls, bk);
    else spade_class_table_batch_kernel<float><<<grid, 256, 0, st>>>(jobs, block_map, (char*)table_base, ncls, bk);
    S2E_CHECK_LAUNCH("spade_class_table_batch_kernel");
    return S2E_OK;
}

// SPADE+Style modulation of the label-uniform rectangles: out = [lrelu] 0.5*((x-mean)*rstd*(1+gamma)+beta + x*(1+s0)+s1) with
// gamma | beta = table[class][cy][cx][0..2C) (fp32, s2e_spade_class_table) where (cy, cx) in 0..4 is the pixel's position class
// {0, 1, interior, H-2, H-1}.  A block takes rectangles uni_list[blockIdx.x], [+gridDim.x], ...; a thread keeps one 16-byte
// channel group, so its constants are loaded once per rectangle (16-byte loads), and walks the rectangle's pixels four at a
// time without a division.  gamma_out as s2e_spade_conv_modulate.  (Measured: 2.7 TB/s of x / out / gamma traffic.  A
// version that streams the whole tensor linearly and skips the dense rectangles' vectors -- per-vector class lookup, all
// classes' interior rows in registers -- was slower: 1.0 vs 0.59 ms per step.)
template <typename T>
__global__ __launch_bounds__(256) void spade_modulate_uniform_kernel(const T* __restrict__ x, const float* __restrict__ stats,
        const float* __restrict__ style, int sld, const float* __restrict__ table, const uint8_t* __restrict__ cls,
        const int* __restrict__ uni_list, const int* __restrict__ counts, T* __restrict__ out, T* __restrict__ gout,
        int H, int W, int C, int tw, int th, int tiles_x, int tiles_y, int lrelu, int x_up) {
    constexpr int VEC = Vec<T>::N;
    const int cg = C / VEC;                                  // channel groups per pixel
    const int n_uni = counts[1];
    // x_up: x is (N, H/2, W/2, C) and the block's nearest 2x upsampling is folded into the read
    auto xoff = [&](int n, int y, int xx) __attribute__((always_inline)) -> size_t {
        return x_up ? ((size_t)(n * (H >> 1) + (y >> 1)) * (W >> 1) + (xx >> 1)) * C : ((size_t)(n * H + y) * W + xx) * C;
    };
    const int lanes_c = cg < 256 ? cg : 256;                 // channel groups fastest: a wave's accesses are contiguous rows
    const int pstep = 256 / lanes_c, prow = threadIdx.x / lanes_c;
    for (int li = blockIdx.x; li < n_uni; li += gridDim.x) {
        const int r = uni_list[li];
        const int c = cls[r];
        const int tx0 = r % tiles_x, ty0 = (r / tiles_x) % tiles_y, n = r / (tiles_x * tiles_y);
        const int y0 = ty0 * th, x0 = tx0 * tw;
        const bool inner = y0 >= 2 && x0 >= 2 && y0 + th <= H - 2 && x0 + tw <= W - 2;    // no pixel within two of the border
        for (int g = threadIdx.x % lanes_c; g < cg; g += lanes_c) {
            if (prow >= pstep) break;
            const int c0 = g * VEC;
            float mu[VEC], rs[VEC], sa[VEC], sb[VEC], gi[VEC], bi[VEC];
            const f32x4_t* stp = (const f32x4_t*)(stats + ((size_t)n * C + c0) * 2);
            const f32x4_t* s0p = (const f32x4_t*)(style + (size_t)n * sld + c0);
            const f32x4_t* s1p = (const f32x4_t*)(style + (size_t)n * sld + C + c0);
            const float* tint = table + ((size_t)c * 25 + 12) * 2 * C;                 // interior case (cy, cx) = (2, 2)
            const f32x4_t* tgp = (const f32x4_t*)(tint + c0);
            const f32x4_t* tbp = (const f32x4_t*)(tint + C + c0);
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const f32x4_t a0 = stp[j / 2], a1 = stp[j / 2 + 1], q0 = s0p[j / 4], q1 = s1p[j / 4], g4 = tgp[j / 4], b4 = tbp[j / 4];
                mu[j] = a0[0]; rs[j] = a0[1]; mu[j + 1] = a0[2]; rs[j + 1] = a0[3]; mu[j + 2] = a1[0]; rs[j + 2] = a1[1]; mu[j + 3] = a1[2]; rs[j + 3] = a1[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) { sa[j + i] = 1.f + q0[i]; sb[j + i] = q1[i]; gi[j + i] = g4[i]; bi[j + i] = b4[i]; }
            }
            auto one = [&](int y, int xx, u32x4_t raw) __attribute__((always_inline)) {
                float ga[VEC], be[VEC];
                const int cy = y < 2 ? y : (y >= H - 2 ? 4 - (H - 1 - y) : 2), cx = xx < 2 ? xx : (xx >= W - 2 ? 4 - (W - 1 - xx) : 2);
                if (inner || (cy == 2 && cx == 2)) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { ga[j] = gi[j]; be[j] = bi[j]; }
                } else {                                     // within two pixels of the image border: its own table row
                    const float* tr = table + ((size_t)c * 25 + cy * 5 + cx) * 2 * C;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { ga[j] = tr[c0 + j]; be[j] = tr[C + c0 + j]; }
                }
                const size_t o = ((size_t)(n * H + y) * W + xx) * C + c0;
                float f[VEC], v[VEC];
                unpack16<T>(raw, f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (f[j] - mu[j]) * rs[j];
                    v[j] = 0.5f * (xh * (1.f + ga[j]) + be[j] + f[j] * sa[j] + sb[j]);
                    if (lrelu) v[j] = lrelu02(v[j]);
                }
                *(u32x4_t*)(out + o) = pack16<T>(v);
                if (gout) *(u32x4_t*)(gout + o) = pack16<T>(ga);
            };
            // pixels prow, prow + pstep, ... of the rectangle (row-major), four in flight; (py, px) advance without a division
            const int npix = tw * th;
            const int dy = pstep / tw, dx = pstep - dy * tw;
            int py = prow / tw, px = prow - py * tw;
            auto step = [&]() __attribute__((always_inline)) { px += dx; py += dy; if (px >= tw) { px -= tw; ++py; } };
            int pp = prow;
            for (; pp + 3 * pstep < npix; pp += 4 * pstep) {
                int yy[4], xs[4]; u32x4_t raw[4]; bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    yy[k] = y0 + py; xs[k] = x0 + px;
                    step();
                    ok[k] = yy[k] < H && xs[k] < W;
                    raw[k] = u32x4_t{0u, 0u, 0u, 0u};
                    if (ok[k]) raw[k] = *(const u32x4_t*)(x + xoff(n, yy[k], xs[k]) + c0);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) if (ok[k]) one(yy[k], xs[k], raw[k]);
            }
            for (; pp < npix; pp += pstep) {
                const int y = y0 + py, xx = x0 + px;
                step();
                if (y < H && xx < W) one(y, xx, *(const u32x4_t*)(x + xoff(n, y, xx) + c0));
            }
        }
    }
}

extern "C" int s2e_spade_modulate_uniform(int dtype, const void* x, const float* stats, const float* style, int style_ld,
                                          const float* table, const uint8_t* cls, const int* uni_list, const int* counts,
                                          void* out, void* gamma_out, int N, int H, int W, int C, int tw, int th, int lrelu, int x_up,
                                          void* stream) {
    if (!x || !stats || !style || !table || !cls || !uni_list || !counts || !out || N <= 0 || H < 5 || W < 5 || C <= 0 || tw <= 0 || th <= 0)
        S2E_FAIL(S2E_ERR_ARG, "s2e_spade_modulate_uniform: bad argument");
    if (x_up && ((H | W) & 1)) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_modulate_uniform: x_up needs even H, W");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_modulate_uniform: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_spade_modulate_uniform: C=%d not a multiple of %d", C, vec);
    if ((((uintptr_t)stats | (uintptr_t)style | (uintptr_t)table) & 15) || (style_ld & 3) || (C & 3))
        S2E_FAIL(S2E_ERR_ARG, "s2e_spade_modulate_uniform: stats, style and table must be 16-byte aligned (style_ld, C multiples of 4)");
    const int tiles_x = ceil_div(W, tw), tiles_y = ceil_div(H, th);
    const long rects = (long)N * tiles_x * tiles_y;
    const int grid = (int)(rects < 2048 ? rects : 2048);
    const int sld = style_ld > 0 ? style_ld : 2 * C;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16)
        spade_modulate_uniform_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, stats, style, sld, table, cls, uni_list, counts,
                                                                  (bf16_t*)out, (bf16_t*)gamma_out, H, W, C, tw, th, tiles_x, tiles_y, lrelu, x_up);
    else
        spade_modulate_uniform_kernel<float><<<grid, 256, 0, st>>>((const float*)x, stats, style, sld, table, cls, uni_list, counts,
                                                                 (float*)out, (float*)gamma_out, H, W, C, tw, th, tiles_x, tiles_y, lrelu, x_up);
    S2E_CHECK_LAUNCH("spade_modulate_uniform_kernel");
    return S2E_OK;
}
