// Label-sparse SPADE BACKWARD (round 4; forward: DESIGN 3.1d).
//
// gamma | beta at a pixel depend on the labels of its 5x5 neighbourhood only (one-hot -> conv3x3 -> ReLU -> conv3x3,
// normalization.py:97-103 of the reference).  In a rectangle whose pixels and 2-pixel halo carry ONE class c -- and whose halo lies
// inside the image ("uniform-interior") -- the hidden activation is one vector a_c, so the gradients the branch's backward needs
// from those pixels collapse to sums of d[gamma | beta]:
//   data gradient  d actv[q][k] = sum_{t, co} W[co][k][t] * dgb[q - t + 1][co]   is only ever used for mlp_shared's weight / bias
//   gradient, and for q in such a rectangle every tap of mlp_shared sees class c, so all nine taps (and the bias) receive
//        m_c[k] * sum_q d actv[q][k] = m_c[k] * sum_{t, co} W[co][k][t] * R_c[t][co],   R_c[t][co] = sum_{q in U_c} dgb[q - t + 1][co]
//   (m_c = [a_c > 0], the ReLU mask): nine SHIFTED sums of dgb per rectangle instead of a 2C -> 128 transposed convolution over it;
//   weight gradient of the [gamma | beta] conv  dW[co][k][t] += sum_q dgb[q][co] * actv[q + t - 1][k] = R_c[centre][co] * a_c[k]
//   for every tap: a rank-1 update per class.
// The convolutions then run on the rectangles that cross a label boundary (or touch the image border) only.
//   s2e_label_rect_lists_bwd   : from the forward's classification, the rectangles the backward still convolves (dense, or uniform
//                                on the image border) and the uniform-interior ones; device-side counts (hipGraph replays follow
//                                label maps that change).
//   s2e_spade_uniform_sums     : R[c][t][co] += the nine shifted sums of dgb over each uniform-interior rectangle (fp32 atomics).
//   s2e_spade_uniform_grads    : all queued layers in two launches: A_c[k] = sum W . R_c (atomics over channel slices), then
//                                dW_sh[k][c][*] += m_c A_c, db_sh += ..., and (when asked) the rank-1 update of [dW_gamma; dW_beta]
//                                and the bias sums.
#include "common.h"

namespace {

// (S2E_UNI_REPLICAS: include/seg2eye_hip.h)

__global__ __launch_bounds__(1024) void rect_lists_bwd_kernel(const uint8_t* __restrict__ cls, int total, int tiles_y, int tiles_x,
        int* __restrict__ work_list, int* __restrict__ ui_list, int* __restrict__ counts) {
    __shared__ int wsum[2][16];
    __shared__ int base[2];
    if (threadIdx.x == 0) { base[0] = 0; base[1] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = tiles_y * tiles_x;
    for (int r0 = 0; r0 < total; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        const bool in = r < total;
        bool ui = false;
        if (in && cls[r] != 255) {
            const int rr = r % per, ty = rr / tiles_x, tx = rr - ty * tiles_x;
            ui = ty > 0 && ty < tiles_y - 1 && tx > 0 && tx < tiles_x - 1;
        }
        const bool wk = in && !ui;
        const unsigned long long bw = __ballot(wk), bu = __ballot(ui);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int pw = __popcll(bw & below), pu = __popcll(bu & below);
        if (lane == 0) { wsum[0][wave] = __popcll(bw); wsum[1][wave] = __popcll(bu); }
        __syncthreads();
        int ow = base[0], ou = base[1];
        for (int k = 0; k < wave; ++k) { ow += wsum[0][k]; ou += wsum[1][k]; }
        if (wk) work_list[ow + pw] = r;
        if (ui) ui_list[ou + pu] = r;
        __syncthreads();
        if (threadIdx.x == 0) { for (int k = 0; k < 16; ++k) { base[0] += wsum[0][k]; base[1] += wsum[1][k]; } }
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = base[0]; counts[1] = base[1]; }
}

// One block per uniform-interior 16 x 16 rectangle.  Thread = (4-channel group g, row slice s): the slice walks rows s, s + S, ...
// of the rectangle's 18 x 18 neighbourhood; per row the three sliding 16-pixel sums; row y feeds the window sums dy in
// [max(0, y - 15), min(2, y)].  Slices are folded through LDS, then one atomic per (tap, channel) into R[class].
template <typename T>
__global__ __launch_bounds__(256) void spade_uniform_sums_kernel(const T* __restrict__ dgb, int H, int W, int C2, int tiles_y, int tiles_x,
        const uint8_t* __restrict__ cls, const int* __restrict__ ui_list, const int* __restrict__ counts, float* __restrict__ R, int ncls) {
    __shared__ float part[9][1024];                          // [window][slice * C2 + c]: slices * C2 = 1024 floats
    const int ri = blockIdx.x;
    if (ri >= counts[1]) return;
    const int r = ui_list[ri];
    const int c = cls[r];
    const int per = tiles_y * tiles_x;
    const int n = r / per, rr = r - n * per, ty = rr / tiles_x, tx = rr - ty * tiles_x;
    const int groups = C2 >> 2, slices = 256 / groups;       // C2 in {128, 256, 512, 1024}: 8 / 4 / 2 / 1 slices
    const int g = threadIdx.x % groups, s = threadIdx.x / groups;
    float S[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) S[a][b][j] = 0.f;
    const T* base = dgb + ((size_t)(n * H + ty * 16 - 1) * W + tx * 16 - 1) * C2 + g * 4;
    for (int y = s; y < 18; y += slices) {
        const T* row = base + (size_t)y * W * C2;
        float v[18][4];
#pragma unroll
        for (int x = 0; x < 18; ++x) {
            if constexpr (sizeof(T) == 2) {
                const u32x2_t w = *(const u32x2_t*)(row + (size_t)x * C2);
                v[x][0] = bf16_bits_to_f32(w[0] & 0xffffu); v[x][1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
                v[x][2] = bf16_bits_to_f32(w[1] & 0xffffu); v[x][3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
            } else {
                const f32x4_t w = *(const f32x4_t*)(row + (size_t)x * C2);
                v[x][0] = w[0]; v[x][1] = w[1]; v[x][2] = w[2]; v[x][3] = w[3];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float rs0 = 0.f;
#pragma unroll
            for (int x = 0; x < 16; ++x) rs0 += v[x][j];
            const float rs1 = rs0 - v[0][j] + v[16][j], rs2 = rs1 - v[1][j] + v[17][j];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                if (y >= dy && y <= dy + 15) { S[dy][0][j] += rs0; S[dy][1][j] += rs1; S[dy][2][j] += rs2; }
        }
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[dy * 3 + dx][s * C2 + g * 4 + j] = S[dy][dx][j];
    __syncthreads();
    // window (dy, dx) = the rectangle shifted by (dy - 1, dx - 1) = q - t + 1 for tap t = (2 - dy, 2 - dx)
    for (int i = threadIdx.x; i < 9 * C2; i += 256) {
        const int w = i / C2, ch = i - w * C2;
        float a = 0.f;
        for (int k = 0; k < slices; ++k) a += part[w][k * C2 + ch];
        const int dy = w / 3, dx = w - dy * 3, t = (2 - dy) * 3 + (2 - dx);
        atomicAdd(R + (((size_t)(ri & (S2E_UNI_REPLICAS - 1)) * ncls + c) * 9 + t) * C2 + ch, a);   // (replicas: ~1000 rectangles add to 4 class slots)
    }
}

struct UniJobs { s2e_spade_uni_job j[16]; };

// A[cls][k] += sum over this block's 8 output channels co and the 9 taps of W[co][k][t] * R[cls][t][co]
__global__ __launch_bounds__(512) void spade_uni_gemv_kernel(const UniJobs jobs) {
    const s2e_spade_uni_job& J = jobs.j[blockIdx.y];
    const int co0 = blockIdx.x * 8;
    if (co0 >= J.C2) return;
    __shared__ float Rs[4][9][8];                            // the replicas of this block's 8 channels, folded
    const int co1 = min(J.C2, co0 + 8);
    for (int i = threadIdx.x; i < J.ncls * 9 * 8; i += 512) {
        const int c = i / 72, t = (i / 8) % 9, j = i & 7;
        float a = 0.f;
        if (co0 + j < co1)
            for (int rep = 0; rep < S2E_UNI_REPLICAS; ++rep) a += J.R[(((size_t)rep * J.ncls + c) * 9 + t) * J.C2 + co0 + j];
        Rs[c][t][j] = a;
    }
    __syncthreads();
    const int k = threadIdx.x % J.nh, c = threadIdx.x / J.nh;
    if (c >= J.ncls) return;
    float acc = 0.f;
    for (int co = co0; co < co1; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            acc += J.w_gb[(size_t)co * J.w_sc + (size_t)k * J.w_sk + (size_t)t * J.w_st] * Rs[c][t][co - co0];
    atomicAdd(J.A + c * J.nh + k, acc);
}

// per layer: mlp_shared's gradients from A; optionally the rank-1 update of the [gamma | beta] conv's weight gradient and its bias sums
__global__ __launch_bounds__(256) void spade_uni_apply_kernel(const UniJobs jobs) {
    const s2e_spade_uni_job& J = jobs.j[blockIdx.y];
    __shared__ float a_c[4][128];
    const int tid = threadIdx.x;
    for (int i = tid; i < J.ncls * J.nh; i += 256) {
        const int c = i / J.nh, k = i - c * J.nh;
        float a = J.b_sh[k];
        for (int t = 0; t < 9; ++t) a += J.w_sh[((size_t)k * J.ncls + c) * 9 + t];
        a_c[c][k] = fmaxf(J.act_bf16 ? (float)(bf16_t)a : a, 0.f);
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int k = tid; k < J.nh; k += 256) {
            float bsum = 0.f;
            for (int c = 0; c < J.ncls; ++c) {
                const float v = a_c[c][k] > 0.f ? J.A[c * J.nh + k] : 0.f;
                bsum += v;
                if (J.dw_sh) for (int t = 0; t < 9; ++t) J.dw_sh[((size_t)k * J.ncls + c) * 9 + t] += v;
            }
            if (J.db_sh) J.db_sh[k] += bsum;
        }
    }
    if (!J.dw_gb) return;
    // [dW_gamma; dW_beta][co][k][t] += sum_c R[c][centre][co] * a_c[k] (every tap);  db[co] += sum_c R[c][centre][co]
    const int co0 = blockIdx.x * 8, co1 = min(J.C2, co0 + 8);
    if (co0 >= J.C2) return;
    __shared__ float rc[4][8];                               // R[c][centre tap][co0 .. co0 + 8), replicas folded
    for (int i = tid; i < J.ncls * (co1 - co0); i += 256) {
        const int c = i / (co1 - co0), j = i - c * (co1 - co0);
        float a = 0.f;
        for (int rep = 0; rep < S2E_UNI_REPLICAS; ++rep) a += J.R[(((size_t)rep * J.ncls + c) * 9 + 4) * J.C2 + co0 + j];
        rc[c][j] = a;
    }
    __syncthreads();
    for (int i = tid; i < (co1 - co0) * J.nh; i += 256) {
        const int co = co0 + i / J.nh, k = i % J.nh;
        float v = 0.f;
        for (int c = 0; c < J.ncls; ++c) v += rc[c][co - co0] * a_c[c][k];
        for (int t = 0; t < 9; ++t) J.dw_gb[(size_t)co * J.w_sc + (size_t)k * J.w_sk + (size_t)t * J.w_st] += v;
    }
    if (J.db_gb)
        for (int co = co0 + tid; co < co1; co += 256) {
            float v = 0.f;
            for (int c = 0; c < J.ncls; ++c) v += rc[c][co - co0];
            J.db_gb[co] += v;
        }
}

}  // namespace

extern "C" int s2e_label_rect_lists_bwd(const uint8_t* cls, int N, int tiles_y, int tiles_x, int* work_list, int* ui_list, int* counts,
                                        void* stream) {
    if (!cls || !work_list || !ui_list || !counts || N <= 0 || tiles_y <= 0 || tiles_x <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_label_rect_lists_bwd: bad argument");
    rect_lists_bwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(cls, N * tiles_y * tiles_x, tiles_y, tiles_x, work_list, ui_list, counts);
    S2E_CHECK_LAUNCH("rect_lists_bwd_kernel");
    return S2E_OK;
}

extern "C" int s2e_spade_uniform_sums(int dtype, const void* dgb, int N, int H, int W, int C2, int ncls, const uint8_t* cls, const int* ui_list,
                                      const int* counts, float* R, void* stream) {
    if (ncls <= 0 || ncls > 4) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_uniform_sums: ncls = %d", ncls);
    if (!dgb || !cls || !ui_list || !counts || !R) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_uniform_sums: null pointer");
    if (H % 16 || W % 16 || H < 48 || W < 48) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_spade_uniform_sums: %dx%d is not made of 16x16 rectangles with an interior", H, W);
    if (C2 != 128 && C2 != 256 && C2 != 512 && C2 != 1024) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_spade_uniform_sums: 2C = %d", C2);
    const int tiles_y = H / 16, tiles_x = W / 16;
    const int grid = N * (tiles_y - 2) * (tiles_x - 2);      // as many blocks as there can be uniform-interior rectangles
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) spade_uniform_sums_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)dgb, H, W, C2, tiles_y, tiles_x, cls, ui_list, counts, R, ncls);
    else if (dtype == S2E_F32) spade_uniform_sums_kernel<float><<<grid, 256, 0, st>>>((const float*)dgb, H, W, C2, tiles_y, tiles_x, cls, ui_list, counts, R, ncls);
    else S2E_FAIL(S2E_ERR_ARG, "s2e_spade_uniform_sums: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("spade_uniform_sums_kernel");
    return S2E_OK;
}

extern "C" int s2e_spade_uniform_grads(const s2e_spade_uni_job* jobs_host, int n_jobs, void* stream) {
    if (!jobs_host || n_jobs <= 0 || n_jobs > 16) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_uniform_grads: 1..16 jobs");
    UniJobs js{};
    int max_c2 = 0;
    bool any_gb = false;
    for (int i = 0; i < n_jobs; ++i) {
        js.j[i] = jobs_host[i];
        const s2e_spade_uni_job& J = jobs_host[i];
        if (!J.R || !J.A || !J.w_gb || !J.w_sh || !J.b_sh || J.nh <= 0 || J.nh > 128 || J.ncls <= 0 || J.ncls > 4 || J.C2 <= 0)
            S2E_FAIL(S2E_ERR_ARG, "s2e_spade_uniform_grads: bad job %d", i);
        if (J.C2 > max_c2) max_c2 = J.C2;
        any_gb = any_gb || J.dw_gb != nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    spade_uni_gemv_kernel<<<dim3(ceil_div(max_c2, 8), n_jobs), 512, 0, st>>>(js);
    S2E_CHECK_LAUNCH("spade_uni_gemv_kernel");
    // (8 rows of [dW_gamma; dW_beta] per block: at 64 rows a thread walked 288 dependent read-modify-writes, 132 us per launch)
    spade_uni_apply_kernel<<<dim3(any_gb ? (max_c2 + 7) / 8 : 1, n_jobs), 256, 0, st>>>(js);
    S2E_CHECK_LAUNCH("spade_uni_apply_kernel");
    return S2E_OK;
}
