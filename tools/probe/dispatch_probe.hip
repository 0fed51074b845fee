// Where do the workgroups of a 2-per-CU persistent grid land?  256 threads, 77 KB of LDS, grid = 2 x CUs: prints, per block,
// (XCC, SE, CU, SIMD, wave slot) of its wave 0 and whether blocks b and b + G/2 (or other pairings) share a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    __shared__ char pad[77 * 1024];
    pad[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    const long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);     // keep every block resident
    if (pad[(threadIdx.x * 7) & 255] == 77) out[0] = 0;
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int G = 2 * cus;
    unsigned* d; hipMalloc(&d, G * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, G * 8);
        probe<<<G, 256>>>(d, 20000);       // 200 us
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * G); hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> by_cu;
        for (int b = 0; b < G; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
            if (rep == 0 && b < 40) printf("b%3d hw=%08x xcc=%u se=%u sh=%u cu=%u simd=%u wave=%u\n", b, hw, xcc, se, sh, cu, (hw >> 4) & 3, hw & 0xf);
        }
        int two = 0, half_pairs = 0, other = 0; std::map<int, int> diffs;
        for (auto& kv : by_cu) { if (kv.second.size() == 2) { ++two; int dlt = kv.second[1] - kv.second[0]; diffs[dlt]++; if (dlt == G / 2) ++half_pairs; } else ++other; }
        printf("rep %d: %d CUs seen, %zu distinct ids, %d with exactly two blocks, %d of them (b, b+G/2), %d ids with another count\n", rep, cus, by_cu.size(), two, half_pairs, other);
        for (auto& kv : diffs) printf("   block-index difference %d: %d CUs\n", kv.first, kv.second);
    }
    return 0;
}
