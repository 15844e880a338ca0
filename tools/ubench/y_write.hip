// Dev microbenchmark: how fast does HBM take mac_kernel's Y stores, by store pattern?  Y[search][pair][8192 float4] (a pair
// = 128 KiB), S searches x P pairs.  Every kernel writes every float4 exactly once, non-temporal, one pair index per step.
//   pat 0  wave = 8 searches x 8 float4 (128-byte pieces), workgroup = 4 waves = 32 adjacent float4   (mac_kernel today)
//   pat 1  same wave, workgroup = 16 waves = 128 adjacent float4 (2 KiB per search per step)
//   pat 2  wave = 1 search x 64 float4 (1 KiB runs), workgroup = 4 waves = 4 searches
//   pat 3  wave = 1 search x 64 float4, workgroup = 4 waves = 256 adjacent float4 of ONE search (4 KiB runs, round 1)
//   pat 4  layout Y[item][pair][chunk][slot][8 float4]: wave = 8 searches x 8 float4 = one contiguous 1 KiB, workgroup 4 KiB
// hipcc --offload-arch=gfx950 -O3 y_write.hip -o y_write && ./y_write
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int S = 1024, P = 118, H = 8192;       // searches, pairs per search, float4 per pair

template <int PAT>
__global__ void wr(f4* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const f4 v = {1.f, 2.f, 3.f, (float)lane};
    if (PAT == 0 || PAT == 1 || PAT == 4) {
        const int chunks = H / (8 * nw);
        const int item = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
        const int slot = lane >> 3, fb = lane & 7;
        const int s = item * 8 + slot;
        for (int i = 0; i < P; ++i) {
            size_t e;
            if (PAT == 4) e = ((((size_t)item * P + i) * (H / 8) + (size_t)chunk * nw + wave) * 8 + slot) * 8 + fb;
            else e = ((size_t)s * P + i) * H + (size_t)chunk * 8 * nw + wave * 8 + fb;
            __builtin_nontemporal_store(v, y + e);
        }
    } else if (PAT == 2) {
        const int chunks = H / 64;
        const int item = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
        const int s = item * nw + wave;
        for (int i = 0; i < P; ++i) __builtin_nontemporal_store(v, y + ((size_t)s * P + i) * H + chunk * 64 + lane);
    } else {
        const int chunks = H / (64 * nw);
        const int s = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
        for (int i = 0; i < P; ++i) __builtin_nontemporal_store(v, y + ((size_t)s * P + i) * H + (chunk * nw + wave) * 64 + lane);
    }
}

template <class F>
static void timeit(const char* name, F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); launch();
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double gb = (double)S * P * H * 16 / 1e9;
    printf("{\"kind\": \"%s\", \"GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.3f}\n", name, gb, ms / 3, gb / (ms / 3));
}

int main() {
    const size_t n = (size_t)S * P * H;
    f4* y;
    if (hipMalloc(&y, n * 16) != hipSuccess) return 1;
    (void)hipMemset(y, 0, n * 16);
    const unsigned total = (unsigned)((size_t)S * H / 64);          // waves
    timeit("pat0_128B_pieces_wg4", [&] { hipLaunchKernelGGL(wr<0>, dim3(total / 4), dim3(256), 0, 0, y); });
    timeit("pat1_128B_pieces_wg16", [&] { hipLaunchKernelGGL(wr<1>, dim3(total / 16), dim3(1024), 0, 0, y); });
    timeit("pat2_1KiB_runs_4searches", [&] { hipLaunchKernelGGL(wr<2>, dim3(total / 4), dim3(256), 0, 0, y); });
    timeit("pat3_4KiB_runs", [&] { hipLaunchKernelGGL(wr<3>, dim3(total / 4), dim3(256), 0, 0, y); });
    timeit("pat4_item_major_layout", [&] { hipLaunchKernelGGL(wr<4>, dim3(total / 4), dim3(256), 0, 0, y); });
    return 0;
}
