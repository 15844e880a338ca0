// Dev microbenchmark: does the ORDER in which mac_kernel's workgroups write Y matter to HBM?
//   pattern 0: Y[search][pair][chunk][256 float4]  (today: a workgroup = (search, chunk) writes 4 KiB pieces 64 KiB apart)
//   pattern 1: Y[chunk][search][pair][256 float4]  (a workgroup writes one contiguous 704 KiB run)
// and the matching read patterns of ifft_kernel (a workgroup = one pair reads 16 chunks):
//   pattern 2: reads of layout 0 (64 KiB contiguous)      pattern 3: reads of layout 1 (16 pieces of 4 KiB)
// hipcc --offload-arch=gfx950 -O3 hbm_pattern.hip -o hbm_pattern && ./hbm_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int S = 1000, P = 176, C = 16;       // searches, pairs per search, chunks per pair

template <int PAT>
__global__ __launch_bounds__(256) void wr(f4* __restrict__ y) {
    const int s = blockIdx.x / C, c = blockIdx.x % C;
    for (int i = 0; i < P; ++i) {
        const size_t e = PAT == 0 ? (((size_t)s * P + i) * C + c) * 256 + threadIdx.x
                                  : (((size_t)c * S + s) * P + i) * 256 + threadIdx.x;
        __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, y + e);
    }
}

template <int PAT>
__global__ __launch_bounds__(512) void rd(const f4* __restrict__ y, float* sink) {
    const int s = blockIdx.x / P, i = blockIdx.x % P;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = t * 512 + threadIdx.x;            // float4 index inside the pair's 4096
        const int c = q / 256, l = q % 256;
        const size_t e = PAT == 0 ? (((size_t)s * P + i) * C + c) * 256 + l : (((size_t)c * S + s) * P + i) * 256 + l;
        acc += y[e];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <class F>
static void timeit(const char* name, F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); launch();
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double gb = (double)S * P * C * 256 * 16 / 1e9;
    printf("{\"kind\": \"%s\", \"GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.3f}\n", name, gb, ms / 5, gb / (ms / 5) );
}

int main() {
    const size_t n = (size_t)S * P * C * 256;
    f4* y; float* sink;
    if (hipMalloc(&y, n * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(y, 0, n * 16);
    timeit("write_layout0_pieces", [&] { hipLaunchKernelGGL(wr<0>, dim3(S * C), dim3(256), 0, 0, y); });
    timeit("write_layout1_runs", [&] { hipLaunchKernelGGL(wr<1>, dim3(S * C), dim3(256), 0, 0, y); });
    timeit("read_layout0_64k", [&] { hipLaunchKernelGGL(rd<0>, dim3(S * P), dim3(512), 0, 0, y, sink); });
    timeit("read_layout1_pieces", [&] { hipLaunchKernelGGL(rd<1>, dim3(S * P), dim3(512), 0, 0, y, sink); });
    return 0;
}
