// Dev microbenchmark: what HBM bandwidth a plain streaming kernel reaches on this part (MI355X), by access
// kind: 16-byte reads, 16-byte stores (plain / non-temporal), copy.  The FFT path's Y round trip is
// 11.6 GB written by mac_kernel and read back by ifft_kernel; these are the ceilings to hold them against.
// hipcc --offload-arch=gfx950 -O3 hbm_bw.hip -o hbm_bw && ./hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, f4* __restrict__ out, size_t n, float* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) { const f4 v = in[i]; acc += v; }
        else if (MODE == 1) { out[i] = f4{1.f, 2.f, 3.f, (float)i}; }
        else if (MODE == 2) { __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, out + i); }
        else if (MODE == 3) { out[i] = in[i]; }
        else { __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i); }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE>
static void run(const char* name, const f4* in, f4* out, size_t n, float* sink, double bytes_per_elem, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n, sink);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("{\"kind\": \"%s\", \"grid\": %d, \"GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.3f}\n", name, grid,
           n * bytes_per_elem / 1e9, ms / reps, n * bytes_per_elem / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const size_t n = (size_t)11600 * 1000 * 1000 / 16;      // 11.6 GB per array
    f4 *in, *out; float* sink;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(in, 0, n * 16);
    for (int grid : {2048, 8192, 65536}) {
        run<0>("read16", in, out, n, sink, 16, grid);
        run<1>("write16", in, out, n, sink, 16, grid);
        run<2>("write16_nt", in, out, n, sink, 16, grid);
        run<3>("copy16", in, out, n, sink, 32, grid);
        run<4>("copy16_nt", in, out, n, sink, 32, grid);
    }
    return 0;
}
