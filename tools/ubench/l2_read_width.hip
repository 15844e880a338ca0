// Dev microbenchmark (gfx950): what does a CU get out of its XCD's L2 per clock, as a function of the WIDTH of the loads?
// ifft_kernel's epilogue reads 192 KB of window-energy prefix values per pair as single-dword loads (a wave instruction =
// 256 contiguous bytes), all L2 hits, none L1 hits -- and sits at ~13 B per clock and CU through its L1s.  Is that a limit
// of bytes, or of requests?
//   Every workgroup (1024 threads, two per CU: 72 KB of LDS like ifft_kernel) streams over its XCD's own region (L2-resident:
//   REGION bytes per XCD, >> the 32 KB L1) again and again; per iteration a thread reads 64 bytes as
//     W = 1: 16 dword loads, lane-contiguous (256 B per wave instruction), 4 KB apart  (ifft_kernel's pattern)
//     W = 2:  8 dwordx2 loads (512 B per wave instruction)
//     W = 4:  4 dwordx4 loads (1 KB per wave instruction)
//   and adds them up.  Output: GB/s over the chip and bytes per ns and CU.
// hipcc --offload-arch=gfx950 -O3 l2_read_width.hip -o l2_read_width && ./l2_read_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int WG = 1024;

template <int W>
__global__ __launch_bounds__(WG, 8) void k(const float* __restrict__ buf, size_t region_floats, int iters, float* out) {
    __shared__ float lds[18432];
    const int tid = threadIdx.x;
    if (tid == 0) lds[0] = 0.f;
    const float* __restrict__ base = buf + (size_t)(blockIdx.x & 7) * region_floats;
    // a workgroup reads 64 KB per iteration; workgroups of one XCD start at different places and walk the region
    const size_t chunk = (size_t)WG * 16;                                  // floats per iteration and workgroup
    const size_t n_chunks = region_floats / chunk;
    size_t c = ((size_t)(blockIdx.x >> 3) * 7) % n_chunks;
    float acc = 0.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int it = 0; it < iters; ++it) {
        const float* p = base + c * chunk;
        if (W == 1) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = p[tid + WG * r];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += v[r];
        } else if (W == 2) {
            f2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = reinterpret_cast<const f2*>(p)[tid + WG * r];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += v[r].x + v[r].y;
        } else {
            f4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = reinterpret_cast<const f4*>(p)[tid + WG * r];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += (v[r].x + v[r].y) + (v[r].z + v[r].w);
        }
        c += 13; if (c >= n_chunks) c -= n_chunks;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc + lds[0];
}

template <int W>
void run(const float* buf, size_t region_floats, float* out, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(WG), 0, 0, buf, region_floats, iters, out);      // warm: the region enters the L2s
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(WG), 0, 0, buf, region_floats, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * iters * WG * 64.0;
    printf("{\"width_dwords\": %d, \"region_MB_per_xcd\": %.2f, \"blocks\": %d, \"ms\": %.3f, \"GB_per_s\": %.0f, \"B_per_ns_per_CU\": %.2f}\n",
           W, region_floats * 4.0 / 1048576.0, blocks, ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 256.0);
}

int main() {
    float* out; hipMalloc(&out, 1 << 20);
    for (size_t region_mb : {1, 2, 3, 16}) {                  // 16 MB per XCD: no longer L2-resident (256 MB memory-side cache / HBM)
        const size_t region_floats = region_mb * 262144;
        float* buf; hipMalloc(&buf, region_floats * 4 * 8);
        hipMemset(buf, 0, region_floats * 4 * 8);
        const int blocks = 512, iters = 2000;
        run<1>(buf, region_floats, out, blocks, iters);
        run<2>(buf, region_floats, out, blocks, iters);
        run<4>(buf, region_floats, out, blocks, iters);
        hipFree(buf);
    }
    return 0;
}
