// Dev microbenchmark (gfx950): do VALU work and LDS traffic on one CU overlap, or do they add up?
// Every operation is inline asm (the compiler sinks plain LDS stores out of such a loop).  Per iteration and wave:
//   V  = 64 v_fma_f32 (independent chains of 16)
//   L  = 16 dwords per lane written to and read back from a wave-private, conflict-free LDS region, as
//        b32 (16 + 16 instructions), b64 (8 + 8) or b128 (4 + 4)
//   mode 0: V only      mode 1: L only      mode 2: V and L in one instruction stream
//   mode 3: even waves V only, odd waves L only (per CU: half of each)
// hipcc --offload-arch=gfx950 -O3 valu_lds_overlap.hip -o valu_lds_overlap && ./valu_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int WG = 1024;
constexpr int ITER = 400;

template <int W>
__device__ __forceinline__ void lds_round(float (&a)[16], unsigned addr_w, unsigned addr_r) {
    if (W == 1) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr_w), "v"(a[u]), "n"(256 * u) : "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr_r), "n"(256 * u) : "memory");
    } else if (W == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f2 v = {a[2 * u], a[2 * u + 1]};
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr_w), "v"(v), "n"(512 * u) : "memory");
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f2 v;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr_r), "n"(512 * u) : "memory");
            a[2 * u] = v.x; a[2 * u + 1] = v.y;
        }
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f4 v = {a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]};
            asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr_w), "v"(v), "n"(1024 * u) : "memory");
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr_r), "n"(1024 * u) : "memory");
            a[4 * u] = v.x; a[4 * u + 1] = v.y; a[4 * u + 2] = v.z; a[4 * u + 3] = v.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int MODE, int W>
__global__ __launch_bounds__(WG, 8) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[18432];      // 72 KB: two workgroups per CU, like ifft_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = tid * 1e-3f + u;
    const float c = 1.0001f, d = 0.5f;
    // wave-private 1024 floats (+ padding); lane-consecutive elements of W dwords: no bank conflicts
    const unsigned base = (unsigned)(size_t)(lds) + 4u * (wave * 1152);
    const unsigned addr_w = base + 4u * W * lane, addr_r = base + 4u * W * (lane ^ 1);
    const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_lds = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(c), "v"(d));
            }
        }
        if (do_lds) lds_round<W>(a, addr_w, addr_r);
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += a[u];
    out[blockIdx.x * WG + tid] = s;
}

template <int MODE, int W>
float run(const char* name) {
    float* out;
    const int blocks = 512;                            // two workgroups per CU, all resident
    (void)hipMalloc(&out, (size_t)blocks * WG * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(WG), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(WG), 0, 0, out, ITER);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %-52s %.3f ms\n", name, ms);
    (void)hipFree(out);
    return ms;
}

template <int W>
void suite(const char* tag) {
    printf("%s\n", tag);
    const float v = run<0, W>("V only (64 fma per iteration per wave)");
    const float l = run<1, W>("L only (16 dwords per lane out and back)");
    const float b = run<2, W>("V and L, one stream per wave");
    const float h = run<3, W>("even waves V, odd waves L (half of each per CU)");
    printf("  V + L = %.3f, max = %.3f, measured both = %.3f; split waves %.3f (overlap: %.3f, no overlap: %.3f)\n",
           v + l, v > l ? v : l, b, h, (v > l ? v : l) / 2, (v + l) / 2);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        suite<1>("ds_write_b32 / ds_read_b32");
        suite<2>("ds_write_b64 / ds_read_b64");
        suite<4>("ds_write_b128 / ds_read_b128");
    }
    return 0;
}
