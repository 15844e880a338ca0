// Dev microbenchmark: issue rate of plain vs packed f32 VALU ops on gfx950 (wave64).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    const unsigned long long c_start = __builtin_readcyclecounter();     // s_memtime: shader cycles, whatever the clock is
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0001f, d = 0.5f;
    const f2 pc = {c, c}, pd = {d, d};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
            }
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - c_start;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    float* out;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned long long* cyc;
    hipMalloc(&cyc, (size_t)blocks * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    const double per_simd = instr_per_wave * waves_per_simd;             // wave-instructions per SIMD
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean_cyc = 0;
    for (auto v : h) mean_cyc += (double)v;
    mean_cyc /= blocks;
    // s_memtime counts at a fixed 100 MHz on gfx9 (not shader cycles): the effective shader clock is what makes
    // ns per instruction x clock = the guide's 2 cycles
    printf("%-14s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz; 2 cycles <=> %.2f GHz); "
           "s_memtime ticks per kernel %.0f (= %.1f MHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, 2.0 / (ms * 1e6 / per_simd),
           mean_cyc, mean_cyc / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_add_f32", w);
        run<3>("v_pk_add_f32", w);
    }
    return 0;
}
