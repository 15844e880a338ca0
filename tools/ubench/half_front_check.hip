// Dev check (gfx950): the packed-half in-wave passes (fft_core.hpp fft_wave_half_front, bound_kernel's transform) against the
// float32 ones (fft_wave_mfma_front) on random packed-half input: the same A_n1[k2] up to the halves' rounding and the 2^-10.
// hipcc --offload-arch=gfx950 -O3 -I sushi_amd/csrc tools/ubench/half_front_check.hip -o tools/ubench/half_front_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "fft_core.hpp"
using namespace sushi_fft;
__device__ const float g_twiddle[2 * TWIDDLE_N] = {
#include "_gen_twiddle16384.inc"
};
__device__ __attribute__((aligned(16))) const unsigned g_b[4 * 64 * 4] = {
#include "_gen_dft16_f16.inc"
};
__device__ __attribute__((aligned(16))) const unsigned g_bh[2 * 64 * 4] = {
#include "_gen_dft16_f16_bound.inc"
};
__global__ void k(const uint4v* y, float* out32, float* outh, unsigned* in2) {
    const int lane = threadIdx.x;
    const cpx* tw = reinterpret_cast<const cpx*>(g_twiddle);
    uint4v yl[4];
    for (int u = 0; u < 4; ++u) yl[u] = y[u * 64 + lane];
    const MfmaB mb = load_mfma_b(lane, reinterpret_cast<const uint4v*>(g_b));
    const WTwiddles wt = load_wtwiddles<1>(lane, tw);
    cpx v[16];
    fft_wave_mfma_front<1>(yl, v, lane, wt, mb);
    const MfmaBh mh = load_mfma_bh(lane, reinterpret_cast<const uint4v*>(g_bh));
    const HTwiddles ht = load_htwiddles(lane, tw);
    h2 vh[16];
    unsigned mi;
    fft_wave_half_front(yl, vh, ht, mh, mi);
    for (int r = 0; r < 16; ++r) {
        out32[(lane * 16 + r) * 2] = v[r].x; out32[(lane * 16 + r) * 2 + 1] = v[r].y;
        outh[(lane * 16 + r) * 2] = (float)vh[r].x * 1024.f; outh[(lane * 16 + r) * 2 + 1] = (float)vh[r].y * 1024.f;
    }
    in2[lane] = mi;
    unsigned m2 = 0u;
    for (int r = 0; r < 16; ++r) m2 = h_max_bits(m2, h_abs2(vh[r]));
    float m2ref = 0.f;
    for (int r = 0; r < 16; ++r) m2ref = fmaxf(m2ref, (float)vh[r].x * (float)vh[r].x + (float)vh[r].y * (float)vh[r].y);
    if (lane < 3) printf("lane %d m2 bits %08x = %g, reference %g\n", lane, m2, __uint_as_float(m2), m2ref);
}
// The low-band group transform (fft_wave_half_front_low, bound_low_kernel's) against a float64 DFT of the same 512 bins on the
// host: entry (g4 * 32 + l) of the group's 2 KB holds, at sub-position j, V_g[m] with m = 64 d1 + 4 d2 + d3 (fft_core.hpp lb_d1 /
// lslot_of_thread), and A_g[k2] = sum_m V_g[m] exp(+2 pi i m k2 / 1024) ends in register 4 e3 + e2lo of lane (e2hi, e1).
__device__ __attribute__((aligned(16))) const unsigned g_bl[2 * 64 * 2] = {
#include "_gen_dft16_f16_bound_low.inc"
};
__global__ void klow(const uint4v* y, float* outh, unsigned* in2) {
    const int lane = threadIdx.x;
    const cpx* tw = reinterpret_cast<const cpx*>(g_twiddle);
    uint4v yl[4];
    for (int u = 0; u < 4; ++u) yl[u] = y[u * 32 + (lane & 31)];
    const MfmaBl mb = load_mfma_bl(lane, reinterpret_cast<const uint2v*>(g_bl));
    const HTwiddles ht = load_htwiddles(lane, tw);
    h2 vh[16];
    unsigned mi;
    fft_wave_half_front_low(yl, vh, ht, mb, mi);
    for (int r = 0; r < 16; ++r) {
        outh[(lane * 16 + r) * 2] = (float)vh[r].x * 1024.f; outh[(lane * 16 + r) * 2 + 1] = (float)vh[r].y * 1024.f;
    }
    in2[lane] = mi;
}
static int check_low() {
    std::vector<unsigned> y(4 * 32 * 4);
    srand(7);
    std::vector<double> re(y.size()), im(y.size());
    for (size_t i = 0; i < y.size(); ++i) {
        _Float16 a = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2000.f), b = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2000.f);
        unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
        y[i] = ua | ((unsigned)ub << 16);
        re[i] = (double)(float)a; im[i] = (double)(float)b;
    }
    // V_g[m] from the layout
    std::vector<double> vr(1024, 0.0), vi(1024, 0.0);
    for (int g4 = 0; g4 < 4; ++g4)
        for (int l = 0; l < 32; ++l)
            for (int j = 0; j < 4; ++j) {
                const int kq = l >> 4, mm = l & 15, d2 = 4 * g4 + (mm & 3), d3 = mm >> 2;
                const int m = 64 * lb_d1(kq, j) + 4 * d2 + d3;
                const size_t w = (size_t)(g4 * 32 + l) * 4 + j;
                vr[m] = re[w]; vi[m] = im[w];
            }
    uint4v* dy; float* dh; unsigned* di;
    hipMalloc(&dy, y.size() * 4); hipMalloc(&dh, 64 * 16 * 2 * 4); hipMalloc(&di, 64 * 4);
    hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(klow, dim3(1), dim3(64), 0, 0, dy, dh, di);
    std::vector<float> b(64 * 16 * 2); std::vector<unsigned> in2(64);
    hipMemcpy(b.data(), dh, b.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(in2.data(), di, 64 * 4, hipMemcpyDeviceToHost);
    double maxa = 0, maxd = 0; int nbad = 0;
    const double PI = 3.14159265358979323846;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int e1 = lane & 15, e2hi = lane >> 4, e2lo = r & 3, e3 = r >> 2;
            const int k2 = e1 + 64 * e2hi + 16 * e2lo + 256 * e3;
            double ar = 0, ai = 0;
            for (int m = 0; m < 1024; ++m) {
                if (vr[m] == 0.0 && vi[m] == 0.0) continue;
                const double ang = 2.0 * PI * (double)((m * k2) & 1023) / 1024.0, c = cos(ang), s = sin(ang);
                ar += vr[m] * c - vi[m] * s; ai += vr[m] * s + vi[m] * c;
            }
            const double gr = b[(lane * 16 + r) * 2], gi = b[(lane * 16 + r) * 2 + 1];
            maxa = fmax(maxa, hypot(ar, ai)); maxd = fmax(maxd, hypot(ar - gr, ai - gi));
            if (!(gr == gr) || !(gi == gi) || std::isinf(gr) || std::isinf(gi)) ++nbad;
        }
    float mi = 0; for (unsigned u : in2) { float f; memcpy(&f, &u, 4); mi = fmaxf(mi, f); }
    printf("{\"low_max_abs_A_f64\": %g, \"low_max_abs_diff\": %g, \"low_diff_over_max\": %g, \"low_non_finite\": %d, \"low_bound_on_diff\": %g}\n",
           maxa, maxd, maxd / maxa, nbad, 0.29 * sqrt(mi) * 1024.0);
    return maxd <= 0.29 * sqrt(mi) * 1024.0 && nbad == 0 ? 0 : 1;
}
int main() {
    const int rc_low = check_low();
    std::vector<unsigned> y(4 * 64 * 4);
    srand(1);
    for (auto& w : y) {
        _Float16 a = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2000.f), b = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2000.f);
        unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
        w = ua | ((unsigned)ub << 16);
    }
    uint4v* dy; float *d32, *dh; unsigned* di;
    hipMalloc(&dy, y.size() * 4); hipMalloc(&d32, 64 * 16 * 2 * 4); hipMalloc(&dh, 64 * 16 * 2 * 4); hipMalloc(&di, 64 * 4);
    hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dy, d32, dh, di);
    std::vector<float> a(64 * 16 * 2), b(64 * 16 * 2); std::vector<unsigned> in2(64);
    hipMemcpy(a.data(), d32, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dh, b.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(in2.data(), di, 64 * 4, hipMemcpyDeviceToHost);
    double maxa = 0, maxd = 0, maxb = 0; int nbad = 0;
    for (size_t i = 0; i < a.size(); i += 2) {
        const double ma = hypot(a[i], a[i + 1]), mb = hypot(b[i], b[i + 1]), d = hypot(a[i] - b[i], a[i + 1] - b[i + 1]);
        maxa = fmax(maxa, ma); maxb = fmax(maxb, mb); maxd = fmax(maxd, d);
        if (!(mb == mb) || std::isinf(mb)) ++nbad;
    }
    float mi = 0; for (unsigned u : in2) { float f; memcpy(&f, &u, 4); mi = fmaxf(mi, f); }
    printf("{\"max_abs_A_f32\": %g, \"max_abs_A_half\": %g, \"max_abs_diff\": %g, \"diff_over_max\": %g, \"non_finite\": %d, \"max_pass1_x1024\": %g, \"bound_on_diff\": %g}\n",
           maxa, maxb, maxd, maxd / maxa, nbad, sqrt(mi) * 1024.0, 0.29 * sqrt(mi) * 1024.0);
    return (maxd <= 0.29 * sqrt(mi) * 1024.0 && nbad == 0 ? 0 : 1) | rc_low;
}
