// tests/host_fft_check.cpp -- runs sushi_amd/csrc/fft_core.hpp on the CPU, one emulated thread at a
// time (barriers become loop boundaries), and compares with a float64 reference DFT.
// Built and run by tests/test_fft_core_host.py; prints "max_rel_err_fwd max_rel_err_inv".
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sushi_amd/csrc/fft_core.hpp"

using namespace sushi_fft;
typedef std::complex<double> cd;

static void ref_fft(std::vector<cd>& a, int dir) {   // iterative radix-2, float64
    const int n = (int)a.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = dir * 2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const cd w(std::cos(ang * k), std::sin(ang * k));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

template <int DIR>
static double run(const std::vector<cpx>& tw, unsigned seed) {
    std::vector<cpx> x(N);
    srand(seed);
    for (int n = 0; n < N; ++n) {
        x[n].x = (float)rand() / RAND_MAX - 0.5f;
        x[n].y = (float)rand() / RAND_MAX - 0.5f;
    }
    std::vector<cd> r(N);
    for (int n = 0; n < N; ++n) r[n] = cd(x[n].x, x[n].y);
    ref_fft(r, DIR);
    std::vector<cpx> regs((size_t)NT * PER), lds(LDS_ELEMS);
    for (int tid = 0; tid < NT; ++tid)
        for (int q = 0; q < PER; ++q) regs[(size_t)tid * PER + q] = x[in_index(tid, q)];
#define ALL(stmt) for (int tid = 0; tid < NT; ++tid) { cpx* v = &regs[(size_t)tid * PER]; \
        const Twiddles t = load_twiddles<DIR>(tid, tw.data()); (void)t; stmt; }
    ALL((pass_compute<8, 1, DIR>(v, cpx{1.f, 0.f}), pass_store<8, 1, true>(v, tid, lds.data())))
    ALL((pass_load<8>(v, tid, lds.data()), pass_compute<8, 8, DIR>(v, t.p2)))
    ALL((pass_store<8, 8, false>(v, tid, lds.data())))
    ALL((pass_load<8>(v, tid, lds.data()), pass_compute<8, 64, DIR>(v, t.p3)))
    ALL((pass_store<8, 64, false>(v, tid, lds.data())))
    ALL((pass_load<16>(v, tid, lds.data()), pass_compute<16, 512, DIR>(v, t.p4)))
    double err2 = 0, ref2 = 0;
    for (int tid = 0; tid < NT; ++tid)
        for (int q = 0; q < PER; ++q) {
            const cpx g = regs[(size_t)tid * PER + q];
            const cd e = r[out_index(tid, q)];
            err2 += std::norm(cd(g.x, g.y) - e);
            ref2 += std::norm(e);
        }
    return std::sqrt(err2 / ref2);
}

int main() {
    std::vector<cpx> tw(N);
    for (int n = 0; n < N; ++n) {
        tw[n].x = (float)std::cos(2.0 * M_PI * n / N);
        tw[n].y = (float)-std::sin(2.0 * M_PI * n / N);
    }
    const double f = run<-1>(tw, 1), b = run<1>(tw, 2);
    printf("%.3e %.3e\n", f, b);
    return (f < 1e-6 && b < 1e-6) ? 0 : 1;
}
