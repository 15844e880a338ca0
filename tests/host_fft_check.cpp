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

template <int NTS, int DIR, bool SPLIT = false>
static double run(const std::vector<cpx>& tw, unsigned seed) {
    constexpr int P = N / NTS;
    std::vector<cpx> x(N);
    srand(seed);
    for (int n = 0; n < N; ++n) {
        x[n].x = (float)rand() / RAND_MAX - 0.5f;
        x[n].y = (float)rand() / RAND_MAX - 0.5f;
    }
    std::vector<cd> r(N);
    for (int n = 0; n < N; ++n) r[n] = cd(x[n].x, x[n].y);
    ref_fft(r, DIR);
    std::vector<cpx> regs((size_t)NTS * P), lds(LDS_ELEMS);
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < P; ++q) regs[(size_t)tid * P + q] = x[in_index_t<NTS>(tid, q)];
#define ALL(stmt) for (int tid = 0; tid < NTS; ++tid) { cpx* v = &regs[(size_t)tid * P]; \
        const Twiddles t = load_twiddles<NTS, DIR>(tid, tw.data()); (void)t; stmt; }
    if (NTS == 512 && SPLIT) {
        // fft8192_split: real parts, then imaginary parts, through a buffer of LDS_ELEMS floats
        std::vector<float> fl(SPLIT_LDS_FLOATS);
#define XCHG(EX, RN) \
        ALL((split_store<EX, 0>(v, tid, fl.data()))) \
        ALL((split_load<EX, RN, 0>(v, tid, fl.data()))) \
        ALL((split_store<EX, 1>(v, tid, fl.data()))) \
        ALL((split_load<EX, RN, 1>(v, tid, fl.data())))
        ALL((pass_compute<16, 8, 1, DIR>(v, cpx{1.f, 0.f})))
        XCHG(1, 8)
        ALL((pass_compute<16, 8, 8, DIR>(v, t.p2)))
        XCHG(2, 8)
        ALL((pass_compute<16, 8, 64, DIR>(v, t.p3)))
        XCHG(3, 16)
        ALL((pass_compute<16, 16, 512, DIR>(v, t.p4)))
#undef XCHG
    } else if (NTS == 512) {
        ALL((pass_compute<16, 8, 1, DIR>(v, cpx{1.f, 0.f}), pass_store<512, 8, 1, true>(v, tid, lds.data())))
        ALL((pass_load<512, 8>(v, tid, lds.data()), pass_compute<16, 8, 8, DIR>(v, t.p2)))
        ALL((pass_store<512, 8, 8, false>(v, tid, lds.data())))
        ALL((pass_load<512, 8>(v, tid, lds.data()), pass_compute<16, 8, 64, DIR>(v, t.p3)))
        ALL((pass_store<512, 8, 64, false>(v, tid, lds.data())))
        ALL((pass_load<512, 16>(v, tid, lds.data()), pass_compute<16, 16, 512, DIR>(v, t.p4)))
    } else {
        ALL((pass_compute<32, 16, 1, DIR>(v, cpx{1.f, 0.f}), pass_store<256, 16, 1, true>(v, tid, lds.data())))
        ALL((pass_load<256, 16>(v, tid, lds.data()), pass_compute<32, 16, 16, DIR>(v, t.p2)))
        ALL((pass_store<256, 16, 16, false>(v, tid, lds.data())))
        ALL((pass_load<256, 32>(v, tid, lds.data()), pass_compute<32, 32, 256, DIR>(v, t.p3)))
    }
    double err2 = 0, ref2 = 0;
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < P; ++q) {
            const cpx g = regs[(size_t)tid * P + q];
            const cd e = r[out_index_t<NTS>(tid, q)];
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
    const double f = run<512, -1>(tw, 1), b = run<512, 1>(tw, 2);
    const double f2 = run<256, -1>(tw, 3), b2 = run<256, 1>(tw, 4);
    const double f3 = run<512, -1, true>(tw, 5), b3 = run<512, 1, true>(tw, 6);
    printf("%.3e %.3e %.3e %.3e %.3e %.3e\n", f, b, f2, b2, f3, b3);
    return (f < 1e-6 && b < 1e-6 && f2 < 1e-6 && b2 < 1e-6 && f3 < 1e-6 && b3 < 1e-6) ? 0 : 1;
}
