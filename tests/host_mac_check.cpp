// tests/host_mac_check.cpp -- runs sushi_amd/csrc/mac_core.hpp (the ring-buffered frequency-domain
// multiply-accumulate) on the CPU against the plain double-sum definition.  Built and run by
// tests/test_fft_core_host.py; prints the largest relative error over a set of (n_seg, npairs) shapes.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sushi_amd/csrc/mac_core.hpp"

using namespace sushi_mac;
typedef std::complex<double> cd;

static float rnd() { return (float)rand() / RAND_MAX - 0.5f; }

template <int SMAX, int STEP>
static double run(int n_seg, int npairs, int nz_avail) {
    std::vector<c2> T(n_seg), Z(STEP * npairs + n_seg + 80), Y(npairs, c2{9e9f, 9e9f, 9e9f, 9e9f});
    for (auto& t : T) t = c2{rnd(), rnd(), rnd(), rnd()};
    for (auto& z : Z) z = c2{rnd(), rnd(), rnd(), rnd()};
    auto lz = [&](int j) { return j < nz_avail ? Z[j] : zero2(); };
    mac_stream<SMAX, STEP>(n_seg, npairs, [&](int s) { return T[s]; }, lz, [&](int i) { return Y[i]; },
                     [&](int i, c2 v) { Y[i] = v; });
    double worst = 0;
    for (int i = 0; i < npairs; ++i) {
        cd a(0, 0), b(0, 0);
        double mag = 1e-30;
        for (int s = 0; s < n_seg; ++s) {
            const c2 z = lz(STEP * i + s);
            a += cd(T[s].ax, T[s].ay) * cd(z.ax, z.ay);
            b += cd(T[s].bx, T[s].by) * cd(z.bx, z.by);
            mag += std::abs(cd(T[s].ax, T[s].ay) * cd(z.ax, z.ay));
        }
        worst = std::fmax(worst, std::abs(a - cd(Y[i].ax, Y[i].ay)) / mag);
        worst = std::fmax(worst, std::abs(b - cd(Y[i].bx, Y[i].by)) / mag);
    }
    return worst;
}

int main() {
    srand(7);
    double worst = 0;
    const int segs[] = {1, 2, 3, 7, 8, 9, 15, 16, 17, 30, 33};
    const int pairs[] = {1, 2, 3, 5, 8, 9, 44, 177};
    for (int s : segs)
        for (int p : pairs) {
            worst = std::fmax(worst, run<8, 2>(s, p, 1 << 30));
            worst = std::fmax(worst, run<16, 2>(s, p, 1 << 30));
            worst = std::fmax(worst, run<16, 2>(s, p, 2 * p - 1));      // stream ends inside the window
            worst = std::fmax(worst, run<2, 2>(s, p, 1 << 30));
            worst = std::fmax(worst, run<12, 2>(s, p, 2 * p + 3));
            worst = std::fmax(worst, run<6, 6>(s, p, 1 << 30));         // pairs 6 units apart (75 % valid layout)
            worst = std::fmax(worst, run<12, 6>(s, p, 1 << 30));
            worst = std::fmax(worst, run<18, 6>(s, p, 6 * p - 2));
            worst = std::fmax(worst, run<24, 6>(s, p, 1 << 30));
            worst = std::fmax(worst, run<30, 6>(s, p, 6 * p + 7));
        }
    printf("%.3e\n", worst);
    return worst < 1e-5 ? 0 : 1;
}
