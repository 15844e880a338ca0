// tests/host_fft_check.cpp -- runs sushi_amd/csrc/fft_core.hpp on the CPU, one emulated thread at a
// time (barriers become loop boundaries), and compares with a float64 reference DFT.
// Built and run by tests/test_fft_core_host.py; prints the relative L2 errors: forward / inverse for Plan<13>, Plan<14> and the wave plan.
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

template <int LOGN, int DIR>
static double run(const std::vector<cpx>& tw, unsigned seed) {
    typedef Plan<LOGN> P;
    constexpr int N = P::N, NTS = P::NT;
    std::vector<cpx> x(N);
    srand(seed);
    for (int n = 0; n < N; ++n) {
        x[n].x = (float)rand() / RAND_MAX - 0.5f;
        x[n].y = (float)rand() / RAND_MAX - 0.5f;
    }
    std::vector<cd> r(N);
    for (int n = 0; n < N; ++n) r[n] = cd(x[n].x, x[n].y);
    ref_fft(r, DIR);
    std::vector<cpx> regs((size_t)NTS * PER);
    std::vector<float> fl(lds_floats<LOGN>(), 1e30f);
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < PER; ++q) regs[(size_t)tid * PER + q] = x[in_index<LOGN>(tid, q)];
#define ALL(stmt) for (int tid = 0; tid < NTS; ++tid) { cpx* v = &regs[(size_t)tid * PER]; \
        const Twiddles t = load_twiddles<LOGN, DIR>(tid, tw.data()); (void)t; stmt; }
#define XCHG(EX) \
        ALL((split_store<LOGN, EX, 0>(v, tid, fl.data()))) \
        ALL((split_load<LOGN, EX, 0>(v, tid, fl.data()))) \
        ALL((split_store<LOGN, EX, 1>(v, tid, fl.data()))) \
        ALL((split_load<LOGN, EX, 1>(v, tid, fl.data())))
    ALL((pass_compute<R1, 1, DIR>(v, cpx{1.f, 0.f})))
    XCHG(1)
    ALL((pass_compute<R2, R1, DIR>(v, t.p2)))
    XCHG(2)
    ALL((pass_compute<P::R3, R1 * R2, DIR>(v, t.p3)))
    XCHG(3)
    ALL((pass_compute<R4, P::NT, DIR>(v, t.p4)))
#undef XCHG
#undef ALL
    double err2 = 0, ref2 = 0;
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < PER; ++q) {
            const cpx g = regs[(size_t)tid * PER + q];
            const cd e = r[out_index<LOGN>(tid, q)];
            err2 += std::norm(cd(g.x, g.y) - e);
            ref2 += std::norm(e);
        }
    return std::sqrt(err2 / ref2);
}

// The wave plan (fft_core.hpp "Wave plan"): the same per-thread pieces, the two permlane swaps emulated on the host.
template <int DIR>
static double run_wave(const std::vector<cpx>& tw, unsigned seed) {
    constexpr int N = WN, NTS = WNT;
    std::vector<cpx> x(N);
    srand(seed);
    for (int n = 0; n < N; ++n) {
        x[n].x = (float)rand() / RAND_MAX - 0.5f;
        x[n].y = (float)rand() / RAND_MAX - 0.5f;
    }
    std::vector<cd> r(N);
    for (int n = 0; n < N; ++n) r[n] = cd(x[n].x, x[n].y);
    ref_fft(r, DIR);
    // what a kernel does: a stored spectrum in slot order, loaded as 4-bin entries = four registers
    std::vector<cpx> stored(N);
    for (int f = 0; f < N; ++f) stored[wslot_of_bin(f)] = x[f];
    std::vector<cpx> regs((size_t)NTS * PER);
    for (int tid = 0; tid < NTS; ++tid)
        for (int u = 0; u < 4; ++u)
            for (int k = 0; k < 4; ++k) regs[(size_t)tid * PER + 4 * u + k] = stored[4 * wslot_uint4(tid, u) + k];
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < PER; ++q) {
            const cpx a = regs[(size_t)tid * PER + q], b = x[wbin(tid, q)];
            if (a.x != b.x || a.y != b.y) return 1.0;                    // slot order and load order disagree
        }
    std::vector<float> fl(W_LDS_FLOATS, 1e30f);
#define ALL(stmt) for (int tid = 0; tid < NTS; ++tid) { cpx* v = &regs[(size_t)tid * PER]; \
        const WTwiddles t = load_wtwiddles<DIR>(tid, tw.data()); (void)t; stmt; }
    ALL((Dft<16, DIR>::run(v)))
    ALL((w_row_store<0>(v, tid, fl.data())))
    ALL((w_row_load<0>(v, tid, fl.data())))
    ALL((w_row_store<1>(v, tid, fl.data())))
    ALL((w_row_load<1>(v, tid, fl.data())))
    ALL((pass_compute<16, 16, DIR>(v, t.g2)))
    // v_permlane32_swap vdst, src0: lanes [32, 64) of vdst <-> lanes [0, 32) of src0; v_permlane16_swap: odd rows of vdst <->
    // even rows of src0
    for (int w = 0; w < NTS / 64; ++w) {
        for (int q = 0; q < 8; ++q)
            for (int l = 0; l < 32; ++l)
                std::swap(regs[(size_t)(64 * w + 32 + l) * PER + q], regs[(size_t)(64 * w + l) * PER + q + 8]);
        for (int q = 0; q < 16; ++q) {
            if (q & 4) continue;
            for (int row = 0; row < 4; row += 2)
                for (int l = 0; l < 16; ++l)
                    std::swap(regs[(size_t)(64 * w + 16 * (row + 1) + l) * PER + q], regs[(size_t)(64 * w + 16 * row + l) * PER + q + 4]);
        }
    }
    ALL((w_pass3<DIR>(v, t.q3)))
    ALL((w_wg_store<0>(v, tid, fl.data())))
    ALL((w_wg_load<0>(v, tid, fl.data())))
    ALL((w_wg_store<1>(v, tid, fl.data())))
    ALL((w_wg_load<1>(v, tid, fl.data())))
    ALL((pass_compute<16, WNT, DIR>(v, t.p4)))
#undef ALL
    double err2 = 0, ref2 = 0;
    for (int tid = 0; tid < NTS; ++tid)
        for (int q = 0; q < PER; ++q) {
            const cpx g = regs[(size_t)tid * PER + q];
            const cd e = r[tid + NTS * q];
            err2 += std::norm(cd(g.x, g.y) - e);
            ref2 += std::norm(e);
        }
    return std::sqrt(err2 / ref2);
}

int main() {
    std::vector<cpx> tw(TWIDDLE_N);
    for (int n = 0; n < TWIDDLE_N; ++n) {
        tw[n].x = (float)std::cos(2.0 * M_PI * n / TWIDDLE_N);
        tw[n].y = (float)-std::sin(2.0 * M_PI * n / TWIDDLE_N);
    }
    const double f13 = run<13, -1>(tw, 1), b13 = run<13, 1>(tw, 2);
    const double f14 = run<14, -1>(tw, 3), b14 = run<14, 1>(tw, 4);
    const double fw = run_wave<-1>(tw, 5), bw = run_wave<1>(tw, 6);
    printf("%.3e %.3e %.3e %.3e %.3e %.3e\n", f13, b13, f14, b14, fw, bw);
    return (f13 < 1e-6 && b13 < 1e-6 && f14 < 1e-6 && b14 < 1e-6 && fw < 1e-6 && bw < 1e-6) ? 0 : 1;
}
