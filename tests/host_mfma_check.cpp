// tests/host_mfma_check.cpp -- the MFMA first pass of the inverse wave plan (fft_core.hpp "First pass ... on the matrix pipe"),
// emulated lane by lane on the CPU: stored order -> 16-byte loads -> v_perm_b32 / v_permlane32_swap -> v_mfma_f32_16x16x32_f16
// (A = data, B = DFT matrix as high + low halves) -> passes 2 .. 4 of fft_core.hpp as they are.  Prints the relative L2 error against a
// float64 FFT of the same half-valued input.  The lane maps of the MFMA are the emulator's ASSUMPTION (A[m][k]: lane m + 16 (k / 8),
// element k % 8; B[k][n]: lane n + 16 (k / 8), element k % 8; D[m][n]: lane n + 16 (m / 4), register m % 4): a 10-second GPU test
// (tools/ubench/mfma_layout.hip) checks exactly that.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../sushi_amd/csrc/fft_core.hpp"
#include "../sushi_amd/csrc/mac_core.hpp"

using namespace sushi_fft;
typedef std::complex<double> cd;

static unsigned to_half(float f) {
    unsigned b; memcpy(&b, &f, 4);
    const unsigned s = (b >> 16) & 0x8000u;
    const float a = std::fabs(f);
    if (a == 0.f) return s;
    int e; const float m = std::frexp(a, &e);
    int he = e + 14;
    if (he <= 0) return s | (unsigned)std::nearbyint(std::ldexp(a, 24));
    unsigned q = (unsigned)std::nearbyint(std::ldexp(m, 11));
    if (q == 2048) { q = 1024; ++he; }
    return s | ((unsigned)he << 10) | (q - 1024);
}
static float from_half(unsigned h) { return sushi_mac::half_bits_to_float(h & 0xffffu); }

static void ref_fft(std::vector<cd>& a, int dir) {
    const int n = (int)a.size();
    for (int i = 1, j = 0; i < n; ++i) { int bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(a[i], a[j]); }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = dir * 2 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const cd w(std::cos(ang * k), std::sin(ang * k));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v; a[i + k + len / 2] = u - v;
            }
    }
}

struct u4 { unsigned w[4]; };
struct h8v { unsigned short h[8]; };

// v_perm_b32 dst, src0, src1, sel: byte i of dst = byte sel[i] of {src0 : src1} (0 .. 3 from src1, 4 .. 7 from src0)
static unsigned perm_b32(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long both = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((both >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}

int main() {
    constexpr int N = WN, NTS = WNT, DIR = 1;
    std::vector<cpx> tw(TWIDDLE_N);
    for (int n = 0; n < TWIDDLE_N; ++n) { tw[n].x = (float)std::cos(2.0 * M_PI * n / TWIDDLE_N); tw[n].y = (float)-std::sin(2.0 * M_PI * n / TWIDDLE_N); }
    srand(11);
    // a spectrum of half values, stored in the MFMA load order as packed words
    std::vector<unsigned> stored(N);
    std::vector<cd> r(N);
    for (int f = 0; f < N; ++f) {
        const unsigned re = to_half(((float)rand() / RAND_MAX - 0.5f) * 300.f), im = to_half(((float)rand() / RAND_MAX - 0.5f) * 300.f);
        stored[mslot_of_bin(f)] = re | (im << 16);
        r[f] = cd(from_half(re), from_half(im));
    }
    {   // the two maps agree, and mslot_of_bin is a permutation
        std::vector<int> seen(N, 0);
        for (int f = 0; f < N; ++f) seen[mslot_of_bin(f)]++;
        for (int e = 0; e < N; ++e) if (seen[e] != 1) { printf("mslot_of_bin is not a permutation\n"); return 1; }
        for (int tid = 0; tid < NTS; ++tid)
            for (int g = 0; g < 4; ++g)
                for (int t = 0; t < 4; ++t)
                    if (mslot_of_bin(mbin(tid, g, t)) != 4 * wslot_uint4(tid, g) + t) { printf("mbin / mslot_of_bin disagree\n"); return 1; }
    }
    ref_fft(r, DIR);
    // DFT-matrix operands: [form][hi / lo][lane][j]
    std::vector<h8v> bop(2 * 2 * 64);
    for (int form = 0; form < 2; ++form)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const double v = dft16_operand(form, l, j, DIR);
                const unsigned hi = to_half((float)v);
                const unsigned lo = to_half((float)(v - (double)from_half(hi)));
                bop[(form * 2 + 0) * 64 + l].h[j] = (unsigned short)hi;
                bop[(form * 2 + 1) * 64 + l].h[j] = (unsigned short)lo;
            }
    std::vector<cpx> regs((size_t)NTS * PER);
    for (int w = 0; w < NTS / 64; ++w) {
        for (int g = 0; g < 4; ++g) {
            // the 16-byte loads of this group, then the shuffle
            u4 ld[64];
            for (int l = 0; l < 64; ++l)
                for (int t = 0; t < 4; ++t) ld[l].w[t] = stored[4 * wslot_uint4(64 * w + l, g) + t];
            unsigned rr01[64], rr23[64], ii01[64], ii23[64];
            for (int l = 0; l < 64; ++l) {
                rr01[l] = perm_b32(ld[l].w[1], ld[l].w[0], 0x05040100u); rr23[l] = perm_b32(ld[l].w[3], ld[l].w[2], 0x05040100u);
                ii01[l] = perm_b32(ld[l].w[1], ld[l].w[0], 0x07060302u); ii23[l] = perm_b32(ld[l].w[3], ld[l].w[2], 0x07060302u);
            }
            // v_permlane32_swap vdst, src0: lanes [32, 64) of vdst <-> lanes [0, 32) of src0:  swap(vdst = rr, src0 = ii)
            for (int l = 0; l < 32; ++l) { std::swap(rr01[32 + l], ii01[l]); std::swap(rr23[32 + l], ii23[l]); }
            h8v a[64];
            for (int l = 0; l < 64; ++l) {
                const unsigned wds[4] = {rr01[l], rr23[l], ii01[l], ii23[l]};
                for (int j = 0; j < 8; ++j) a[l].h[j] = (unsigned short)((wds[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            }
            // D = A B (high part, then low part accumulated), for the real and the imaginary parts of the result
            for (int form = 0; form < 2; ++form)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int m = 4 * (l >> 4) + i, n = l & 15;
                        float acc = 0.f;
                        for (int part = 0; part < 2; ++part) {                 // high, low
                            double s = 0;
                            for (int k = 0; k < 32; ++k)
                                s += (double)from_half(a[m + 16 * (k >> 3)].h[k & 7]) * (double)from_half(bop[(form * 2 + part) * 64 + n + 16 * (k >> 3)].h[k & 7]);
                            acc = (float)((double)acc + s);
                        }
                        cpx& out = regs[(size_t)(64 * w + l) * PER + 4 * g + i];
                        if (form == 0) out.x = acc; else out.y = acc;
                    }
        }
    }
    // the wave plan from pass 2 on (fft_core.hpp), as tests/host_fft_check.cpp runs it
    std::vector<float> fl(W_LDS_FLOATS, 1e30f);
#define ALL(stmt) for (int tid = 0; tid < NTS; ++tid) { cpx* v = &regs[(size_t)tid * PER]; \
        const WTwiddles t = load_wtwiddles<DIR>(tid, tw.data()); (void)t; stmt; }
    ALL((pass_compute<16, 16, DIR>(v, t.g2)))
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
    printf("%.3e\n", std::sqrt(err2 / ref2));
    return std::sqrt(err2 / ref2) < 1e-6 ? 0 : 1;
}
