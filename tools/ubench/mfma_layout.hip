// tools/ubench/mfma_layout.hip -- which lane holds which element of v_mfma_f32_16x16x32_f16's operands (gfx950).
// Checks the maps tests/host_mfma_check.cpp assumes:  A[m][k]: lane m + 16 (k / 8), element k % 8;  B[k][n]: lane n + 16 (k / 8),
// element k % 8;  D[m][n]: lane n + 16 (m / 4), register m % 4 -- against a CPU matrix product of asymmetric integer matrices.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_layout.hip -o tools/ubench/mfma_layout && tools/ubench/mfma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k(const _Float16* a_mat, const _Float16* b_mat, float* d_mat) {     // a: 16 x 32, b: 32 x 16, d: 16 x 16, row major
    const int l = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = a_mat[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = b_mat[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d_mat[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}

int main() {
    _Float16 ha[16 * 32], hb[32 * 16];
    float ref[16 * 16], out[16 * 16];
    for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 32; ++kk) ha[m * 32 + kk] = (_Float16)(float)((m * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < 32; ++kk) for (int n = 0; n < 16; ++n) hb[kk * 16 + n] = (_Float16)(float)((kk * 5 + n * 13 + kk * n) % 7 - 3);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float s = 0; for (int kk = 0; kk < 32; ++kk) s += (float)ha[m * 32 + kk] * (float)hb[kk * 16 + n];
        ref[m * 16 + n] = s;
    }
    _Float16 *da, *db; float* dd;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(out));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(out, dd, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += out[i] != ref[i];
    printf("mfma_f32_16x16x32_f16 operand maps: %s (%d of 256 results differ)\n", bad ? "NOT as assumed" : "as assumed", bad);
    return bad != 0;
}
