// Operand / result layout of v_mfma_f32_16x16x32_f16 and of v_permlane16_swap_b32 on gfx950, checked against what
// foldingdiff_amd/csrc/seq_attn16.hip assumes:
//   A[i][k]: lane i + 16 (k / 8), half k % 8;  B[k][j]: lane j + 16 (k / 8), half k % 8;  D[i][j]: lane j + 16 (i / 4), register i % 4
//   build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma16_layout_probe.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(const float* A, const float* B, float* D, unsigned* sw) {
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)A[c * 32 + 8 * g + e];
    b[e] = (_Float16)B[(8 * g + e) * 16 + c];
  }
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
  for (int e = 0; e < 4; ++e) D[(4 * g + e) * 16 + c] = d[e];
  unsigned x = lane, y = 100 + lane;
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  sw[lane] = r[0];
  sw[64 + lane] = r[1];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 13) % 9 - 4);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD; unsigned* dS; unsigned hS[128];
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dS, sizeof hS);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += std::fabs(hD[i] - ref[i]) > 1e-3f;
  printf("mfma_f32_16x16x32_f16 layout as assumed: %s (%d of 256 elements differ)\n", bad ? "NO" : "yes", bad);
  // expected: rows 1, 3 of vdst <-> rows 0, 2 of src: r0 = [x row0, y row0, x row2, y row2], r1 = [x row1 ... ] ?
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    const int row = l >> 4, cc = l & 15;
    const unsigned e0 = (row & 1) ? 100 + (16 * (row - 1) + cc) : (unsigned)l;      // vdst: odd rows take src's even row below
    const unsigned e1 = (row & 1) ? 100 + l : (unsigned)(16 * (row + 1) + cc);      // src: even rows take vdst's odd row above
    ok &= hS[l] == e0 && hS[64 + l] == e1;
  }
  printf("v_permlane16_swap as assumed: %s\n", ok ? "yes" : "NO");
  if (!ok) { for (int l = 0; l < 64; l += 16) printf("  lane %2d: vdst %u src %u\n", l, hS[l], hS[64 + l]); }
  return (bad || !ok) ? 1 : 0;
}
