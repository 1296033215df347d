// NeRF: internal coordinates (backbone dihedrals + bond angles [+ bond lengths]) -> Cartesian N, CA, C
// coordinates, the step that runs right after sampling for every structure
// (foldingdiff/nerf.py:27-204 NERFBuilder.cartesian_coords / place_dihedral, called from
// create_new_chain_nerf, foldingdiff/angles_and_coords.py:112-184).  SURVEY 8(f) "next" row N1.
//
// The recurrence is sequential along a chain (each atom is placed in the frame of the previous three)
// and independent across chains: one lane per chain, fp64 throughout except where the reference itself
// works in float32 (the angle arrays are float32, so numpy evaluates cos / sin and the products with the
// bond length in float32 before the float64 frame multiply).  A 512-chain batch is ~40k fp64 flops per
// lane -- microseconds next to the 10 s sampling run; nothing here is worth tiling.
#include "fdmi_kernels.h"

namespace fdmi {

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ D3 unit(D3 a) {
  const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return {a.x / n, a.y / n, a.z / n};
}
// float32 sin / cos as numpy computes them on a float32 array (correctly rounded from the double result)
__device__ __forceinline__ float sin32(float v) { return (float)sin((double)v); }
__device__ __forceinline__ float cos32(float v) { return (float)cos((double)v); }

// nerf.py:145-204.  dtype follows numpy's promotion in the reference call: the torsion (and an angle /
// length that is a FEATURE) is a float32 array element, a python-float length is "weak" (stays float32
// against a float32 operand), but a DEFAULT bond angle is a python float whose cos / sin are float64 and
// pull the products to float64.
__device__ __forceinline__ D3 place(D3 a, D3 b, D3 c, bool angle_is_feature, float angle_f, double angle_default,
                                    bool length_is_feature, float length_f, double length_default, float torsion) {
  const D3 ab = sub(b, a);
  const D3 bc = unit(sub(c, b));
  const D3 n = unit(cross(ab, bc));
  const D3 nbc = cross(n, bc);
  const float bl = length_is_feature ? length_f : (float)length_default;
  const float blc = bl * cos32(torsion), bls = bl * sin32(torsion);        // float32 in every case
  double d0, d1, d2;
  if (angle_is_feature) {
    const float ca = cos32(angle_f), sa = sin32(angle_f);
    d0 = (double)(-bl * ca);
    d1 = (double)(blc * sa);
    d2 = (double)(bls * sa);
  } else {
    const double ca = cos(angle_default), sa = sin(angle_default);
    d0 = -(length_is_feature ? (double)length_f : length_default) * ca;
    d1 = (double)blc * sa;
    d2 = (double)bls * sa;
  }
  // m = [bc | nbc | n] (columns);  d = m . (d0, d1, d2) + c
  return {bc.x * d0 + nbc.x * d1 + n.x * d2 + c.x, bc.y * d0 + nbc.y * d1 + n.y * d2 + c.y,
          bc.z * d0 + nbc.z * d1 + n.z * d2 + c.z};
}

__global__ void nerf_kernel(const float* __restrict__ feats, const int* __restrict__ lens, int B, int L, int F,
                            NerfFeatures fx, int center, double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int len = lens[b];
  const float* f = feats + (size_t)b * L * F;
  double* o = out + (size_t)b * 3 * L * 3;
  auto get = [&](int idx, int i, float dflt) { return idx >= 0 ? f[(size_t)i * F + idx] : dflt; };
  D3 p0 = {17.047, 14.099, 3.625}, p1 = {16.967, 12.784, 4.338}, p2 = {15.685, 12.755, 5.133};  // nerf.py:22-24
  auto put = [&](int atom, D3 v) { o[atom * 3] = v.x; o[atom * 3 + 1] = v.y; o[atom * 3 + 2] = v.z; };
  put(0, p0); put(1, p1); put(2, p2);
  D3 sum = {p0.x + p1.x + p2.x, p0.y + p1.y + p2.y, p0.z + p1.z + p2.z};
  for (int i = 0; i + 1 < len; ++i) {
    // next N: C->N bond, CA:C:1N angle, psi_i;  next CA: N->CA bond, C:1N:1CA angle, omega_i;
    // next C: CA->C bond, N:CA:C angle AT INDEX i (reference quirk), phi_{i+1}
    const double PI = 3.14159265358979323846;
    const D3 n = place(p0, p1, p2, fx.ang_ca_c_n >= 0, get(fx.ang_ca_c_n, i, 0.f), 115.0 / 180.0 * PI,
                       fx.len_c_n >= 0, get(fx.len_c_n, i, 0.f), 1.34, f[(size_t)i * F + fx.psi]);
    const D3 ca = place(p1, p2, n, fx.ang_c_n_ca >= 0, get(fx.ang_c_n_ca, i, 0.f), 121.0 / 180.0 * PI,
                        fx.len_n_ca >= 0, get(fx.len_n_ca, i, 0.f), 1.46, f[(size_t)i * F + fx.omega]);
    const D3 c = place(p2, n, ca, fx.ang_n_ca_c >= 0, get(fx.ang_n_ca_c, i, 0.f), 109.0 / 180.0 * PI,
                       fx.len_ca_c >= 0, get(fx.len_ca_c, i, 0.f), 1.54, f[(size_t)(i + 1) * F + fx.phi]);
    put(3 * i + 3, n); put(3 * i + 4, ca); put(3 * i + 5, c);
    sum.x += n.x + ca.x + c.x; sum.y += n.y + ca.y + c.y; sum.z += n.z + ca.z + c.z;
    p0 = n; p1 = ca; p2 = c;
  }
  const int natom = 3 * len;
  if (center) {
    const double mx = sum.x / natom, my = sum.y / natom, mz = sum.z / natom;
    for (int a = 0; a < natom; ++a) { o[a * 3] -= mx; o[a * 3 + 1] -= my; o[a * 3 + 2] -= mz; }
  }
  for (int a = natom; a < 3 * L; ++a) { o[a * 3] = 0.0; o[a * 3 + 1] = 0.0; o[a * 3 + 2] = 0.0; }  // padding residues
}

void launch_nerf(const float* feats, const int* lens, int B, int L, int F, const NerfFeatures& fx, int center,
                 double* out, hipStream_t s) {
  hipLaunchKernelGGL(nerf_kernel, dim3((B + 63) / 64), dim3(64), 0, s, feats, lens, B, L, F, fx, center, out);
}

}  // namespace fdmi
