// Closed-form symmetric 3x3 eigen-decomposition and the 3-vector helpers it needs — ONE source for the host tree builder
// (csrc/host/tree_builder.cpp, bit-identical to the oracle's) and for the device tree builder (csrc/hip/tree_build.hip.h,
// SURVEY 8 row f-1), so the two cannot drift apart.  On the host this is compiled by g++ exactly as before; on the device
// the same expressions run with -ffp-contract=off, but atan2 / cos / sin come from the device math library, whose last
// bit may differ from libm's (~5 % of the solves on identical covariances): a device-built node can differ from the
// host-built one in the last bits of its eigenvectors and extents.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

#if defined(__HIPCC__)
#define MADICP_HD __host__ __device__
#else
#define MADICP_HD
#endif

namespace madicp_host {

MADICP_HD inline double sum3c(double x0, double x1, double x2) {
#ifdef MADICP_REDUX_SCALAR_ONLY
  return x0 + (x1 + x2);
#else
  return (x0 + x1) + x2;
#endif
}
MADICP_HD inline double sum3s(double x0, double x1, double x2) { return x0 + (x1 + x2); }

MADICP_HD inline double dot3c(const double* a, const double* b) { return sum3c(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
MADICP_HD inline double norm3(const double* a) { return std::sqrt(dot3c(a, a)); }
MADICP_HD inline void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// ---------------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition in closed form — the algorithm behind
// Eigen::SelfAdjointEigenSolver<Matrix3d>::computeDirect (reference call site mad_tree.cpp:59-61).
// S: row-major, lower triangle read.  w ascending, V row-major with eigenvectors in columns.
// ---------------------------------------------------------------------------------------------------
namespace detail {
struct Sym3 {  // lower triangle of a symmetric 3x3
  double a00, a10, a11, a20, a21, a22;
  MADICP_HD double at(int r, int c) const {
    if (r < c) { const int t = r; r = c; c = t; }
    return r == 0 ? a00 : (r == 1 ? (c == 0 ? a10 : a11) : (c == 0 ? a20 : (c == 1 ? a21 : a22)));
  }
  MADICP_HD void col(int c, double* o) const { o[0] = at(0, c); o[1] = at(1, c); o[2] = at(2, c); }
};
MADICP_HD inline void kernel_vector(const Sym3& m, double* res, double* repr) {
  int i0 = 0;
  double best = std::fabs(m.a00);
  if (std::fabs(m.a11) > best) { best = std::fabs(m.a11); i0 = 1; }
  if (std::fabs(m.a22) > best) { i0 = 2; }
  // columns i0, i0 + 1, i0 + 2 (mod 3) of the symmetric matrix, picked element by element with selects on NAMED scalars: an
  // index into the matrix (m.at(r, c) with a run-time column) is turned into a table in scratch memory by the device compiler —
  // three 16-byte stores and nine dependent loads per call, ~1.5 us of every eigen-solve on the GPU
  const bool is0 = i0 == 0, is1 = i0 == 1;
  const double k0[3] = {m.a00, m.a10, m.a20}, k1[3] = {m.a10, m.a11, m.a21}, k2[3] = {m.a20, m.a21, m.a22};
  double c1[3], c2[3], x1[3], x2[3];
  for (int i = 0; i < 3; ++i) {
    repr[i] = is0 ? k0[i] : (is1 ? k1[i] : k2[i]);
    c1[i] = is0 ? k1[i] : (is1 ? k2[i] : k0[i]);
    c2[i] = is0 ? k2[i] : (is1 ? k0[i] : k1[i]);
  }
  cross3(repr, c1, x1);
  cross3(repr, c2, x2);
  const double n1 = dot3c(x1, x1), n2 = dot3c(x2, x2);
  // (selects, not a pointer pick: on the device a pointer into a local array sends the array to scratch memory)
  const bool first = n1 > n2;
  const double s = std::sqrt(first ? n1 : n2);
  for (int i = 0; i < 3; ++i) res[i] = (first ? x1[i] : x2[i]) / s;
}
}  // namespace detail

MADICP_HD inline void eig3_sym(const double* S, double* w, double* V) {
  using detail::Sym3;
  const double eps = std::numeric_limits<double>::epsilon();
  const double shift = (S[0] + S[4] + S[8]) / 3.0;
  Sym3 m{S[0] - shift, S[3], S[4] - shift, S[6], S[7], S[8] - shift};
  // max |coeff| over the full (mirrored) matrix, column-major visiting order, first maximum kept
  double scale = std::fabs(m.a00);
  auto upd = [&scale](double v) {
    if (std::fabs(v) > scale) scale = std::fabs(v);
  };
  upd(m.a10); upd(m.a20); upd(m.a10); upd(m.a11); upd(m.a21); upd(m.a20); upd(m.a21); upd(m.a22);
  if (scale > 0.0) {
    m.a00 /= scale; m.a10 /= scale; m.a11 /= scale; m.a20 /= scale; m.a21 /= scale; m.a22 /= scale;
  }
  // roots of the characteristic polynomial, trigonometric form
  {
    const double inv3 = 1.0 / 3.0, sqrt3 = std::sqrt(3.0);
    const double c0 = m.a00 * m.a11 * m.a22 + 2.0 * m.a10 * m.a20 * m.a21 - m.a00 * m.a21 * m.a21 - m.a11 * m.a20 * m.a20 -
                      m.a22 * m.a10 * m.a10;
    const double c1 = m.a00 * m.a11 - m.a10 * m.a10 + m.a00 * m.a22 - m.a20 * m.a20 + m.a11 * m.a22 - m.a21 * m.a21;
    const double c2 = m.a00 + m.a11 + m.a22;
    const double c2_3 = c2 * inv3;
    double a_3 = (c2 * c2_3 - c1) * inv3;
    if (a_3 < 0.0) a_3 = 0.0;
    const double half_b = 0.5 * (c0 + c2_3 * (2.0 * c2_3 * c2_3 - c1));
    double q = a_3 * a_3 * a_3 - half_b * half_b;
    if (q < 0.0) q = 0.0;
    const double rho = std::sqrt(a_3);
    const double theta = std::atan2(std::sqrt(q), half_b) * inv3;
    const double ct = std::cos(theta), st = std::sin(theta);
    w[0] = c2_3 - rho * (ct + sqrt3 * st);
    w[1] = c2_3 - rho * (ct - sqrt3 * st);
    w[2] = c2_3 + 2.0 * rho * ct;
  }
  double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};
  if (!((w[2] - w[0]) <= eps)) {
    double d0 = w[2] - w[1];
    const double d1 = w[1] - w[0];
    int k = 0, l = 2;
    if (d0 > d1) { k = 2; l = 0; d0 = d1; }
    double vk[3], vl[3];
    Sym3 t = m;
    const double wk = (k == 0) ? w[0] : w[2], wl = (l == 0) ? w[0] : w[2];
    t.a00 -= wk; t.a11 -= wk; t.a22 -= wk;
    detail::kernel_vector(t, vk, vl);
    if (d0 <= 2.0 * eps * d1) {
      const double p = dot3c(vk, vl);
      for (int i = 0; i < 3; ++i) vl[i] -= p * vl[i];
      const double n = norm3(vl);
      for (int i = 0; i < 3; ++i) vl[i] /= n;
    } else {
      t = m;
      t.a00 -= wl; t.a11 -= wl; t.a22 -= wl;
      double dummy[3];
      detail::kernel_vector(t, vl, dummy);
    }
    for (int i = 0; i < 3; ++i) {  // l == 2 - k
      v0[i] = (k == 0) ? vk[i] : vl[i];
      v2[i] = (k == 0) ? vl[i] : vk[i];
    }
    cross3(v2, v0, v1);
    const double n = norm3(v1);
    for (int i = 0; i < 3; ++i) v1[i] /= n;
  }
  for (int i = 0; i < 3; ++i) {
    V[3 * i + 0] = v0[i];
    V[3 * i + 1] = v1[i];
    V[3 * i + 2] = v2[i];
    w[i] = w[i] * scale + shift;
  }
}

}  // namespace madicp_host
