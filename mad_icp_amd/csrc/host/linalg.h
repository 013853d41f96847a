// Host-side fixed-size fp64 helpers for the product's C++ layer (tree builder, Pipeline bookkeeping).
// Row-major 3x3 (double[9]) and 6x6 (double[36]); a pose is {R[9], t[3]}.
//
// The reference does this arithmetic through Eigen (absent here).  The evaluation order of every sum
// follows what Eigen's evaluators do on baseline x86-64 (SSE2), because the tree builder's branch
// decisions (leaf test, split side) must reproduce the reference's:
//   sum3c : contiguous 3-vector reduction (Vector3d::dot/squaredNorm) -> (x0 + x1) + x2
//   sum3s : strided reduction (row of a 3x3 times a vector)            -> x0 + (x1 + x2)
// -DMADICP_REDUX_SCALAR_ONLY switches sum3c to the scalar order, in lock-step with the device code.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

namespace madicp_host {

inline double sum3c(double x0, double x1, double x2) {
#ifdef MADICP_REDUX_SCALAR_ONLY
  return x0 + (x1 + x2);
#else
  return (x0 + x1) + x2;
#endif
}
inline double sum3s(double x0, double x1, double x2) { return x0 + (x1 + x2); }

inline double dot3c(const double* a, const double* b) { return sum3c(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline double norm3(const double* a) { return std::sqrt(dot3c(a, a)); }
inline void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// o = R v   (R row-major; rows of a column-major Eigen matrix are strided)
inline void matvec3(const double* R, const double* v, double* o) {
  const double x = v[0], y = v[1], z = v[2];
  for (int r = 0; r < 3; ++r) o[r] = sum3s(R[3 * r] * x, R[3 * r + 1] * y, R[3 * r + 2] * z);
}
inline void matmul3(const double* A, const double* B, double* O) {
  double T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[3 * r + c] = sum3s(A[3 * r] * B[c], A[3 * r + 1] * B[3 + c], A[3 * r + 2] * B[6 + c]);
  std::memcpy(O, T, sizeof(T));
}

struct Pose {
  double R[9];
  double t[3];
  static Pose identity() {
    Pose p;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(p.R, I, sizeof(I));
    p.t[0] = p.t[1] = p.t[2] = 0.0;
    return p;
  }
};
// X * p = R p + t (Isometry3d * Vector3d)
inline void apply(const Pose& X, const double* p, double* o) {
  double rp[3];
  matvec3(X.R, p, rp);
  for (int i = 0; i < 3; ++i) o[i] = X.t[i] + rp[i];
}
// A * B = (Ra Rb, Ra tb + ta)
inline Pose compose(const Pose& A, const Pose& B) {
  Pose r;
  matmul3(A.R, B.R, r.R);
  double rt[3];
  matvec3(A.R, B.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = rt[i] + A.t[i];
  return r;
}
inline Pose inverse(const Pose& A) {
  Pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[3 * i + j] = A.R[3 * j + i];
  double rt[3];
  matvec3(r.R, A.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = -rt[i];
  return r;
}

// lie_algebra.h:39-52 (reference) — Rodrigues with the reference's first-order branch
inline void exp_so3(const double* w, double* R) {
  const double th2 = dot3c(w, w);
  const double th = std::sqrt(th2);
  const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  if (th2 < 1e-8) {
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
    return;
  }
  double K[9], cK[9], cKK[9];
  const double omc = 2.0 * std::sin(th / 2.0) * std::sin(th / 2.0);
  const double s = std::sin(th);
  for (int i = 0; i < 9; ++i) {
    K[i] = W[i] / th;
    cK[i] = omc * K[i];
  }
  matmul3(cK, K, cKK);
  for (int i = 0; i < 9; ++i) R[i] = (((i % 4 == 0) ? 1.0 : 0.0) + s * K[i]) + cKK[i];
}

// lie_algebra.h:54-89 (reference) — only used by deskew
inline void log_so3(const double* R, double* w) {
  const double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    double f;
    if (std::fabs(R[8] + 1.0) > 1e-5) {
      f = M_PI / std::sqrt(2.0 + 2.0 * R[8]);
      w[0] = f * R[2]; w[1] = f * R[5]; w[2] = f * (1.0 + R[8]);
    } else if (std::fabs(R[4] + 1.0) > 1e-5) {
      f = M_PI / std::sqrt(2.0 + 2.0 * R[4]);
      w[0] = f * R[1]; w[1] = f * (1.0 + R[4]); w[2] = f * R[7];
    } else {
      f = M_PI / std::sqrt(2.0 + 2.0 * R[0]);
      w[0] = f * (1.0 + R[0]); w[1] = f * R[3]; w[2] = f * R[6];
    }
    return;
  }
  double mag;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-7) {
    const double th = std::acos((tr - 1.0) / 2.0);
    mag = th / (2.0 * std::sin(th));
  } else {
    mag = 0.5 - tr_3 * tr_3 / 12.0;
  }
  w[0] = mag * (R[7] - R[5]);
  w[1] = mag * (R[2] - R[6]);
  w[2] = mag * (R[3] - R[1]);
}

// ---------------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition in closed form — the algorithm behind
// Eigen::SelfAdjointEigenSolver<Matrix3d>::computeDirect (reference call site mad_tree.cpp:59-61).
// S: row-major, lower triangle read.  w ascending, V row-major with eigenvectors in columns.
// ---------------------------------------------------------------------------------------------------
namespace detail {
struct Sym3 {  // lower triangle of a symmetric 3x3
  double a00, a10, a11, a20, a21, a22;
  double at(int r, int c) const {
    if (r < c) { const int t = r; r = c; c = t; }
    return r == 0 ? a00 : (r == 1 ? (c == 0 ? a10 : a11) : (c == 0 ? a20 : (c == 1 ? a21 : a22)));
  }
  void col(int c, double* o) const { o[0] = at(0, c); o[1] = at(1, c); o[2] = at(2, c); }
};
inline void kernel_vector(const Sym3& m, double* res, double* repr) {
  int i0 = 0;
  double best = std::fabs(m.a00);
  if (std::fabs(m.a11) > best) { best = std::fabs(m.a11); i0 = 1; }
  if (std::fabs(m.a22) > best) { i0 = 2; }
  double c1[3], c2[3], x1[3], x2[3];
  m.col(i0, repr);
  m.col((i0 + 1) % 3, c1);
  m.col((i0 + 2) % 3, c2);
  cross3(repr, c1, x1);
  cross3(repr, c2, x2);
  const double n1 = dot3c(x1, x1), n2 = dot3c(x2, x2);
  const double* pick = (n1 > n2) ? x1 : x2;
  const double s = std::sqrt((n1 > n2) ? n1 : n2);
  for (int i = 0; i < 3; ++i) res[i] = pick[i] / s;
}
}  // namespace detail

inline void eig3_sym(const double* S, double* w, double* V) {
  using detail::Sym3;
  const double eps = std::numeric_limits<double>::epsilon();
  const double shift = (S[0] + S[4] + S[8]) / 3.0;
  Sym3 m{S[0] - shift, S[3], S[4] - shift, S[6], S[7], S[8] - shift};
  // max |coeff| over the full (mirrored) matrix, column-major visiting order, first maximum kept
  double scale = std::fabs(m.a00);
  const double seq[8] = {m.a10, m.a20, m.a10, m.a11, m.a21, m.a20, m.a21, m.a22};
  for (double v : seq)
    if (std::fabs(v) > scale) scale = std::fabs(v);
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
    t.a00 -= w[k]; t.a11 -= w[k]; t.a22 -= w[k];
    detail::kernel_vector(t, vk, vl);
    if (d0 <= 2.0 * eps * d1) {
      const double p = dot3c(vk, vl);
      for (int i = 0; i < 3; ++i) vl[i] -= p * vl[i];
      const double n = norm3(vl);
      for (int i = 0; i < 3; ++i) vl[i] /= n;
    } else {
      t = m;
      t.a00 -= w[l]; t.a11 -= w[l]; t.a22 -= w[l];
      double dummy[3];
      detail::kernel_vector(t, vl, dummy);
    }
    std::memcpy(k == 0 ? v0 : v2, vk, sizeof(vk));
    std::memcpy(l == 0 ? v0 : v2, vl, sizeof(vl));
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

// ---------------------------------------------------------------------------------------------------
// 6x6: LDLT solve with diagonal pivoting (Eigen::LDLT, lower) for VelEstimator (vel_estimator.cpp:95),
// and det(A^-1) via partial-pivot LU for the keyframe weight (pipeline.cpp:223).  A row-major.
// ---------------------------------------------------------------------------------------------------
inline void ldlt6_solve(const double* A, const double* rhs, double* x) {
  double m[6][6];
  int tr[6];
  double tmp[6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) m[r][c] = A[r * 6 + c];
  for (int k = 0; k < 6; ++k) {
    int big = k;
    double best = std::fabs(m[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(m[i][i]) > best) { best = std::fabs(m[i][i]); big = i; }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; ++j) std::swap(m[k][j], m[big][j]);
      for (int i = big + 1; i < 6; ++i) std::swap(m[i][k], m[i][big]);
      std::swap(m[k][k], m[big][big]);
      for (int i = k + 1; i < big; ++i) std::swap(m[i][k], m[big][i]);
    }
    if (k > 0) {
      for (int j = 0; j < k; ++j) tmp[j] = m[j][j] * m[k][j];
      double a = m[k][0] * tmp[0];
      for (int j = 1; j < k; ++j) a += m[k][j] * tmp[j];
      m[k][k] -= a;
      for (int i = k + 1; i < 6; ++i) {
        double s = m[i][0] * tmp[0];
        for (int j = 1; j < k; ++j) s += m[i][j] * tmp[j];
        m[i][k] -= s;
      }
    }
    const double akk = m[k][k];
    const bool ok = std::fabs(akk) > 0.0;
    if (k == 0 && !ok) {
      for (int j = 0; j < 6; ++j) tr[j] = j;
      break;
    }
    if (ok)
      for (int i = k + 1; i < 6; ++i) m[i][k] /= akk;
  }
  for (int i = 0; i < 6; ++i) x[i] = rhs[i];
  for (int k = 0; k < 6; ++k)
    if (tr[k] != k) std::swap(x[k], x[tr[k]]);
  for (int i = 1; i < 6; ++i) {
    double a = m[i][0] * x[0];
    for (int j = 1; j < i; ++j) a += m[i][j] * x[j];
    x[i] -= a;
  }
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < 6; ++i) x[i] = (std::fabs(m[i][i]) > tol) ? x[i] / m[i][i] : 0.0;
  for (int i = 4; i >= 0; --i) {
    double a = m[i + 1][i] * x[i + 1];
    for (int j = i + 2; j < 6; ++j) a += m[j][i] * x[j];
    x[i] -= a;
  }
  for (int k = 5; k >= 0; --k)
    if (tr[k] != k) std::swap(x[k], x[tr[k]]);
}

namespace detail {
inline int lu6(double a[6][6], int* perm) {
  int sign = 1;
  for (int i = 0; i < 6; ++i) perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(a[i][k]) > std::fabs(a[piv][k])) piv = i;
    if (piv != k) {
      for (int j = 0; j < 6; ++j) std::swap(a[k][j], a[piv][j]);
      std::swap(perm[k], perm[piv]);
      sign = -sign;
    }
    if (a[k][k] != 0.0)
      for (int i = k + 1; i < 6; ++i) a[i][k] /= a[k][k];
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j < 6; ++j) a[i][j] -= a[i][k] * a[k][j];
  }
  return sign;
}
}  // namespace detail

// det(A^-1) computed the way `H.inverse().determinant()` does it: explicit inverse, then its determinant
inline double det_of_inverse6(const double* A) {
  double a[6][6], inv[6][6];
  int perm[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) a[i][j] = A[i * 6 + j];
  detail::lu6(a, perm);
  for (int c = 0; c < 6; ++c) {
    double x[6];
    for (int i = 0; i < 6; ++i) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 1; i < 6; ++i)
      for (int j = 0; j < i; ++j) x[i] -= a[i][j] * x[j];
    for (int i = 5; i >= 0; --i) {
      for (int j = i + 1; j < 6; ++j) x[i] -= a[i][j] * x[j];
      x[i] /= a[i][i];
    }
    for (int i = 0; i < 6; ++i) inv[i][c] = x[i];
  }
  const int sign = detail::lu6(inv, perm);
  double d = sign;
  for (int i = 0; i < 6; ++i) d *= inv[i][i];
  return d;
}

}  // namespace madicp_host
