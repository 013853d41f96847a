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

#include "../common/eig3.h"  // sum3c, dot3c, norm3, cross3, eig3_sym (shared with the device tree builder)

namespace madicp_host {

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
// Isometry3d * Vector3d; -DMADICP_XFORM_HOMOGENEOUS: the homogeneous-product order (oracle/linalg.h apply())
inline void apply(const Pose& X, const double* p, double* o) {
#ifdef MADICP_XFORM_HOMOGENEOUS
  const double x = p[0], y = p[1], z = p[2];
  double r[3];
  for (int i = 0; i < 3; ++i) r[i] = ((X.R[3 * i] * x + X.R[3 * i + 1] * y) + X.R[3 * i + 2] * z) + X.t[i];
  for (int i = 0; i < 3; ++i) o[i] = r[i];
#else
  double rp[3];
  matvec3(X.R, p, rp);
  for (int i = 0; i < 3; ++i) o[i] = X.t[i] + rp[i];
#endif
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
// [t; omega] -> (expSO3(omega), t)   (pipeline.cpp:101-104,147-152 of the reference)
inline Pose motion_from_twist(const double* dx) {
  Pose p;
  exp_so3(dx + 3, p.R);
  p.t[0] = dx[0]; p.t[1] = dx[1]; p.t[2] = dx[2];
  return p;
}

inline void log_so3(const double* R, double* w) {
  const double tr = R[0] + (R[4] + R[8]);  // Matrix3d::trace(): strided 3-term redux, a + (b + c) (oracle/linalg.h trace3)
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
