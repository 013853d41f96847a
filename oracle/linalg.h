// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it.
//
// Fixed-size fp64 linear algebra for the CPU restatement of the MAD-ICP hot path.
//
// The reference does all of this through Eigen (>=3.3, fallback pin 3.4.0: mad_icp/CMakeLists.txt:16-35),
// which is NOT available in this environment (no headers on disk, no network), so the arithmetic that
// lives in Eigen is restated here from its published algorithms.  PARITY UNPINNED: none of it could be
// diffed against a real Eigen build; every place where Eigen's evaluation order matters is called out.
//
// Reduction orders (x86-64 baseline = SSE2, Packet2d; the reference builds with no -march flag,
// mad_icp/CMakeLists.txt:6-8,38-40):
//   * sum over a CONTIGUOUS fixed 3-vector expression (Vector3d::dot / squaredNorm / norm): Eigen's
//     redux takes the linear-vectorised path (EIGEN_UNALIGNED_VECTORIZE=1): one Packet2d of elements
//     0,1 reduced first, then the scalar tail:  (a0*b0 + a1*b1) + a2*b2            -> dotc()
//   * coefficient of a small lazy matrix product whose lhs row is STRIDED (Matrix3d * Vector3d,
//     Matrix3d * Matrix3d): scalar unrolled redux, split in halves:  a0*b0 + (a1*b1 + a2*b2) -> dots()
// Compile with -DMADICP_REDUX_SCALAR_ONLY to force the scalar order everywhere (what a build with
// EIGEN_DONT_VECTORIZE would do); the product code has the same switch so both stay in lock-step.
// -DMADICP_XFORM_HOMOGENEOUS is a second such switch, for Isometry3d * Vector3d (see apply()).
//
// No FMA contraction anywhere: build with -ffp-contract=off (g++ -O3 on baseline x86-64 emits
// mulsd+addsd anyway; the flag makes that explicit).
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>

namespace oracle {

struct Vec3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
static_assert(sizeof(Vec3) == 24, "Vec3 must be layout-identical to Eigen::Vector3d (mad_tree.h:42)");

inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline Vec3 operator*(double s, const Vec3& a) { return {{s * a[0], s * a[1], s * a[2]}}; }

// strided / scalar redux order
inline double dots(const double a0, const double a1, const double a2, const double b0, const double b1, const double b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);
}
// contiguous / packet redux order
inline double dotc(const Vec3& a, const Vec3& b) {
#ifdef MADICP_REDUX_SCALAR_ONLY
  return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
#else
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
#endif
}
// Matrix3d::trace() = diagonal().sum(): the diagonal is a STRIDED 3-coefficient expression -> scalar unrolled redux,
// split in halves like dots():  a + (b + c)   (call site lie_algebra.h:60)
inline double trace3(double a, double b, double c) { return a + (b + c); }
inline double sqnorm(const Vec3& a) { return dotc(a, a); }
inline double norm(const Vec3& a) { return std::sqrt(sqnorm(a)); }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}

// column-major 3x3, m[c][r] — same storage as Eigen::Matrix3d
struct Mat3 {
  double m[3][3];
  double& operator()(int r, int c) { return m[c][r]; }
  const double& operator()(int r, int c) const { return m[c][r]; }
  Vec3 col(int c) const { return {{m[c][0], m[c][1], m[c][2]}}; }
  void setCol(int c, const Vec3& v) { m[c][0] = v[0]; m[c][1] = v[1]; m[c][2] = v[2]; }
  static Mat3 Identity() {
    Mat3 r{};
    r(0, 0) = r(1, 1) = r(2, 2) = 1.0;
    return r;
  }
};

// Matrix3d * Vector3d (strided lhs rows)
inline Vec3 mul(const Mat3& A, const Vec3& x) {
  Vec3 r;
  for (int i = 0; i < 3; ++i) r[i] = dots(A(i, 0), A(i, 1), A(i, 2), x[0], x[1], x[2]);
  return r;
}
// Matrix3d * Matrix3d
inline Mat3 mul(const Mat3& A, const Mat3& B) {
  Mat3 r;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) r(i, j) = dots(A(i, 0), A(i, 1), A(i, 2), B(0, j), B(1, j), B(2, j));
  return r;
}
// Matrix3d^T * Vector3d — rows of the transpose are contiguous columns of A (utils.h:89, R passed as
// eigenvectors_.transpose())
inline Vec3 mulT(const Mat3& A, const Vec3& x) {
  Vec3 r;
  for (int i = 0; i < 3; ++i) r[i] = dotc(A.col(i), x);
  return r;
}
inline Mat3 transpose(const Mat3& A) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = A(j, i);
  return r;
}

// Eigen::Isometry3d restated as (R, t).  X * p = R p + t ; X * Y = (Rx Ry, Rx ty + tx)
// (Eigen Transform.h: transform_right_product_impl / transform_transform_product_impl, Isometry mode)
struct Iso3 {
  Mat3 R;
  Vec3 t;
  static Iso3 Identity() { return {Mat3::Identity(), {{0, 0, 0}}}; }
};
// Isometry3d * Vector3d (mad_icp.cpp:78, pipeline.cpp:121).  Default: linear() * p + translation(), the product in the
// scalar order.  -DMADICP_XFORM_HOMOGENEOUS: what Eigen 3.3+'s vector specialisation of transform_right_product_impl would
// give with SSE2 packets if it multiplies the 4x4 matrix by the homogeneous 4-vector — ((r0 p0 + r1 p1) + r2 p2) + t * 1,
// row by row (candidate (a)); product and oracle switch together, like MADICP_REDUX_SCALAR_ONLY.
inline Vec3 apply(const Iso3& X, const Vec3& p) {
#ifdef MADICP_XFORM_HOMOGENEOUS
  Vec3 r;
  for (int i = 0; i < 3; ++i) r[i] = ((X.R(i, 0) * p[0] + X.R(i, 1) * p[1]) + X.R(i, 2) * p[2]) + X.t[i];
  return r;
#else
  const Vec3 rp = mul(X.R, p);
  return {{X.t[0] + rp[0], X.t[1] + rp[1], X.t[2] + rp[2]}};
#endif
}
inline Iso3 compose(const Iso3& A, const Iso3& B) {
  Iso3 r;
  r.R = mul(A.R, B.R);
  const Vec3 rt = mul(A.R, B.t);
  r.t = {{rt[0] + A.t[0], rt[1] + A.t[1], rt[2] + A.t[2]}};
  return r;
}
inline Iso3 inverse(const Iso3& A) {  // Isometry: R^T, -R^T t
  Iso3 r;
  r.R = transpose(A.R);
  const Vec3 rt = mul(r.R, A.t);
  r.t = {{-rt[0], -rt[1], -rt[2]}};
  return r;
}

struct Vec6 {
  double v[6];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
  void setZero() { for (double& x : v) x = 0.0; }
};
// column-major 6x6
struct Mat6 {
  double m[6][6];
  double& operator()(int r, int c) { return m[c][r]; }
  const double& operator()(int r, int c) const { return m[c][r]; }
  void setZero() {
    for (auto& c : m)
      for (double& x : c) x = 0.0;
  }
};

// ---------------------------------------------------------------------------------------------------
// Eigen::SelfAdjointEigenSolver<Matrix3d>::computeDirect, restated (call site mad_tree.cpp:59-61).
// Published algorithm (Eigen 3.4.0, Eigenvalues/SelfAdjointEigenSolver.h, direct_selfadjoint_eigenvalues
// <Solver,3,false>): shift by trace/3, scale by max |coeff|, closed-form trigonometric roots (ascending),
// eigenvector of the best separated eigenvalue from the larger of two cross products of columns of
// (A - lambda I), the other extreme one likewise (or by orthogonalisation when the remaining two
// eigenvalues coincide), middle one = col2 x col0, normalised.  Only the lower triangle is read.
// NaN input (single-point node, utils.h:70 divides by k-1 = 0) flows through exactly as the IEEE
// comparisons dictate: every `>`/`<=` on NaN is false.
// ---------------------------------------------------------------------------------------------------
inline void eig3_roots(const Mat3& m, double roots[3]) {
  const double s_inv3 = 1.0 / 3.0;
  const double s_sqrt3 = std::sqrt(3.0);
  const double c0 = m(0, 0) * m(1, 1) * m(2, 2) + 2.0 * m(1, 0) * m(2, 0) * m(2, 1) - m(0, 0) * m(2, 1) * m(2, 1) -
                    m(1, 1) * m(2, 0) * m(2, 0) - m(2, 2) * m(1, 0) * m(1, 0);
  const double c1 = m(0, 0) * m(1, 1) - m(1, 0) * m(1, 0) + m(0, 0) * m(2, 2) - m(2, 0) * m(2, 0) + m(1, 1) * m(2, 2) -
                    m(2, 1) * m(2, 1);
  const double c2 = m(0, 0) + m(1, 1) + m(2, 2);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = (a_over_3 < 0.0) ? 0.0 : a_over_3;  // numext::maxi(a, 0): (a < 0) ? 0 : a  -> NaN stays NaN
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = (q < 0.0) ? 0.0 : q;
  const double rho = std::sqrt(a_over_3);
  const double theta = std::atan2(std::sqrt(q), half_b) * s_inv3;
  const double cos_theta = std::cos(theta);
  const double sin_theta = std::sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0 * rho * cos_theta;
}

inline void eig3_extract_kernel(const Mat3& mat, Vec3& res, Vec3& representative) {
  // index of the largest |diagonal| entry; first maximum wins, NaN never wins (visitor uses `>`)
  int i0 = 0;
  double best = std::fabs(mat(0, 0));
  for (int i = 1; i < 3; ++i) {
    const double a = std::fabs(mat(i, i));
    if (a > best) { best = a; i0 = i; }
  }
  representative = mat.col(i0);
  const Vec3 c0 = cross(representative, mat.col((i0 + 1) % 3));
  const Vec3 c1 = cross(representative, mat.col((i0 + 2) % 3));
  const double n0 = sqnorm(c0);
  const double n1 = sqnorm(c1);
  if (n0 > n1) {
    const double s = std::sqrt(n0);
    res = {{c0[0] / s, c0[1] / s, c0[2] / s}};
  } else {
    const double s = std::sqrt(n1);
    res = {{c1[0] / s, c1[1] / s, c1[2] / s}};
  }
}

// eigenvalues ascending in evals, matching eigenvectors in the columns of evecs
inline void eig3_compute_direct(const Mat3& A, double evals[3], Mat3& evecs) {
  const double eps = std::numeric_limits<double>::epsilon();
  const double shift = (A(0, 0) + A(1, 1) + A(2, 2)) / 3.0;
  Mat3 scaled;  // selfadjointView<Lower>
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) scaled(i, j) = (i >= j) ? A(i, j) : A(j, i);
  for (int i = 0; i < 3; ++i) scaled(i, i) -= shift;
  double scale = std::fabs(scaled(0, 0));
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      const double a = std::fabs(scaled(i, j));
      if (a > scale) scale = a;
    }
  if (scale > 0.0)
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) scaled(i, j) /= scale;

  eig3_roots(scaled, evals);

  if ((evals[2] - evals[0]) <= eps) {
    evecs = Mat3::Identity();
  } else {
    Mat3 tmp = scaled;
    double d0 = evals[2] - evals[1];
    const double d1 = evals[1] - evals[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      k = 2;
      l = 0;
      d0 = d1;
    }
    Vec3 vk, vl;
    for (int i = 0; i < 3; ++i) tmp(i, i) -= evals[k];
    eig3_extract_kernel(tmp, vk, vl);
    if (d0 <= 2.0 * eps * d1) {
      const double p = dotc(vk, vl);
      vl = {{vl[0] - p * vl[0], vl[1] - p * vl[1], vl[2] - p * vl[2]}};
      const double n = norm(vl);
      vl = {{vl[0] / n, vl[1] / n, vl[2] / n}};
    } else {
      tmp = scaled;
      for (int i = 0; i < 3; ++i) tmp(i, i) -= evals[l];
      Vec3 dummy;
      eig3_extract_kernel(tmp, vl, dummy);
    }
    evecs.setCol(k, vk);
    evecs.setCol(l, vl);
    const Vec3 mid = cross(evecs.col(2), evecs.col(0));
    const double n = norm(mid);
    evecs.setCol(1, {{mid[0] / n, mid[1] / n, mid[2] / n}});
  }
  for (int i = 0; i < 3; ++i) evals[i] = evals[i] * scale + shift;
}

// ---------------------------------------------------------------------------------------------------
// Eigen::LDLT<Matrix6d> (Lower) factor + solve, restated (call sites mad_icp.cpp:111,
// vel_estimator.cpp:95).  Published algorithm (Eigen 3.4.0, Cholesky/LDLT.h: ldlt_inplace<Lower>::
// unblocked and LDLT::_solve_impl): diagonal pivoting on the largest |remaining diagonal|, symmetric
// row/column swap restricted to the lower triangle, right-looking update through temp = D*L_k^T,
// division by the pivot when it is non-zero; solve = P, unit-lower forward substitution, pseudo-inverse
// of D with tolerance numeric_limits<double>::min(), unit-upper back substitution, P^T.
// Inner sums run sequentially in index order (Eigen's dynamic-size blocks inside the factorisation use
// its default scalar traversal / gemv kernels; last-bit differences there are not pinned).
// ---------------------------------------------------------------------------------------------------
inline Vec6 ldlt6_solve(const Mat6& A, const Vec6& rhs) {
  const int n = 6;
  Mat6 mat = A;
  int transp[6];
  double temp[6];

  for (int k = 0; k < n; ++k) {
    int big = k;
    double best = std::fabs(mat(k, k));
    for (int i = k + 1; i < n; ++i) {
      const double a = std::fabs(mat(i, i));
      if (a > best) { best = a; big = i; }
    }
    transp[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; ++j) { const double t = mat(k, j); mat(k, j) = mat(big, j); mat(big, j) = t; }
      for (int i = 0; i < s; ++i) {
        const double t = mat(big + 1 + i, k);
        mat(big + 1 + i, k) = mat(big + 1 + i, big);
        mat(big + 1 + i, big) = t;
      }
      { const double t = mat(k, k); mat(k, k) = mat(big, big); mat(big, big) = t; }
      for (int i = k + 1; i < big; ++i) {
        const double t = mat(i, k);
        mat(i, k) = mat(big, i);
        mat(big, i) = t;
      }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = mat(j, j) * mat(k, j);
      double acc = 0.0;
      for (int j = 0; j < k; ++j) acc = (j == 0) ? mat(k, 0) * temp[0] : acc + mat(k, j) * temp[j];
      mat(k, k) -= acc;
      for (int i = 0; i < rs; ++i) {
        double a = 0.0;
        for (int j = 0; j < k; ++j) a = (j == 0) ? mat(k + 1 + i, 0) * temp[0] : a + mat(k + 1 + i, j) * temp[j];
        mat(k + 1 + i, k) -= a;
      }
    }
    const double akk = mat(k, k);
    const bool pivot_ok = std::fabs(akk) > 0.0;
    if (k == 0 && !pivot_ok) {
      for (int j = 0; j < n; ++j) transp[j] = j;
      break;
    }
    if (rs > 0 && pivot_ok)
      for (int i = 0; i < rs; ++i) mat(k + 1 + i, k) /= akk;
  }

  Vec6 x = rhs;
  for (int k = 0; k < n; ++k)
    if (transp[k] != k) { const double t = x[k]; x[k] = x[transp[k]]; x[transp[k]] = t; }
  for (int i = 1; i < n; ++i) {  // L y = Pb, unit diagonal
    double acc = mat(i, 0) * x[0];
    for (int j = 1; j < i; ++j) acc += mat(i, j) * x[j];
    x[i] -= acc;
  }
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < n; ++i) {
    if (std::fabs(mat(i, i)) > tol) x[i] /= mat(i, i);
    else x[i] = 0.0;
  }
  for (int i = n - 2; i >= 0; --i) {  // L^T z = y
    double acc = mat(i + 1, i) * x[i + 1];
    for (int j = i + 2; j < n; ++j) acc += mat(j, i) * x[j];
    x[i] -= acc;
  }
  for (int k = n - 1; k >= 0; --k)
    if (transp[k] != k) { const double t = x[k]; x[k] = x[transp[k]]; x[transp[k]] = t; }
  return x;
}

// Matrix6d::inverse().determinant() (call site pipeline.cpp:223).  Eigen routes both through
// PartialPivLU for sizes > 4; restated as det(A^-1) computed from an explicit partial-pivot LU inverse
// followed by a partial-pivot LU determinant.  Only ever compared between frames (pipeline.cpp:238).
inline bool lu6_factor(double a[6][6] /*row-major a[r][c]*/, int perm[6], int& sign) {
  sign = 1;
  for (int i = 0; i < 6; ++i) perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = std::fabs(a[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(a[i][k]) > best) { best = std::fabs(a[i][k]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { const double t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
      sign = -sign;
    }
    if (a[k][k] != 0.0) {
      for (int i = k + 1; i < 6; ++i) a[i][k] /= a[k][k];
    }
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j < 6; ++j) a[i][j] -= a[i][k] * a[k][j];
  }
  return true;
}
inline double det6(const Mat6& A) {
  double a[6][6];
  int perm[6], sign;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) a[i][j] = A(i, j);
  lu6_factor(a, perm, sign);
  double d = sign;
  for (int i = 0; i < 6; ++i) d *= a[i][i];
  return d;
}
inline Mat6 inverse6(const Mat6& A) {
  double a[6][6];
  int perm[6], sign;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) a[i][j] = A(i, j);
  lu6_factor(a, perm, sign);
  Mat6 inv;
  for (int c = 0; c < 6; ++c) {
    double x[6];
    for (int i = 0; i < 6; ++i) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 1; i < 6; ++i)
      for (int j = 0; j < i; ++j) x[i] -= a[i][j] * x[j];
    for (int i = 5; i >= 0; --i) {
      for (int j = i + 1; j < 6; ++j) x[i] -= a[i][j] * x[j];
      x[i] /= a[i][i];
    }
    for (int i = 0; i < 6; ++i) inv(i, c) = x[i];
  }
  return inv;
}

// ---------------------------------------------------------------------------------------------------
// lie_algebra.h:33-89
// ---------------------------------------------------------------------------------------------------
inline Mat3 skew(const Vec3& v) {  // lie_algebra.h:33-37
  Mat3 S;
  S(0, 0) = 0.0;   S(0, 1) = -v[2]; S(0, 2) = v[1];
  S(1, 0) = v[2];  S(1, 1) = 0.0;   S(1, 2) = -v[0];
  S(2, 0) = -v[1]; S(2, 1) = v[0];  S(2, 2) = 0.0;
  return S;
}

inline Mat3 expMapSO3(const Vec3& omega) {  // lie_algebra.h:39-52
  Mat3 R;
  const double theta_square = dotc(omega, omega);
  const double theta = std::sqrt(theta_square);
  const Mat3 W = skew(omega);
  if (theta_square < 1e-8) {
    R = Mat3::Identity();
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) R(i, j) = R(i, j) + W(i, j);
  } else {
    Mat3 K;
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) K(i, j) = W(i, j) / theta;
    const double one_minus_cos = 2.0 * std::sin(theta / 2.0) * std::sin(theta / 2.0);
    const double s = std::sin(theta);
    // I + sin(theta)*K + (one_minus_cos*K)*K : Eigen evaluates `one_minus_cos * K * K` left to right
    Mat3 cK;
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) cK(i, j) = one_minus_cos * K(i, j);
    const Mat3 cKK = mul(cK, K);
    const Mat3 I = Mat3::Identity();
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) R(i, j) = (I(i, j) + s * K(i, j)) + cKK(i, j);
  }
  return R;
}

inline Vec3 logMapSO3(const Mat3& R) {  // lie_algebra.h:54-89 (only used by deskew, pipeline.cpp:85)
  const double R11 = R(0, 0), R12 = R(0, 1), R13 = R(0, 2);
  const double R21 = R(1, 0), R22 = R(1, 1), R23 = R(1, 2);
  const double R31 = R(2, 0), R32 = R(2, 1), R33 = R(2, 2);
  const double tr = trace3(R11, R22, R33);
  const double pi = M_PI, two = 2.0;
  Vec3 omega;
  if (tr + 1.0 < 1e-10) {
    // `abs` at lie_algebra.h:69,71 is header-dependent like mad_icp.cpp:93: fabs intended
    if (std::fabs(R33 + 1.0) > 1e-5) {
      omega = (pi / std::sqrt(two + two * R33)) * Vec3{{R13, R23, 1.0 + R33}};
    } else if (std::fabs(R22 + 1.0) > 1e-5) {
      omega = (pi / std::sqrt(two + two * R22)) * Vec3{{R12, 1.0 + R22, R32}};
    } else {
      omega = (pi / std::sqrt(two + two * R11)) * Vec3{{1.0 + R11, R21, R31}};
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-7) {
      const double theta = std::acos((tr - 1.0) / two);
      magnitude = theta / (two * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 * tr_3 / 12.0;
    }
    omega = magnitude * Vec3{{R32 - R23, R13 - R31, R21 - R12}};
  }
  return omega;
}

}  // namespace oracle
