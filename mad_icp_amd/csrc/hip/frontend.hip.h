// Scan ingest and motion compensation on the device: what happens to a scan between the sensor
// driver and MADtree::build, for callers that keep the scan in HBM (madicp_cloud_*):
//   ingest : float32 (x, y, z, intensity) records -> fp64 points, range filter, NaN filter, optional KITTI vertical
//            angle correction — apps/cpp_runners/bin_runner.cpp:126-166 of the reference
//   deskew : Pipeline::deskew, mad_icp/src/odometry/pipeline.cpp:79-123 — azimuth sort + per-chunk constant-velocity
//            compensation
// All of it is HBM-bound streaming work (24-32 bytes per point per pass); the kernels are coalesced grid-stride
// passes, the sort is rocPRIM's radix sort, scans are the three-kernel tile scans of tree_build.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tree_build.hip.h"

#pragma clang fp contract(off)

namespace madicp {
namespace fe {

// ---- upload of a cloud whose coordinates are all exactly floats (every LiDAR driver delivers float32: a KITTI .bin, a
// PointCloud2): half the bytes cross PCIe and are widened here — the same doubles, bit for bit ----------------------
__global__ __launch_bounds__(256) void cloud_widen_f32(const float* __restrict__ in, double* __restrict__ out, long n3) {
  const long i = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 3 < n3) {
    const float4 v = *reinterpret_cast<const float4*>(in + i);
    out[i] = (double)v.x; out[i + 1] = (double)v.y; out[i + 2] = (double)v.z; out[i + 3] = (double)v.w;
  } else {
    for (long k = i; k < n3; ++k) out[k] = (double)in[k];
  }
}

// ---- ingest ----------------------------------------------------------------------------------------------------
// keep[i] = the record survives bin_runner.cpp:149-151: NOT (|p| < min_range or |p| > max_range or a NaN coordinate),
// |p| evaluated in float like Eigen::Vector3f::norm() (squares summed as x^2 + (y^2 + z^2): the unrolled scalar
// reduction of a 3-vector, no packet for three floats), compared in double.
__global__ void ingest_mark(const float* __restrict__ rec, long n, int stride, double min_range, double max_range,
                            uint32_t* __restrict__ keep) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i <= n; i += (long)gridDim.x * blockDim.x) {
    uint32_t k = 0;
    if (i < n) {
      const float x = rec[i * stride], y = rec[i * stride + 1], z = rec[i * stride + 2];
      const float nrm = sqrtf(x * x + (y * y + z * z));
      const bool drop = (double)nrm < min_range || (double)nrm > max_range || isnan(x) || isnan(y) || isnan(z);
      k = drop ? 0u : 1u;
    }
    keep[i] = k;  // (entry n: the scan needs a terminator)
  }
}
// compaction in input order + conversion + the "kitti magic correction" (bin_runner.cpp:153-158): rotate the point by
// VERTICAL_ANGLE_OFFSET about the normalised p x (0,0,1).  sin / cos of the constant angle come from the host (libm).
// The rotation is Eigen's AngleAxisd::toRotationMatrix() followed by a 3x3 * vector product.
__global__ void ingest_scatter(const float* __restrict__ rec, long n, int stride, const uint32_t* __restrict__ keep,
                               const uint32_t* __restrict__ pos, int kitti, double sin_a, double cos_a, double* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (!keep[i]) continue;
    const double x = (double)rec[i * stride], y = (double)rec[i * stride + 1], z = (double)rec[i * stride + 2];
    double o0 = x, o1 = y, o2 = z;
    if (kitti) {
      // rotation_vector = p.cross((0,0,1)) = (y*1 - z*0, z*0 - x*1, x*0 - y*0)
      const double r0 = y * 1.0 - z * 0.0, r1 = z * 0.0 - x * 1.0, r2 = x * 0.0 - y * 0.0;
      double a0 = r0, a1 = r1, a2 = r2;
      const double sq = madicp_host::sum3c(r0 * r0, r1 * r1, r2 * r2);  // squaredNorm of a contiguous Vector3d
      if (sq > 0.0) {  // Eigen's normalized(): left alone when the squared norm is not positive
        const double nn = sqrt(sq);
        a0 = r0 / nn; a1 = r1 / nn; a2 = r2 / nn;
      }
      const double s0 = sin_a * a0, s1 = sin_a * a1, s2 = sin_a * a2;
      const double c1_0 = (1.0 - cos_a) * a0, c1_1 = (1.0 - cos_a) * a1, c1_2 = (1.0 - cos_a) * a2;
      double R[9];
      double tmp = c1_0 * a1;
      R[1] = tmp - s2; R[3] = tmp + s2;
      tmp = c1_0 * a2;
      R[2] = tmp + s1; R[6] = tmp - s1;
      tmp = c1_1 * a2;
      R[5] = tmp - s0; R[7] = tmp + s0;
      R[0] = c1_0 * a0 + cos_a; R[4] = c1_1 * a1 + cos_a; R[8] = c1_2 * a2 + cos_a;
      o0 = madicp_host::sum3s(R[0] * x, R[1] * y, R[2] * z);
      o1 = madicp_host::sum3s(R[3] * x, R[4] * y, R[5] * z);
      o2 = madicp_host::sum3s(R[6] * x, R[7] * y, R[8] * z);
    }
    const long d = pos[i];
    out[3 * d] = o0; out[3 * d + 1] = o1; out[3 * d + 2] = o2;
  }
}

// ---- deskew ----------------------------------------------------------------------------------------------------
// pipeline.cpp:89-97: azimuth of every point, then an ascending sort by it (radix sort of (azimuth, index) pairs)
__global__ void deskew_keys(const double* __restrict__ xyz, long n, double* __restrict__ key, uint32_t* __restrict__ idx) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    key[i] = atan2(xyz[3 * i + 1], xyz[3 * i]);
    idx[i] = (uint32_t)i;
  }
}

// pipeline.cpp:108-122 walks the sorted points from the largest azimuth down and moves to the next time chunk at most
// ONCE per point, when the point's azimuth is below the current threshold.  With d = n-1-j the position in that walk,
// T_d = number of thresholds above the point's azimuth (thresholds decrease: a binary search in the host-made table of
// the reference's own running `angle`), the chunk after point d is  k_d = min(k_{d-1} + 1, T_d)  =  d + min(1, min_{j<=d}
// (T_j - j)): a prefix minimum.  g[d] = T_d - d is written here, in WALK order (d ascending).
__global__ void deskew_targets(const double* __restrict__ key_sorted, long n, const double* __restrict__ thresholds, int n_thr,
                               int32_t* __restrict__ g) {
  for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < n; d += (long)gridDim.x * blockDim.x) {
    const double a = key_sorted[n - 1 - d];
    // thresholds[k] strictly decreasing; T = #{k : a < thresholds[k]} = first k with !(a < thresholds[k])
    int lo = 0, hi = n_thr;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (a < thresholds[m]) lo = m + 1; else hi = m;
    }
    g[d] = lo - (int)d;
  }
}

// inclusive prefix minimum of g over d, three kernels like the tile scan (1024 per workgroup)
__global__ __launch_bounds__(256) void pmin_tiles(const int32_t* __restrict__ g, long n, int32_t* __restrict__ tile_min) {
  __shared__ int s_w[4];
  const long base = (long)blockIdx.x * tb::kScanTile + threadIdx.x * 4;
  int v = 0x7fffffff;
  for (int k = 0; k < 4; ++k)
    if (base + k < n) v = min(v, g[base + k]);
  for (int m = 32; m > 0; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) tile_min[blockIdx.x] = min(min(s_w[0], s_w[1]), min(s_w[2], s_w[3]));
}
// one workgroup: tile_min[t] <- min over the tiles BEFORE t (exclusive), sequential carry over 256-wide strips
__global__ __launch_bounds__(256) void pmin_top(int32_t* __restrict__ tile_min, int n_tiles) {
  __shared__ int s_w[4];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0x7fffffff;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < n_tiles; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < n_tiles ? tile_min[i] : 0x7fffffff;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl = min(incl, o);
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int before = s_carry;                       // everything before this strip
    for (int k = 0; k < wv; ++k) before = min(before, s_w[k]);
    const int excl_in_wave = __shfl_up(incl, 1, 64);
    const int excl = (lane == 0) ? before : min(before, excl_in_wave);
    if (i < n_tiles) tile_min[i] = excl;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = min(before, incl);
    __syncthreads();
  }
}
// the chunk of every point and its compensated position: out[j] = pose[k_d] * p_sorted[j]  (pipeline.cpp:121; the
// output is in azimuth order, like the reference's).  poses: (n_poses, 12) R row-major | t, made by the host with the
// reference's own running time (pipeline.cpp:103-106,113-117).
__global__ __launch_bounds__(256) void deskew_apply(const double* __restrict__ xyz, const uint32_t* __restrict__ idx_sorted, long n,
                                                    const int32_t* __restrict__ g, const int32_t* __restrict__ tile_min,
                                                    const double* __restrict__ poses, int n_poses, double* __restrict__ out,
                                                    int32_t* __restrict__ chunk_of /* optional, walk order */) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long base = (long)blockIdx.x * tb::kScanTile + threadIdx.x * 4;  // walk positions d
  int m[4], v = 0x7fffffff;
  for (int k = 0; k < 4; ++k) {
    m[k] = (base + k < n) ? g[base + k] : 0x7fffffff;
    v = min(v, m[k]);
  }
  int incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl = min(incl, o);
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int before = tile_min[blockIdx.x];
  for (int k = 0; k < wv; ++k) before = min(before, s_w[k]);
  const int up = __shfl_up(incl, 1, 64);
  int run = (lane == 0) ? before : min(before, up);
  for (int k = 0; k < 4; ++k) {
    const long d = base + k;
    if (d >= n) break;
    run = min(run, m[k]);
    int kd = (int)d + min(1, run);  // k_d
    kd = max(0, min(kd, n_poses - 1));
    const long j = n - 1 - d;
    const long src = idx_sorted[j];
    const double x = xyz[3 * src], y = xyz[3 * src + 1], z = xyz[3 * src + 2];
    const double* P = poses + 12 * (long)kd;
#ifdef MADICP_XFORM_HOMOGENEOUS  // (Isometry3d * Vector3d in the homogeneous-product order: oracle/linalg.h apply())
    out[3 * j] = ((P[0] * x + P[1] * y) + P[2] * z) + P[9];
    out[3 * j + 1] = ((P[3] * x + P[4] * y) + P[5] * z) + P[10];
    out[3 * j + 2] = ((P[6] * x + P[7] * y) + P[8] * z) + P[11];
#else
    out[3 * j] = P[9] + madicp_host::sum3s(P[0] * x, P[1] * y, P[2] * z);
    out[3 * j + 1] = P[10] + madicp_host::sum3s(P[3] * x, P[4] * y, P[5] * z);
    out[3 * j + 2] = P[11] + madicp_host::sum3s(P[6] * x, P[7] * y, P[8] * z);
#endif
    if (chunk_of) chunk_of[d] = kd;
  }
}

}  // namespace fe
}  // namespace madicp
