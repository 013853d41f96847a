// micro-benchmark: latency / throughput of the closed-form 3x3 eigen-solve (common/eig3.h) on gfx950, one solve per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mad_icp_amd/csrc/common/eig3.h"
#pragma clang fp contract(off)
__global__ void eig_kernel(const double* __restrict__ cov, double* __restrict__ out, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double C[9];
  for (int k = 0; k < 9; ++k) C[k] = cov[9 * (long)i + k];
  double acc = 0.0;
  for (int r = 0; r < reps; ++r) {
    double w[3], V[9];
    madicp_host::eig3_sym(C, w, V);
    acc += V[2] + w[0];
    C[0] += 1e-9 * V[0];  // dependent chain across repetitions
  }
  out[i] = acc;
}
int main() {
  const int threads = 256;
  for (int blocks : {1, 256, 1024, 4096}) {
    const long n = (long)blocks * threads;
    std::vector<double> h(9 * n);
    for (long i = 0; i < n; ++i) {
      double a = 1.0 + (i % 7) * 0.1, b = 0.5 + (i % 5) * 0.05, c = 0.01 + (i % 3) * 0.001, d = 0.02 * ((i % 11) - 5);
      double M[9] = {a, d, 0.01, d, b, 0.003, 0.01, 0.003, c};
      for (int k = 0; k < 9; ++k) h[9 * i + k] = M[k];
    }
    double *dc, *dout;
    hipMalloc(&dc, sizeof(double) * 9 * n);
    hipMalloc(&dout, sizeof(double) * n);
    hipMemcpy(dc, h.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int reps : {1, 11}) {
      hipLaunchKernelGGL(eig_kernel, dim3(blocks), dim3(threads), 0, 0, dc, dout, reps);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(eig_kernel, dim3(blocks), dim3(threads), 0, 0, dc, dout, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %5d reps %2d : %.2f us per launch\n", blocks, reps, 1e3 * ms / 20);
    }
    hipFree(dc); hipFree(dout);
  }
  return 0;
}
