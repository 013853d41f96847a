// micro-benchmark: issue cost (cycles per wave-instruction, one wave per SIMD and 4 waves per SIMD) of the
// cross-lane primitives considered for the (H,b) reduction.  hipcc --offload-arch=gfx950 -O3 xlane_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(unsigned* out, long long* cyc, int iters) {
  unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u;
  double d = a * 1e-3, e = b * 1e-3;
  __syncthreads();
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
      if (OP == 1) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
      if (OP == 2) { a = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xC, false); b = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false); }
      if (OP == 3) { a = __builtin_amdgcn_ds_bpermute(((threadIdx.x ^ 32) & 63) << 2, b); b = a + 1; }
      if (OP == 4) { d = d + e; e = e + d; }
      if (OP == 5) { a = (b > a) ? b + 1 : a; b = b ^ a; }
    }
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ (unsigned)(d + e);
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 16);
  const char* names[] = {"permlane32_swap", "permlane16_swap", "update_dpp x2", "ds_bpermute+add", "v_add_f64 x2", "cndmask-ish x2"};
  for (int threads : {64, 256, 1024}) {
    for (int op = 0; op < 6; ++op) {
      const int iters = 2000;
      void (*f)(unsigned*, long long*, int) = op == 0 ? k<0> : op == 1 ? k<1> : op == 2 ? k<2> : op == 3 ? k<3> : op == 4 ? k<4> : k<5>;
      hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
      hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
      long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      printf("threads/block %4d  %-18s  %.2f shader-cycles per unrolled op-group (clock64), wall %.2f\n", threads, names[op], double(h[0]) / (iters * 16.0), double(h[1]) / (iters * 16.0));
    }
  }
  return 0;
}
