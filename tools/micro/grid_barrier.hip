// Micro-benchmark: what an in-launch grid-wide barrier costs on gfx950 when it has to carry DATA between workgroups on
// different XCDs (agent-scope release / acquire: L2 write-back + invalidate), against a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/micro/grid_barrier.hip && /tmp/grid_barrier
// Every round each workgroup writes `words` ints (its round number) to its own slice, crosses the barrier, and reads the
// slice of the workgroup `shift` places on (another XCD: consecutive workgroups go round-robin over the XCDs); a stale
// value is counted as an error.  Variants: fence = 1 (release/acquire at agent scope), 0 (relaxed: expected to FAIL the
// check — it shows the check has teeth), and the same rounds as separate launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (FENCE) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (true) {
      const unsigned v = FENCE ? __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      if (++spins > 4000000u) { ok = false; break; }  // bounded: a grid that is not co-resident must not hang the box
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

template <int FENCE>
__global__ __launch_bounds__(256) void persistent(int* data, int words, int rounds, int shift, unsigned* ctr, int* errors) {
  const int G = gridDim.x;
  int bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    int* mine = data + (long)blockIdx.x * words;
    for (int i = threadIdx.x; i < words; i += 256) mine[i] = r;
    if (!grid_barrier<FENCE>(ctr, (unsigned)(2 * r - 1) * G)) { if (threadIdx.x == 0) atomicAdd(errors + 1, 1); return; }
    const int* other = data + (long)((blockIdx.x + shift) % G) * words;
    for (int i = threadIdx.x; i < words; i += 256) bad += other[i] != r;
    if (!grid_barrier<FENCE>(ctr, (unsigned)(2 * r) * G)) { if (threadIdx.x == 0) atomicAdd(errors + 1, 1); return; }
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ __launch_bounds__(256) void step_write(int* data, int words, int r) {
  int* mine = data + (long)blockIdx.x * words;
  for (int i = threadIdx.x; i < words; i += 256) mine[i] = r;
}
__global__ __launch_bounds__(256) void step_read(const int* data, int words, int r, int shift, int* errors) {
  const int* other = data + (long)((blockIdx.x + shift) % gridDim.x) * words;
  int bad = 0;
  for (int i = threadIdx.x; i < words; i += 256) bad += other[i] != r;
  if (bad) atomicAdd(errors, bad);
}

int main() {
  const int rounds = 200;
  int* errors; unsigned* ctr;
  CK(hipMalloc(&errors, 8)); CK(hipMalloc(&ctr, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {256, 512}) for (int words : {64, 2048, 16384}) {
    int* data; CK(hipMalloc(&data, sizeof(int) * (size_t)G * words));
    for (int variant = 0; variant < 3; ++variant) {
      CK(hipMemset(errors, 0, 8)); CK(hipMemset(ctr, 0, 4)); CK(hipMemset(data, 0, sizeof(int) * (size_t)G * words));
      float best = 1e30f;
      int h_err[2] = {0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4));
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(persistent<1>, dim3(G), dim3(256), 0, 0, data, words, rounds, 3, ctr, errors);
        else if (variant == 1) hipLaunchKernelGGL(persistent<0>, dim3(G), dim3(256), 0, 0, data, words, rounds, 3, ctr, errors);
        else for (int r = 1; r <= rounds; ++r) {
          hipLaunchKernelGGL(step_write, dim3(G), dim3(256), 0, 0, data, words, r);
          hipLaunchKernelGGL(step_read, dim3(G), dim3(256), 0, 0, (const int*)data, words, r, 3, errors);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(h_err, errors, 8, hipMemcpyDeviceToHost));
      printf("G=%3d words/wg=%5d %-28s %7.2f us per (write, sync, read, sync)   stale reads %d, timeouts %d\n", G, words,
             variant == 0 ? "persistent, release/acquire" : variant == 1 ? "persistent, relaxed (no fence)" : "two launches per round",
             1e3f * best / rounds, h_err[0], h_err[1]);
    }
    CK(hipFree(data));
  }
  return 0;
}
