// tools/micro/launch_gap_probe.hip -- round 6: what makes the ~5.6 us idle gaps in front of some launches of the loop (profiles/
// r06_iteration_kernel_sequence.txt: adam -> fwd, fwd -> pair, dgrad -> bwd, bwd -> wgrad 10.4, wgrad -> adam; none between the
// convolution / contextual launches)?  Pairs [A ; B] of dependent launches on one stream, 400 pairs between two events; A is a
// 256-workgroup kernel of ~10 us; B varies in ONE property.  Printed: time per pair.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/launch_gap_probe.hip -o build_ab/launch_gap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { float v[700]; };      // 2.8 KB of kernel arguments
struct Small { float v[8]; };

__device__ __forceinline__ void spin(long long cycles) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
}
template <typename Args>
__global__ void k_args(Args a, float* out, long long cycles) {
  spin(cycles);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a.v[3];
}
__global__ void k_lds(float* out, long long cycles) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  spin(cycles);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = sm[5];
}
// every thread writes `per_thread` float4s (streaming or plain): bytes = grid * 256 * per_thread * 16
template <bool NT>
__global__ void k_write(float4* buf, int per_thread, long long cycles) {
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  const float4 v = {1.f, 2.f, 3.f, 4.f};
  for (int i = 0; i < per_thread; ++i) {
    if (NT) __builtin_nontemporal_store(v.x, &buf[base + i * stride].x), __builtin_nontemporal_store(v.y, &buf[base + i * stride].y),
            __builtin_nontemporal_store(v.z, &buf[base + i * stride].z), __builtin_nontemporal_store(v.w, &buf[base + i * stride].w);
    else buf[base + i * stride] = v;
  }
  spin(cycles);
}
__global__ void k_read(const float4* buf, float* out, int per_thread) {
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  float s = 0.f;
  for (int i = 0; i < per_thread; ++i) s += buf[base + i * stride].x;
  if (s == 12345.f) out[1] = s;
}

int main() {
  float* out; float4* buf;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&buf, 512u << 20));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long cyc = 1000;       // 100 MHz ticks: ~10 us
  Big big{}; Small sml{};
  auto timeit = [&](const char* name, auto launchB) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 400; ++i) {
        hipLaunchKernelGGL((k_args<Small>), dim3(256), dim3(256), 0, 0, sml, out, cyc);
        launchB();
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-64s %7.2f us per [A ; B] pair\n", name, ms * 1e3 / 400);
    }
    return 0;
  };
  if (timeit("B = A (small args, no LDS, no writes)", [&] { hipLaunchKernelGGL((k_args<Small>), dim3(256), dim3(256), 0, 0, sml, out, cyc); })) return 1;
  if (timeit("B with 2.8 KB of kernel arguments", [&] { hipLaunchKernelGGL((k_args<Big>), dim3(256), dim3(256), 0, 0, big, out, cyc); })) return 1;
  if (timeit("B with 66 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 66 * 1024, 0, out, cyc); })) return 1;
  if (timeit("B with 160 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 160 * 1024, 0, out, cyc); })) return 1;
  if (timeit("B with 416 workgroups of 78 KB LDS (two per CU)", [&] { hipLaunchKernelGGL(k_lds, dim3(416), dim3(256), 78 * 1024, 0, out, cyc); })) return 1;
  for (int mb : {8, 32, 64, 128, 256}) {
    const int per_thread = mb * 1024 * 1024 / (1024 * 256 * 16);
    char nm[128];
    snprintf(nm, sizeof nm, "B writes %3d MB (plain stores), then A", mb);
    if (timeit(nm, [&] { hipLaunchKernelGGL((k_write<false>), dim3(1024), dim3(256), 0, 0, buf, per_thread, 0LL); })) return 1;
    snprintf(nm, sizeof nm, "B writes %3d MB (non-temporal stores), then A", mb);
    if (timeit(nm, [&] { hipLaunchKernelGGL((k_write<true>), dim3(1024), dim3(256), 0, 0, buf, per_thread, 0LL); })) return 1;
    snprintf(nm, sizeof nm, "B reads %3d MB (cold-ish), then A", mb);
    if (timeit(nm, [&] { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, 0, buf, out, per_thread); })) return 1;
  }
  // the loop's regime: both kernels long (~40 us, 416 workgroups), only B's kernel-argument block differs
  struct Mid { float v[120]; };     // 480 B
  Mid mid{};
  const long long cyc2 = 80000;     // shader clocks: ~40 us
  auto pair2 = [&](const char* name, auto launchB) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 200; ++i) {
        hipLaunchKernelGGL((k_args<Small>), dim3(416), dim3(256), 0, 0, sml, out, cyc2);
        launchB();
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-64s %7.2f us per [A ; B] pair\n", name, ms * 1e3 / 200);
    }
    return 0;
  };
  if (pair2("long A ; long B, 32 B of arguments", [&] { hipLaunchKernelGGL((k_args<Small>), dim3(416), dim3(256), 0, 0, sml, out, cyc2); })) return 1;
  if (pair2("long A ; long B, 480 B of arguments", [&] { hipLaunchKernelGGL((k_args<Mid>), dim3(416), dim3(256), 0, 0, mid, out, cyc2); })) return 1;
  if (pair2("long A ; long B, 2.8 KB of arguments", [&] { hipLaunchKernelGGL((k_args<Big>), dim3(416), dim3(256), 0, 0, big, out, cyc2); })) return 1;
  if (pair2("long A ; long B, 32 B of arguments (again)", [&] { hipLaunchKernelGGL((k_args<Small>), dim3(416), dim3(256), 0, 0, sml, out, cyc2); })) return 1;
  printf("done\n");
  return 0;
}
