// Microbenchmark behind DESIGN.md section 5: what HBM write rate do different store
// patterns reach on MI355X, with no arithmetic at all?  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// Patterns (all write the same 256 x 1e7 doubles = 20.48 GB):
//   fill      : grid-stride dwordx4 stores, neighbouring workgroups write neighbouring bytes
//   chunk2    : 1 wave / WG, each wave walks TPC consecutive 8 KB tiles, 16 dwordx2 stores per tile
//               (the lean sampler's pattern)
//   chunk4    : same walk, 8 dwordx4 stores per tile (lane owns 2 adjacent samples)
//   inter2    : 1 wave / WG, wave w writes tiles w, w+G, w+2G, ... (dwordx2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_fill(double2* out, long n2, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < n2; i += stride) out[i] = make_double2(v, v);
}

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_chunk2(double* out, long n_tiles, int tpc, double v) {
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

// occupancy pinned by a dynamic-LDS request: waves/CU = floor(160 KB / lds_bytes)
__global__ void __launch_bounds__(64) k_chunk2_lds(double* out, long n_tiles, int tpc, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;   // keep the allocation alive
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_chunk4(double* out, long n_tiles, int tpc, double v) {
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double2* p = reinterpret_cast<double2*>(out + t * 1024) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[64 * k] = make_double2(v + k, v);
  }
}

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_inter2(double* out, long n_tiles, double v) {
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

// 4 waves per WG, WG walks consecutive 32 KB super-tiles
__global__ void __launch_bounds__(256) k_wg4(double* out, long n_tiles, int tpc, double v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long t0 = (long)blockIdx.x * tpc * 4;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt * 4 + wave;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

template <typename F>
static void timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 10;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= R;
  printf("%-28s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms * 1e-9);
}

int main() {
  CK(hipFuncSetAttribute((const void*)k_chunk2_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  double* out;
  CK(hipMalloc(&out, n * 8));
  const double bytes = (double)n * 8;
  timeit("fill x4 grid-stride 8192wg", [&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double2*)out, n / 2, 1.0); }, bytes);
  timeit("fill x4 one-shot", [&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)(n / 2 / 256)), dim3(256), 0, 0, (double2*)out, n / 2, 1.0); }, bytes);
  for (int tpc : {1, 4, 8, 32, 128}) {
    char nm[64];
    const unsigned g = (unsigned)((n_tiles + tpc - 1) / tpc);
    snprintf(nm, sizeof nm, "chunk2 occ3 tpc=%d", tpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chunk2<3>, dim3(g), dim3(64), 0, 0, out, n_tiles, tpc, 1.0); }, bytes);
    snprintf(nm, sizeof nm, "chunk2 occ8 tpc=%d", tpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chunk2<8>, dim3(g), dim3(64), 0, 0, out, n_tiles, tpc, 1.0); }, bytes);
    snprintf(nm, sizeof nm, "chunk4 occ3 tpc=%d", tpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chunk4<3>, dim3(g), dim3(64), 0, 0, out, n_tiles, tpc, 1.0); }, bytes);
    snprintf(nm, sizeof nm, "wg4 tpc=%d", tpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_wg4, dim3((g + 3) / 4), dim3(256), 0, 0, out, n_tiles, tpc, 1.0); }, bytes);
  }
  for (int wpc : {4, 8, 12, 16, 24, 32}) {
    char nm[64];
    const int tpc = 32;
    const unsigned g = (unsigned)((n_tiles + tpc - 1) / tpc);
    const unsigned lds = (160 * 1024 / wpc) & ~255u;
    snprintf(nm, sizeof nm, "chunk2 tpc=32 waves/CU=%d", wpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chunk2_lds, dim3(g), dim3(64), lds, 0, out, n_tiles, tpc, 1.0); }, bytes);
  }
  for (unsigned g : {3072u, 6144u, 12288u, 65536u}) {
    char nm[64];
    snprintf(nm, sizeof nm, "inter2 occ3 grid=%u", g);
    timeit(nm, [&] { hipLaunchKernelGGL(k_inter2<3>, dim3(g), dim3(64), 0, 0, out, n_tiles, 1.0); }, bytes);
  }
  CK(hipFree(out));
  return 0;
}
