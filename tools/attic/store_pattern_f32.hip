// Store-width probe for the fp32 sampler: the lean kernel writes a 1024-sample wave tile as
// 16 dword stores (256 B per wave instruction).  Same walk (1 wave / WG, TPC consecutive tiles,
// XCD-aware chunk map), same bytes (256 x 1e7 floats = 10.24 GB), three store widths, with a
// few dependent FMAs per element in between to mimic the evaluation.
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern_f32.hip -o /tmp/spf && /tmp/spf
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ long chunk_of(long n_chunks) {
  const long b = blockIdx.x, per = (n_chunks + 7) >> 3;
  const long c = (b & 7) * per + (b >> 3);
  return c < n_chunks ? c : -1;
}

template <int VEC, int WORK>
__global__ void __launch_bounds__(64, 3) k_walk(float* out, long n_tiles, long n_chunks, int tpc, float v) {
  const long c = chunk_of(n_chunks);
  if (c < 0) return;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = c * tpc + tt;
    if (t >= n_tiles) break;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a = v + k + threadIdx.x;
#pragma unroll
      for (int w = 0; w < WORK; ++w) a = a * 1.0001f + 0.5f;
      acc[k] = a;
    }
    float* base = out + t * 1024;
    if (VEC == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) base[threadIdx.x + 64 * k] = acc[k];
    } else if (VEC == 2) {
      float2* p = reinterpret_cast<float2*>(base) + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 8; ++k) p[64 * k] = make_float2(acc[2 * k], acc[2 * k + 1]);
    } else {
      float4* p = reinterpret_cast<float4*>(base) + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) p[64 * k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
  }
}

template <typename F>
static int timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 20;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= R;
  printf("%-36s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms * 1e-9);
  return 0;
}

int main() {
  const long n = 256L * 10000000L, n_tiles = n / 1024;
  float* out;
  CK(hipMalloc(&out, n * 4));
  const double bytes = (double)n * 4;
  for (int tpc : {4, 8, 16}) {
    const long n_chunks = (n_tiles + tpc - 1) / tpc;
    const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
    char nm[64];
#define RUN(V, W) snprintf(nm, sizeof nm, "f32 vec%d work%d tpc=%d", V, W, tpc); \
    timeit(nm, [&] { hipLaunchKernelGGL((k_walk<V, W>), dim3(g), dim3(64), 0, 0, out, n_tiles, n_chunks, tpc, 1.0f); }, bytes);
    RUN(1, 0) RUN(2, 0) RUN(4, 0) RUN(1, 12) RUN(2, 12) RUN(4, 12)
  }
  CK(hipFree(out));
  return 0;
}
