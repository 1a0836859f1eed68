// Store-pattern probe for SLOW boxes (sampler at 3.58 ms where torch fill does 3.03): the sampler's
// walk (1 wave / workgroup, 8 tiles of 8 KB each, XCD-aware chunk map) against
//   B  4 waves / workgroup walking tiles of 32 KB (256 lanes x 16 samples), tpc tiles each
//   C  1 wave / workgroup, 16-B stores (each lane two adjacent samples), tiles of 16 KB
//   D  1 wave / workgroup, tiles handed out ROUND-ROBIN inside a group of G consecutive chunks: wave w of
//      the group writes tiles w, w + G, ... of the group's span (what a linear write front would need)
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern12.hip -o /tmp/sp12 && /tmp/sp12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ long xcd_chunk(long n_chunks) {
  const long b = blockIdx.x, per = (n_chunks + 7) >> 3;
  const long c = (b & 7) * per + (b >> 3);
  return c < n_chunks ? c : -1;
}

// A / B: WAVES waves per workgroup, tile = WAVES * 1024 samples
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_walk(double* out, long n_tiles, int tpc, long n_chunks, double v) {
  const long c = xcd_chunk(n_chunks);
  if (c < 0) return;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = c * tpc + tt;
    if (t >= n_tiles) break;
    double* p = out + t * (1024L * WAVES) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * WAVES * k] = v + k;
  }
}

// C: 16-byte stores
__global__ void __launch_bounds__(64) k_walk16(double* out, long n_tiles, int tpc, long n_chunks, double v) {
  const long c = xcd_chunk(n_chunks);
  if (c < 0) return;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = c * tpc + tt;
    if (t >= n_tiles) break;
    double2* p = reinterpret_cast<double2*>(out + t * 2048) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = make_double2(v + k, v);
  }
}

// D: round-robin tiles inside groups of G chunks
__global__ void __launch_bounds__(64) k_rr(double* out, long n_tiles, int tpc, long n_chunks, int G, double v) {
  const long c = xcd_chunk(n_chunks);
  if (c < 0) return;
  const long grp = c / G, w = c % G;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = grp * G * tpc + (long)tt * G + w;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 8;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / R;
}

int main() {
  const long n = 256L * 10000000L;
  double* out;
  CK(hipMalloc(&out, n * 8));
  auto grid = [](long n_chunks) { return (unsigned)(((n_chunks + 7) / 8) * 8); };
  const unsigned lds12 = (160 * 1024 / 12) & ~255u;          // 12 one-wave workgroups per CU, like the sampler
  {
    const long n_tiles = n / 1024;
    for (int tpc : {8}) {
      const long nc = (n_tiles + tpc - 1) / tpc;
      float a = timeit([&] { hipLaunchKernelGGL(k_walk<1>, dim3(grid(nc)), dim3(64), lds12, 0, out, n_tiles, tpc, nc, 1.0); });
      printf("A  1 wave/WG  8 KB tiles  tpc=%2d (12 waves/CU): %6.3f ms %5.2f TB/s\n", tpc, a, n * 8.0 / a * 1e-9);
    }
    for (int G : {2, 4, 8, 16, 48, 384}) {
      const int tpc = 8;
      const long nc = (n_tiles + tpc - 1) / tpc;
      float a = timeit([&] { hipLaunchKernelGGL(k_rr, dim3(grid(nc)), dim3(64), lds12, 0, out, n_tiles, tpc, nc, G, 1.0); });
      printf("D  round-robin in groups of %3d chunks, tpc=8:          %6.3f ms %5.2f TB/s\n", G, a, n * 8.0 / a * 1e-9);
    }
  }
  {
    const long n_tiles = n / 4096;
    for (int tpc : {1, 2, 4, 8}) {
      const long nc = (n_tiles + tpc - 1) / tpc;
      for (unsigned lds : {(160u * 1024 / 3) & ~255u, (160u * 1024 / 4) & ~255u, 0u}) {
        float a = timeit([&] { hipLaunchKernelGGL(k_walk<4>, dim3(grid(nc)), dim3(256), lds, 0, out, n_tiles, tpc, nc, 1.0); });
        printf("B  4 waves/WG 32 KB tiles tpc=%2d lds %6u:            %6.3f ms %5.2f TB/s\n", tpc, lds, a, n * 8.0 / a * 1e-9);
      }
    }
  }
  {
    const long n_tiles = n / 2048;
    for (int tpc : {4, 8}) {
      const long nc = (n_tiles + tpc - 1) / tpc;
      float a = timeit([&] { hipLaunchKernelGGL(k_walk16, dim3(grid(nc)), dim3(64), lds12, 0, out, n_tiles, tpc, nc, 1.0); });
      printf("C  1 wave/WG 16-B stores 16 KB tiles tpc=%2d:           %6.3f ms %5.2f TB/s\n", tpc, a, n * 8.0 / a * 1e-9);
    }
  }
  return 0;
}
