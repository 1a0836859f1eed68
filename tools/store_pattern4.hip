// Fourth sweep: one-shot 4 KB workgroups (the fast `fill` shape) with the WG -> address map
// permuted, to separate "number of concurrent sequential streams" from wave lifetime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: identity; 1: multiplicative scatter; 2: G interleaved sequential streams
__global__ void __launch_bounds__(256) k_perm(double* out, long nwg, int mode, long G, double v) {
  long w = blockIdx.x;
  if (mode == 1) w = (w * 7919L) % nwg;
  else if (mode == 2) { const long per = nwg / G; const long s = w % G, i = w / G; w = s * per + i; if (i >= per) return; }
  double2* p = reinterpret_cast<double2*>(out + w * 512) + threadIdx.x;
  *p = make_double2(v, v);
}

// persistent interleaved walk, 1 wave per WG, tile = 8 KB (16 dwordx2 stores)
__global__ void __launch_bounds__(64) k_inter(double* out, long n_tiles, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

template <typename F>
static void timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 10;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
  printf("%-44s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms * 1e-9);
}

int main() {
  const long n = 256L * 10000000L;
  const long nwg = n / 512;
  double* out;
  CK(hipMalloc(&out, n * 8));
  char nm[80];
  timeit("one-shot identity", [&] { hipLaunchKernelGGL(k_perm, dim3((unsigned)nwg), dim3(256), 0, 0, out, nwg, 0, 1L, 1.0); }, n * 8.0);
  timeit("one-shot scatter", [&] { hipLaunchKernelGGL(k_perm, dim3((unsigned)nwg), dim3(256), 0, 0, out, nwg, 1, 1L, 1.0); }, n * 8.0);
  for (long G : {2L, 8L, 64L, 512L, 4096L, 32768L}) {
    snprintf(nm, sizeof nm, "one-shot %ld streams", G);
    timeit(nm, [&] { hipLaunchKernelGGL(k_perm, dim3((unsigned)nwg), dim3(256), 0, 0, out, nwg, 2, G, 1.0); }, (nwg / G) * G * 4096.0);
  }
  CK(hipFuncSetAttribute((const void*)k_inter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int wpc : {12, 32}) {
    const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
    for (unsigned mult : {1u, 4u}) {
      const unsigned g = 256u * wpc * mult;
      snprintf(nm, sizeof nm, "persistent interleaved w/CU=%d grid=%u", wpc, g);
      timeit(nm, [&] { hipLaunchKernelGGL(k_inter, dim3(g), dim3(64), lds, 0, out, n / 1024, 1.0); }, n * 8.0);
    }
  }
  CK(hipFree(out));
  return 0;
}
