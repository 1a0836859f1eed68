// Fifth sweep: one-shot 1-wave workgroups in linear order, each writing TPC consecutive 8 KB
// tiles, with the number of resident waves per CU pinned by an LDS request.  Rate as a
// function of the "active write window" = resident waves x bytes per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NS>
__global__ void __launch_bounds__(64) k_tile(double* out, long n_tiles, int tpc, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * (64 * NS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; ++k) p[64 * k] = v + k;
  }
}

// ticket variant: persistent waves take the next tile from a global counter
__global__ void __launch_bounds__(64) k_ticket(double* out, long n_tiles, int tpc, unsigned long long* ctr, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  for (;;) {
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = atomicAdd(ctr, (unsigned long long)tpc);
    t0 = __shfl(t0, 0);
    if ((long)t0 >= n_tiles) break;
    for (int tt = 0; tt < tpc; ++tt) {
      const long t = (long)t0 + tt;
      if (t >= n_tiles) break;
      double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
    }
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
  unsigned long long* ctr;
  CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&ctr, 8));
  CK(hipFuncSetAttribute((const void*)k_tile<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tile<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tile<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_ticket, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("one-shot 1-wave WGs, linear order: ms (TB/s)\n%-22s", "waves/CU ->");
  const int wpcs[] = {2, 4, 8, 12, 16, 32};
  for (int wpc : wpcs) printf("%16d", wpc);
  printf("\n");
  auto row = [&](const char* nm, auto kern, int ns, int tpc) {
    printf("%-22s", nm);
    for (int wpc : wpcs) {
      const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
      const long n_tiles = n / (64 * ns);
      const unsigned g = (unsigned)((n_tiles + tpc - 1) / tpc);
      float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(64), lds, 0, out, n_tiles, tpc, 1.0); });
      printf("  %6.3f (%5.2f)", ms, n * 8.0 / ms * 1e-9);
    }
    printf("\n");
  };
  row("NS=4  tpc=1 (2KB)", k_tile<4>, 4, 1);
  row("NS=8  tpc=1 (4KB)", k_tile<8>, 8, 1);
  row("NS=16 tpc=1 (8KB)", k_tile<16>, 16, 1);
  row("NS=16 tpc=2 (16KB)", k_tile<16>, 16, 2);
  row("NS=16 tpc=4 (32KB)", k_tile<16>, 16, 4);
  row("NS=16 tpc=8 (64KB)", k_tile<16>, 16, 8);
  row("NS=16 tpc=32 (256KB)", k_tile<16>, 16, 32);
  for (int tpc : {1, 2, 4, 8}) {
    printf("ticket tpc=%-11d", tpc);
    for (int wpc : wpcs) {
      const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
      float ms = timeit([&] {
        hipMemsetAsync(ctr, 0, 8, 0);
        hipLaunchKernelGGL(k_ticket, dim3(256u * wpc), dim3(64), lds, 0, out, n / 1024, tpc, ctr, 1.0);
      });
      printf("  %6.3f (%5.2f)", ms, n * 8.0 / ms * 1e-9);
    }
    printf("\n");
  }
  return 0;
}
