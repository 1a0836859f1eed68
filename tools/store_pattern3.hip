// Third sweep: is it the back-to-back burst of stores per wave that costs bandwidth?
// chunk walk as the lean sampler does (1 wave / WG, TPC consecutive 8 KB tiles, 16 dwordx2
// stores per tile), with SLEEP*64 idle cycles inserted after every GROUP stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int GROUP, int SLEEP>
__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      p[64 * k] = v + k;
      if (SLEEP > 0 && (k % GROUP) == GROUP - 1) __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
}

// tile = 64 lanes x 16 samples, but lane owns 16 CONSECUTIVE samples (128 B): the wave's
// store k writes 8 B at stride 128 B (uncoalesced per instruction, full lines per tile)
__global__ void __launch_bounds__(64) k_lane_rows(double* out, long n_tiles, int tpc, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long t0 = (long)blockIdx.x * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double2* p = reinterpret_cast<double2*>(out + t * 1024 + threadIdx.x * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = make_double2(v + k, v);
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

template <int GROUP, int SLEEP>
static void run(double* out, long n_tiles, int tpc, int wpc) {
  char nm[80];
  const unsigned g = (unsigned)((n_tiles + tpc - 1) / tpc);
  const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
  CK(hipFuncSetAttribute((const void*)k_chunk<GROUP, SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  snprintf(nm, sizeof nm, "chunk tpc=%d w/CU=%d sleep %dx64 per %d st", tpc, wpc, SLEEP, GROUP);
  timeit(nm, [&] { hipLaunchKernelGGL((k_chunk<GROUP, SLEEP>), dim3(g), dim3(64), lds, 0, out, n_tiles, tpc, 1.0); }, n_tiles * 8192.0);
}

int main() {
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  double* out;
  CK(hipMalloc(&out, n * 8));
  for (int wpc : {12, 32}) {
    run<1, 0>(out, n_tiles, 32, wpc);
    run<1, 1>(out, n_tiles, 32, wpc);
    run<1, 2>(out, n_tiles, 32, wpc);
    run<1, 4>(out, n_tiles, 32, wpc);
    run<4, 4>(out, n_tiles, 32, wpc);
    run<4, 8>(out, n_tiles, 32, wpc);
    run<16, 16>(out, n_tiles, 32, wpc);
    run<16, 32>(out, n_tiles, 32, wpc);
  }
  CK(hipFuncSetAttribute((const void*)k_lane_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int wpc : {12, 32}) {
    char nm[80];
    const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
    snprintf(nm, sizeof nm, "lane-rows tpc=32 w/CU=%d", wpc);
    timeit(nm, [&] { hipLaunchKernelGGL(k_lane_rows, dim3((unsigned)((n_tiles + 31) / 32)), dim3(64), lds, 0, out, n_tiles, 32, 1.0); }, n_tiles * 8192.0);
  }
  CK(hipFree(out));
  return 0;
}
