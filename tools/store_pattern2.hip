// Second store-pattern sweep: which property of `fill` makes it reach 6.9 TB/s?
//   one-shot kernels: THREADS per WG, VEC doubles per lane per store, S stores per thread;
//   a WG writes one contiguous span of THREADS*VEC*S doubles, store s of the WG covers
//   sub-span s (so each store instruction of a wave is a contiguous 64*VEC*8 bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int THREADS, int VEC, int S>
__global__ void __launch_bounds__(THREADS) k_span(double* out, long n, double v) {
  const long base = (long)blockIdx.x * (THREADS * VEC * S) + (long)threadIdx.x * VEC;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const long i = base + (long)s * THREADS * VEC;
    if (i + VEC <= n) {
      if (VEC == 2) *reinterpret_cast<double2*>(out + i) = make_double2(v + s, v);
      else out[i] = v + s;
    }
  }
}

// same, but a little ALU work + a dependent wait between stores (mimics compute between stores)
template <int THREADS, int VEC, int S>
__global__ void __launch_bounds__(THREADS) k_span_nt(double* out, long n, double v) {
  const long base = (long)blockIdx.x * (THREADS * VEC * S) + (long)threadIdx.x * VEC;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const long i = base + (long)s * THREADS * VEC;
    if (i + VEC <= n) {
      if (VEC == 2) __builtin_nontemporal_store(v + s, out + i), __builtin_nontemporal_store(v, out + i + 1);
      else __builtin_nontemporal_store(v + s, out + i);
    }
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
  printf("%-36s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms * 1e-9);
}

template <int THREADS, int VEC, int S>
static void run(double* out, long n) {
  char nm[64];
  const long span = (long)THREADS * VEC * S;
  const unsigned g = (unsigned)((n + span - 1) / span);
  snprintf(nm, sizeof nm, "span thr=%d vec=%d S=%d", THREADS, VEC, S);
  timeit(nm, [&] { hipLaunchKernelGGL((k_span<THREADS, VEC, S>), dim3(g), dim3(THREADS), 0, 0, out, n, 1.0); }, n * 8.0);
}
template <int THREADS, int VEC, int S>
static void run_nt(double* out, long n) {
  char nm[64];
  const long span = (long)THREADS * VEC * S;
  const unsigned g = (unsigned)((n + span - 1) / span);
  snprintf(nm, sizeof nm, "span-nt thr=%d vec=%d S=%d", THREADS, VEC, S);
  timeit(nm, [&] { hipLaunchKernelGGL((k_span_nt<THREADS, VEC, S>), dim3(g), dim3(THREADS), 0, 0, out, n, 1.0); }, n * 8.0);
}

int main() {
  const long n = 256L * 10000000L;
  double* out;
  CK(hipMalloc(&out, n * 8));
  run<256, 2, 1>(out, n); run<256, 1, 1>(out, n); run<64, 2, 1>(out, n); run<64, 1, 1>(out, n);
  run<256, 2, 2>(out, n); run<256, 2, 4>(out, n); run<256, 2, 8>(out, n); run<256, 2, 16>(out, n);
  run<256, 1, 2>(out, n); run<256, 1, 4>(out, n); run<256, 1, 8>(out, n); run<256, 1, 16>(out, n);
  run<64, 1, 2>(out, n); run<64, 1, 4>(out, n); run<64, 1, 8>(out, n); run<64, 1, 16>(out, n);
  run<64, 2, 4>(out, n); run<64, 2, 8>(out, n);
  run<1024, 1, 1>(out, n); run<1024, 2, 1>(out, n); run<1024, 1, 4>(out, n);
  run_nt<64, 1, 16>(out, n); run_nt<256, 2, 1>(out, n); run_nt<256, 1, 16>(out, n);
  CK(hipFree(out));
  return 0;
}
