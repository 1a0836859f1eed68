// Streaming structures for a single-pass IIR: 64 rows x 1e7 doubles in, the same out.
// A wave owns a chunk of 64 lane blocks x 32 samples; the samples go through an LDS transpose on the
// way in (coalesced load -> lane block) and on the way out, with WORK dependent FMAs per sample in
// between to stand for the two sweeps.  Variants:
//   0  one chunk per workgroup (the shipped iir_onepass structure), chunk = blockIdx
//   1  persistent waves, per-row tickets, no prefetch
//   2  persistent waves, per-row tickets, the next chunk's loads in flight (registers) while the
//      current one is processed from LDS
//   hipcc -O3 --offload-arch=gfx950 tools/iir_stream_probe.hip -o /tmp/isp && /tmp/isp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int LB = 32, CHUNK = 64 * LB;

template <int WORK>
__device__ __forceinline__ double work(double x, double& z) {
#pragma unroll
  for (int w = 0; w < WORK; ++w) z = fma(z, 0.999, x);   // a dependent chain through z, like the recurrence
  return WORK ? z : x;
}

template <int VARIANT, int WORK, int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_stream(const double* __restrict__ in, double* __restrict__ out, long n,
                                                      long nchunks, int rows, unsigned* __restrict__ ticket) {
  __shared__ double tile[64][LB + 1];
  const int lane = threadIdx.x;
  const int row = blockIdx.x % rows;
  const double* xrow = in + (long)row * n;
  double* yrow = out + (long)row * n;
  auto take = [&]() -> long {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(ticket + 16 * row, 1u);
    return (long)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
  };
  long c = VARIANT == 0 ? blockIdx.x / rows : take();
  if (c >= nchunks) return;
  double v[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) v[i] = xrow[c * CHUNK + i * 64 + lane];
  for (;;) {
    long cn = nchunks;
    if (VARIANT != 0) cn = take();
#pragma unroll
    for (int i = 0; i < LB; ++i) { const int j = i * 64 + lane; tile[j / LB][j % LB] = v[i]; }
    __syncthreads();
    if (VARIANT == 2 && cn < nchunks) {
#pragma unroll
      for (int i = 0; i < LB; ++i) v[i] = xrow[cn * CHUNK + i * 64 + lane];
    }
    double z = 0.0;
    // "sweep 1" (state only), then "sweep 2" writing the results in place
#pragma unroll 8
    for (int i = 0; i < LB; ++i) (void)work<WORK>(tile[lane][i], z);
#pragma unroll 8
    for (int i = 0; i < LB; ++i) tile[lane][i] = work<WORK>(tile[lane][i], z);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LB; ++i) { const int j = i * 64 + lane; yrow[c * CHUNK + j] = tile[j / LB][j % LB]; }
    __syncthreads();
    if (VARIANT == 0 || cn >= nchunks) break;
    if (VARIANT == 1) {
#pragma unroll
      for (int i = 0; i < LB; ++i) v[i] = xrow[cn * CHUNK + i * 64 + lane];
    }
    c = cn;
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

template <int VARIANT, int WORK, int WAVES>
static void run(const double* in, double* out, long n, int rows, unsigned* ticket, int waves_per_cu) {
  const long nchunks = n / CHUNK;                      // (whole chunks only: a probe)
  const unsigned grid = VARIANT == 0 ? (unsigned)(nchunks * rows) : (unsigned)(256 * waves_per_cu / rows * rows);
  float ms = timeit([&] {
    if (VARIANT != 0) CK(hipMemsetAsync(ticket, 0, rows * 64, 0));
    hipLaunchKernelGGL((k_stream<VARIANT, WORK, WAVES>), dim3(grid), dim3(64), 0, 0, in, out, n, nchunks, rows, ticket);
  });
  printf("variant %d  work %2d  launch_bounds %d  grid %7u : %6.3f ms  %5.2f TB/s (read + write)\n", VARIANT, WORK, WAVES, grid,
         ms, 2.0 * rows * nchunks * CHUNK * 8 / ms * 1e-9);
}

int main() {
  const int rows = 64;
  const long n = 10000000L / CHUNK * CHUNK;
  double *in, *out;
  unsigned* ticket;
  CK(hipMalloc(&in, rows * n * 8)); CK(hipMalloc(&out, rows * n * 8)); CK(hipMalloc(&ticket, rows * 64));
  CK(hipMemset(in, 0, rows * n * 8));
  run<0, 0, 2>(in, out, n, rows, ticket, 8);
  run<1, 0, 2>(in, out, n, rows, ticket, 8);
  run<2, 0, 2>(in, out, n, rows, ticket, 8);
  run<2, 0, 2>(in, out, n, rows, ticket, 9);
  run<0, 6, 2>(in, out, n, rows, ticket, 8);
  run<1, 6, 2>(in, out, n, rows, ticket, 8);
  run<2, 6, 2>(in, out, n, rows, ticket, 8);
  run<2, 6, 2>(in, out, n, rows, ticket, 9);
  run<0, 12, 2>(in, out, n, rows, ticket, 8);
  run<2, 12, 2>(in, out, n, rows, ticket, 8);
  run<2, 12, 2>(in, out, n, rows, ticket, 9);
  return 0;
}
