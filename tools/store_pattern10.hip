// Tenth probe: per-XCD finish times of the contiguous-eighth walk, per buffer.  If a "slow"
// buffer is one where ONE eighth is slow (static partition -> load imbalance), it shows here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, long n_chunks,
                                              unsigned long long* tmin, unsigned long long* tmax, int rot, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long b = blockIdx.x, x = b & 7, k = b >> 3;
  const long per = (n_chunks + 7) / 8;
  if (k >= per) return;
  const long c = ((x + rot) & 7) * per + k;
  if (c >= n_chunks) return;
  if (threadIdx.x == 0 && k == 0) atomicMin(tmin + x, (unsigned long long)wall_clock64());
  const long t0 = c * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[64 * i] = v + i;
  }
  __builtin_amdgcn_s_waitcnt(0);
  if (threadIdx.x == 0 && k >= per - 400) atomicMax(tmax + x, (unsigned long long)wall_clock64());
}

int main() {
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  const int NB = 6;
  double* bufs[NB];
  for (auto& p : bufs) CK(hipMalloc(&p, n * 8));
  unsigned long long *tmin, *tmax, h0[8], h1[8];
  CK(hipMalloc(&tmin, 64)); CK(hipMalloc(&tmax, 64));
  CK(hipFuncSetAttribute((const void*)k_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int wpc = 12, tpc = 8;
  const unsigned lds = (160 * 1024 / wpc) & ~255u;
  const long n_chunks = (n_tiles + tpc - 1) / tpc;
  const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
  for (int i = 0; i < NB; ++i) for (int rot = 0; rot < (i < 2 ? 3 : 1); ++rot) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(tmin, 0xff, 64)); CK(hipMemset(tmax, 0, 64));
      hipLaunchKernelGGL(k_chunk, dim3(g), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, tmin, tmax, rot, 1.0);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h0, tmin, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, tmax, 64, hipMemcpyDeviceToHost));
    unsigned long long t0 = h0[0];
    for (int x = 1; x < 8; ++x) t0 = h0[x] < t0 ? h0[x] : t0;
    printf("buf%d region=(xcd+%d)%%8  per-XCD finish ms:", i, rot);
    for (int x = 0; x < 8; ++x) printf(" %6.3f", (h1[x] - t0) * 1e-5);
    printf("\n");
  }
  return 0;
}
