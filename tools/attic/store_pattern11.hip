// Eleventh probe: dynamic balancing of the contiguous-eighth walk.  Odd XCDs write ~20% slower
// than even ones (store_pattern10), so equal static eighths leave the even XCDs idle at the
// end.  Here every workgroup CLAIMS its chunk: next chunk of its own XCD's eighth (one atomic
// on a per-XCD counter), and once that is exhausted, the next chunk of another eighth
// (partner XCD first).  Stores only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool DYN>
__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, long n_chunks, unsigned* ctr, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long per = (n_chunks + 7) / 8;
  long c = -1;
  if (!DYN) {
    const long b = blockIdx.x, x = b & 7, k = b >> 3;
    if (k < per) c = x * per + k;
  } else {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
    int cc = -1;
    if (threadIdx.x == 0) {
      for (int i = 0; i < 8 && cc < 0; ++i) {
        const unsigned vx = xcc ^ i;               // own eighth, then the IOD partner, then the rest
        const long size = std::min<long>(per, n_chunks - (long)vx * per);
        if (size <= 0) continue;
        if (i > 0 && __hip_atomic_load(ctr + vx * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)size) continue;
        const unsigned j = atomicAdd(ctr + vx * 32, 1u);
        if (j < (unsigned)size) cc = (int)(vx * per + j);
      }
    }
    c = __builtin_amdgcn_readfirstlane(cc);
  }
  if (c < 0 || c >= n_chunks) return;
  const long t0 = c * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[64 * i] = v + i;
  }
}

__device__ __forceinline__ long claim(unsigned* ctr, unsigned xcc, long per, long n_chunks) {
  int cc = -1;
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8 && cc < 0; ++i) {
      const unsigned vx = xcc ^ i;
      const long size = std::min<long>(per, n_chunks - (long)vx * per);
      if (size <= 0) continue;
      if (i > 0 && __hip_atomic_load(ctr + vx * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)size) continue;
      const unsigned j = atomicAdd(ctr + vx * 32, 1u);
      if (j < (unsigned)size) cc = (int)(vx * per + j);
    }
  }
  return cc;   // valid in lane 0 only
}

// persistent workgroups: claim the NEXT chunk before writing the current one
__global__ void __launch_bounds__(64) k_persist(double* out, long n_tiles, int tpc, long n_chunks, unsigned* ctr, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long per = (n_chunks + 7) / 8;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
  long c = __builtin_amdgcn_readfirstlane((int)claim(ctr, xcc, per, n_chunks));
  while (c >= 0) {
    const long nxt = claim(ctr, xcc, per, n_chunks);     // in flight while we store
    const long t0 = c * tpc;
    for (int tt = 0; tt < tpc; ++tt) {
      const long t = t0 + tt;
      if (t >= n_tiles) break;
      double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 16; ++i) p[64 * i] = v + i;
    }
    c = __builtin_amdgcn_readfirstlane((int)nxt);
  }
}

int main() {
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  const int NB = 4;
  double* bufs[NB];
  for (auto& p : bufs) CK(hipMalloc(&p, n * 8));
  unsigned* ctr; CK(hipMalloc(&ctr, 8 * 32 * 4));
  CK(hipFuncSetAttribute((const void*)k_chunk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_chunk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int tpc : {8, 4, 2, 1}) for (int wpc : {12, 8}) {
    const unsigned lds = (160 * 1024 / wpc) & ~255u;
    const long n_chunks = (n_tiles + tpc - 1) / tpc;
    const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
    for (int dyn = 0; dyn < 3; ++dyn) {
      printf("tpc=%d w/CU=%2d %s:", tpc, wpc, dyn == 2 ? "persist" : dyn ? "claimed" : "static ");
      for (int i = 0; i < NB; ++i) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto launch = [&] {
          if (dyn == 2) { hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0); hipLaunchKernelGGL(k_persist, dim3(256 * wpc), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, ctr, 1.0); }
          else if (dyn) { hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0); hipLaunchKernelGGL(k_chunk<true>, dim3(g), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, ctr, 1.0); }
          else hipLaunchKernelGGL(k_chunk<false>, dim3(g), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, ctr, 1.0);
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf(" %6.3f", ms / 5);
      }
      printf(" ms\n");
    }
  }
  // verify coverage of the claimed walk once: fill with NaN pattern, run, count unwritten
  return 0;
}
