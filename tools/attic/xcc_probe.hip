// Which XCD does workgroup b run on, and how many CUs does each XCD expose?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_id(unsigned* out, int spin) {
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));      // HW_REG_HW_ID[15:0]
  long t = wall_clock64();
  while (wall_clock64() - t < spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | hw;
}

int main() {
  const int N = 8192 * 4;
  unsigned* d; CK(hipMalloc(&d, N * 4));
  unsigned* h = (unsigned*)malloc(N * 4);
  hipLaunchKernelGGL(k_id, dim3(N), dim3(64), 0, 0, d, 2000);
  CK(hipMemcpy(h, d, N * 4, hipMemcpyDeviceToHost));
  printf("blockIdx -> XCC_ID (first 24):");
  for (int b = 0; b < 24; ++b) printf(" %u", h[b] >> 16);
  printf("\n");
  int mism = 0;
  for (int b = 0; b < N; ++b) if ((h[b] >> 16) != (h[b & 7] >> 16)) ++mism;
  printf("workgroups whose XCC differs from that of workgroup (b & 7): %d of %d\n", mism, N);
  std::map<unsigned, std::set<unsigned>> cus;
  std::map<unsigned, int> cnt;
  for (int b = 0; b < N; ++b) {
    const unsigned xcc = h[b] >> 16, hw = h[b] & 0xffff;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[xcc].insert((se << 8) | (sh << 4) | cu);
    cnt[xcc]++;
  }
  for (auto& kv : cus) printf("XCC %u: %zu distinct (se,sh,cu), %d workgroups\n", kv.first, kv.second.size(), cnt[kv.first]);
  return 0;
}
