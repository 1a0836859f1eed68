/* Plain-C consumer of include/wfk.h: proves the boundary needs nothing but C.
 * Builds the flattened program of  0.5 * gaussian(sigma) * cos(w (t - s))  on [-1, 1),
 * creates a grid plan (host-only when no GPU is visible), checks the integer piece
 * indices against np.searchsorted semantics, and -- if a device is present -- samples and
 * compares with libm.  Exit code 0 = ok.  (tests/test_abi_cpu.py compiles and runs it.) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wfk.h"

int main(void) {
  const int32_t ch_member_off[] = {0, 1}, mb_piece_off[] = {0, 3};
  const double ch_offset[] = {0}, ch_tshift[] = {0}, lo[] = {-INFINITY}, hi[] = {INFINITY};
  const double pc_bound[] = {-1.0, 1.0, INFINITY};
  const int32_t pc_term_off[] = {0, 0, 1, 1};
  const double amp_re[] = {0.5}, amp_im[] = {0.0};
  const int32_t tm_factor_off[] = {0, 2};
  const int32_t fc_type[] = {WFK_GAUSSIAN, WFK_COS};
  const double fc_power[] = {1, 1}, fc_shift[] = {0.0, 0.25};
  const int64_t fc_arg_off[] = {0, 1, 2};
  const double pool[] = {0.6, 7.0};
  wfk_program P = {1, 1, 3, 1, 2, 2, ch_member_off, ch_offset, ch_tshift, lo, hi, mb_piece_off,
                   pc_bound, pc_term_off, amp_re, amp_im, tm_factor_off, fc_type, fc_power,
                   fc_shift, fc_arg_off, pool};
  const int64_t n = 1001;
  wfk_grid g = {-2.0, 4.0 / 1000, n, 1, 2.0};   /* np.linspace(-2, 2, 1001) */
  wfk_plan* plan = NULL;
  if (wfk_abi_version() != WFK_ABI_VERSION) return 10;
  if (wfk_plan_create_grid(&P, &g, &plan) != WFK_OK) {
    fprintf(stderr, "create: %s\n", wfk_last_error());
    return 11;
  }
  int64_t idx[3];
  if (wfk_plan_member_index(plan, 0, idx, 3) != 3) return 12;
  /* t[250] = -1.0 exactly, t[750] = 1.0 exactly: side='left' puts both in the LATER piece */
  if (idx[0] != 250 || idx[1] != 750 || idx[2] != n) {
    fprintf(stderr, "idx %lld %lld %lld\n", (long long)idx[0], (long long)idx[1], (long long)idx[2]);
    return 13;
  }
  wfk_plan_info info;
  if (wfk_plan_get_info(plan, &info) != WFK_OK || info.n != n || info.n_channels != 1) return 14;
  int ndev = 0;
  wfk_device_count(&ndev);
  if (ndev > 0) {
    double* y = (double*)malloc(sizeof(double) * n);
    if (wfk_plan_run_host(plan, y, n, WFK_OUT_F64) != WFK_OK) {
      fprintf(stderr, "run: %s\n", wfk_last_error());
      return 15;
    }
    for (int64_t i = 0; i < n; ++i) {
      double t = i == n - 1 ? 2.0 : (double)i * g.step + g.t0;
      double want = (i >= 250 && i < 750) ? 0.5 * exp(-(t / 0.6) * (t / 0.6)) * cos(7.0 * (t - 0.25)) : 0.0;
      if (fabs(y[i] - want) > 1e-12) {
        fprintf(stderr, "sample %lld: %g vs %g\n", (long long)i, y[i], want);
        return 16;
      }
    }
    free(y);
    printf("abi_smoke: sampled on the device, parity ok\n");
  } else {
    if (wfk_plan_run_host(plan, idx, n, WFK_OUT_F64) == WFK_OK) return 17; /* must fail loudly */
    printf("abi_smoke: host-only plan ok (%s)\n", wfk_last_error());
  }
  wfk_plan_destroy(plan);
  return 0;
}
