// Sanitizer harness (CPU only): the host compiler wfk_compile.cpp built with
// -fsanitize=address,undefined behind a C entry point that a ctypes driver can call with
// the same wfk_program structs the product flattener emits.  The digest walks every table
// the compiler produced, so out-of-bounds content would be read (and caught) here too.
#include <cstdint>
#include <string>

#include "wfk.h"
#include "wfk_internal.h"

extern "C" int wfk_san_compile(const wfk_program* prog, const wfk_grid* grid, const double* tlist,
                               int64_t n, double* digest, char* err, int errcap) {
  HostPlan H;
  std::string e;
  const int rc = wfk_compile(prog, grid, tlist, n, H, e);
  if (err && errcap > 0) {
    int i = 0;
    for (; i + 1 < errcap && i < (int)e.size(); ++i) err[i] = e[i];
    err[i] = 0;
  }
  double d = 0.0;
  if (rc == 0) {
    for (const DevChannel& c : H.channels) d += c.offset + c.piece_begin + c.piece_end;
    for (const DevPiece& p : H.pieces) {
      d += (double)p.start + (double)p.stop + p.n_blk;
      // every block of the piece must lie inside params[] and its header must be consistent
      int64_t off = p.par_off;
      int32_t len = p.first_len;
      for (int b = 0; b < p.n_blk; ++b) {
        if (off < 0 || off + len > (int64_t)H.params.size() || (int32_t)H.params[off] != len) return -1000;
        for (int32_t i = 0; i < len; ++i) d += H.params[off + i] == H.params[off + i] ? 1e-9 : 0.0;
        off += len;
        if (b + 1 < p.n_blk) len = (int32_t)H.params[off];
      }
    }
    for (double v : H.pool) d += v == v ? 1e-9 : 0.0;
    for (int32_t v : H.chunk_first) {
      if (v < 0 || v >= (int32_t)H.pieces.size()) return -1001;
      d += v;
    }
    for (const auto& m : H.member_idx)
      for (int64_t v : m) {
        if (v < 0 || v > H.n) return -1002;
        d += (double)v;
      }
  }
  if (digest) *digest = d;
  return rc;
}

// canary: a deliberate heap overflow, so the harness can prove the sanitizer is live
extern "C" int wfk_san_canary(int n) {
  volatile int* p = new int[4];
  const int v = p[n];   // n == 4: one past the end
  delete[] p;
  return v;
}
