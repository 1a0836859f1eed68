// Sanitizer harness (CPU only): the host compiler wfk_compile.cpp built with
// -fsanitize=address,undefined behind a C entry point that a ctypes driver can call with
// the same wfk_program structs the product flattener emits.  The digest walks every table
// the compiler produced, so out-of-bounds content would be read (and caught) here too.
#include <cstdint>
#include <string>
#include <vector>

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
    if (H.shortp) {
      // short tier (WFK_SH_*): every unit, slot and record reference must stay inside its table; the
      // units of a channel, together with the pieces left to the general kernel (mixed plans), must
      // tile [0, n) in order; the slots of a unit its sample range
      int64_t next_j = 0;
      int32_t ch = -1, fp = 0;      // fp: next piece of the channel to look at for a foreign stretch
      auto skip_foreign = [&](int32_t c, int64_t upto) -> bool {   // foreign pieces starting at next_j, up to `upto`
        for (; fp < H.channels[c].piece_end; ++fp) {
          const DevPiece& p = H.pieces[fp];
          if (p.stop <= next_j) continue;
          if (p.n_blk != 0 && !(p.flags & WFK_PF_SHORT) && p.start == next_j && p.stop <= upto) { next_j = p.stop; continue; }
          break;
        }
        return next_j == upto;
      };
      for (const ShortUnit& u : H.s_units) {
        if (u.ch != ch) {
          if (ch >= 0 && !skip_foreign(ch, H.n)) return -1100;
          for (int32_t c = ch + 1; c < u.ch; ++c) {     // channels made of foreign pieces only
            next_j = 0; fp = H.channels[c].piece_begin;
            if (!skip_foreign(c, H.n)) return -1101;
          }
          ch = u.ch; next_j = 0; fp = H.channels[ch].piece_begin;
        }
        if (!skip_foreign(ch, u.j0)) return -1102;
        if (u.n_samples <= 0 || u.n_samples > WFK_SH_LCAP) return -1102;
        next_j += u.n_samples;
        if (u.n_slots < 0 || u.n_slots > 64 || u.slot0 < 0 || (size_t)u.slot0 + (size_t)u.n_slots > H.s_slots.size()) return -1103;
        int32_t covered = 0, prev_end = 0;
        for (int32_t k = 0; k < u.n_slots; ++k) {
          const uint32_t w = H.s_slots[(size_t)u.slot0 + k];
          if (!(w >> 31)) return -1104;
          const int32_t o = (int32_t)((w >> 16) & 0x3ff), len = (int32_t)((w >> 26) & 15) + 1;
          if (o < prev_end || o + len > u.n_samples || len > WFK_SH_R) return -1105;
          prev_end = o + len; covered += len;
          int64_t at = 2 * (u.rec0 + (int64_t)(w & 0xffff));
          for (;;) {                                   // the record's ops, up to the one flagged last
            if (at < 0 || at + WFK_SH_OP1 > (int64_t)H.params.size()) return -1106;
            uint64_t word;
            __builtin_memcpy(&word, &H.params[(size_t)at], sizeof word);
            const int deg = (int)(word & 3);
            const int64_t ref = (int64_t)(word >> 32);
            if (ref > u.j0 + o || u.j0 + o - ref >= WFK_SH_SUB) return -1107;
            const int sz = deg > 1 ? WFK_SH_OP3 : WFK_SH_OP1;
            if (at + sz > (int64_t)H.params.size()) return -1108;
            for (int i = 1; i < sz; ++i) d += H.params[(size_t)at + i] == H.params[(size_t)at + i] ? 1e-9 : 0.0;
            at += sz;
            if (word & WFK_SH_LAST) break;
          }
        }
        if (((u.gaps & 1) == 0) != (covered == u.n_samples) && u.n_slots > 0) return -1109;
        d += (double)u.j0 + u.n_samples + u.rec0;
      }
      if (H.n > 0) {
        if (ch >= 0 && !skip_foreign(ch, H.n)) return -1110;
        for (int32_t c = ch + 1; c < H.n_channels; ++c) {
          next_j = 0; fp = H.channels[c].piece_begin;
          if (!skip_foreign(c, H.n)) return -1110;
        }
      }
    }
    if (H.shortp) {
      // window tables of the fused sampler -> FIR chain at AWG rates (wfk_chain_windows): for K = 1024 and
      // K = 1537 geometries, the entries of every half window are in order, disjoint, inside the half and
      // inside [0, n), reference records inside params[], and cover exactly the evaluated pieces' samples
      for (int geo = 0; geo < 2; ++geo) {
        const int64_t hopb = geo ? 10 : 12, hop = 256 * hopb, half = 128 * (16 + hopb), lead = geo ? 1536 : 1023;
        const int64_t nblk = (H.n + hop - 1) / hop, npairs = (nblk + 1) / 2;
        std::vector<ShortWin> wins;
        std::vector<uint32_t> ents;
        std::string why;
        if (wfk_chain_windows(H, H.n, hop, lead, half, npairs, wins, ents, why) != 0) { d += 1.0; continue; }   // (record span limit: a legal refusal)
        if ((int64_t)wins.size() != npairs * H.n_channels * 2) return -1200;
        for (int32_t c = 0; c < H.n_channels; ++c)
          for (int64_t pr = 0; pr < npairs; ++pr)
            for (int hf = 0; hf < 2; ++hf) {
              const ShortWin& W = wins[((size_t)c * npairs + pr) * 2 + hf];
              const int64_t h0 = 2 * pr * hop - lead + half * hf;
              const int64_t w0 = h0 > 0 ? h0 : 0, w1 = h0 + half < H.n ? h0 + half : H.n;
              if (W.cnt < 0 || W.ccnt < 0 || W.e0 < 0 || W.e0 + W.cnt + W.ccnt > (int64_t)ents.size() || W.rec0 < 0) return -1201;
              if (!H.mixed && W.ccnt != 0) return -1206;
              int64_t covered = 0, prev = w0;
              for (int32_t k = 0; k < W.cnt; ++k) {
                const uint32_t w = ents[(size_t)(W.e0 + k)];
                const int64_t o = (w >> 16) & 0xfff, len = (w >> 28) + 1, j = h0 + o;
                if (j < prev || j + len > w1 || len > WFK_SH_R) return -1202;
                prev = j + len; covered += len;
                const int64_t at = 2 * (W.rec0 + (int64_t)(w & 0xffff));
                if (at < 0 || at + WFK_SH_OP1 > (int64_t)H.params.size()) return -1203;
                uint64_t word;
                __builtin_memcpy(&word, &H.params[(size_t)at], sizeof word);
                const int64_t ref = (int64_t)(word >> 32);
                if (ref > j || j - ref >= WFK_SH_SUB) return -1204;
              }
              int64_t copied = 0;      // runs of the general kernel's pieces (mixed plans): in order, disjoint, inside the half
              prev = w0;
              for (int32_t k = 0; k < W.ccnt; ++k) {
                const uint32_t w = ents[(size_t)(W.e0 + W.cnt + k)];
                const int64_t o = (w >> 16) & 0xfff, len = (w >> 28) + 1, j = h0 + o;
                if (j < prev || j + len > w1 || len > WFK_SH_R) return -1207;
                prev = j + len; copied += len;
              }
              int64_t want = 0, want_copy = 0;   // evaluated / copied samples of the channel inside the half
              for (int32_t q = H.channels[c].piece_begin; q < H.channels[c].piece_end; ++q) {
                const DevPiece& P = H.pieces[q];
                if (P.n_blk == 0) continue;
                const int64_t a = P.start > w0 ? P.start : w0, b = P.stop < w1 ? P.stop : w1;
                if (b > a) ((P.flags & WFK_PF_SHORT) ? want : want_copy) += b - a;
              }
              if (covered != want || copied != want_copy) return -1205;
              d += (double)W.cnt + W.pad;
            }
      }
    }
    for (const DevPiece& p : H.pieces) {
      d += (double)p.start + (double)p.stop + p.n_blk;
      if (p.flags & WFK_PF_SHORT) {
        if (p.par_off < 0 || p.par_off + (int64_t)p.n_blk * p.first_len > (int64_t)H.params.size()) return -1111;
        continue;
      }
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
