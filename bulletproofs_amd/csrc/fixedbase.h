// Fixed-base window tables for the generator set of the reference
// (`gs`, `hs`, `h`, `gen`; /root/reference/Bulletproofs/Utils.hs:14-25).
//
// For every base B and every WB-bit window w the table holds d * 2^(WB*w) * B for
// d = 1..2^(WB-1) as affine points, so k*B costs one mixed add per window of the
// signed-digit form of k (digits in [-(2^(WB-1) - 1), 2^(WB-1)], least significant
// window first; the top window also absorbs the final carry).
//
// The window width is a property of the context, chosen when the generators are
// installed: HBM is large (288 GB), so small generator sets get 16-bit windows
// (17 mixed adds per term, 2 MiB of table per window and base) and large sets fall
// back to 12- or 8-bit windows.  WB = 8: 33 windows x 128 entries; WB = 12:
// 22 x 2048; WB = 16: 17 x 32768; WB = 20: 13 x 524288 (32 MiB per window and base;
// only taken when the table budget is raised, BP_FB_TABLE_GIB).
#pragma once
#include "ec.h"

namespace bp {

BP_HD constexpr int fb_windows(int wb) { return 256 / wb + 1; }
BP_HD constexpr int fb_entries(int wb) { return 1 << (wb - 1); }  // magnitudes 1..2^(wb-1)
// digits are packed into u32 words: 4 bytes per word for wb <= 8, 2 halfwords for wb <= 16, else one word each
BP_HD constexpr int fb_slot_bits(int wb) { return wb <= 8 ? 8 : (wb <= 16 ? 16 : 32); }
BP_HD constexpr int fb_dwords(int wb) { return (fb_windows(wb) * fb_slot_bits(wb) + 31) / 32; }
constexpr int FB_MAX_DWORDS = 13;  // wb = 20: 13 words
constexpr int FB_BUILD_BLOCK = 256;  // table entries produced per thread of the build kernel

// base numbering inside the table for a context with `ng` generator pairs
BP_HD uint32_t fb_base_g(uint32_t i) { return i; }
BP_HD uint32_t fb_base_h(uint32_t ng, uint32_t i) { return ng + i; }
BP_HD uint32_t fb_base_G(uint32_t ng) { return 2 * ng; }      // curve generator `gen`
BP_HD uint32_t fb_base_H(uint32_t ng) { return 2 * ng + 1; }  // `h`
// GS_k = sum_{i < 2^k} gs_i  (RangeProof/Internal.hs:200), k = 0..log2(ng)
BP_HD uint32_t fb_base_gsum(uint32_t ng, uint32_t k) { return 2 * ng + 2 + k; }

// bits [lo, lo + WB) of k, zero beyond bit 255 (compile-time lo: static limb indices only)
template <int WB>
BP_HD uint32_t fb_window_bits(const u256& k, int lo) {
  if (lo >= 256) return 0;
  int limb = lo >> 5, sh = lo & 31;
  uint32_t v = k.v[limb] >> sh;
  if (sh + WB > 32 && limb + 1 < 8) v |= k.v[limb + 1] << (32 - sh);
  return v & ((1u << WB) - 1u);
}

// Signed WB-bit digits of a canonical scalar, one slot per window: slot = digit mod 2^slot_bits
// (two's complement), where 2^(WB-1) means +2^(WB-1) (the negative of it is never produced).
// Window w sits in slot (w % per_word) of word (w / per_word).
template <int WB>
BP_HD void fb_recode(const u256& k, uint32_t (&out)[fb_dwords(WB)]) {
  constexpr int SLOT = fb_slot_bits(WB), PER = 32 / SLOT, W = fb_windows(WB);
  constexpr uint32_t HALF = 1u << (WB - 1), FULL = 1u << WB;
  constexpr uint32_t SMASK = SLOT == 32 ? 0xFFFFFFFFu : (uint32_t)((1ull << SLOT) - 1ull);
  uint32_t carry = 0;
#pragma unroll
  for (int g = 0; g < fb_dwords(WB); g++) {
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < PER; b++) {
      int w = PER * g + b;
      if (w < W) {
        uint32_t v = carry + fb_window_bits<WB>(k, w * WB);
        if (v > HALF) {
          carry = 1;
          v = (v - FULL) & SMASK;  // negative digit (v = 2^WB -> 0)
        } else {
          carry = 0;
        }
        word |= v << (SLOT * b);
      }
    }
    out[g] = word;
  }
}

// decode one slot: magnitude (0 = skip) and sign
template <int WB>
BP_HD uint32_t fb_decode(uint32_t code, bool& neg) {
  constexpr int SLOT = fb_slot_bits(WB);
  constexpr uint32_t HALF = 1u << (WB - 1);
  // negative digits are stored as 2^SLOT - mag with mag < HALF, i.e. code > 2^SLOT - HALF
  neg = code > HALF;
  constexpr uint32_t SMASK = SLOT == 32 ? 0xFFFFFFFFu : (uint32_t)((1ull << SLOT) - 1ull);
  return neg ? ((0u - code) & SMASK) : code;
}

}  // namespace bp
