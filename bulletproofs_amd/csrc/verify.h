// Shared definitions of the batch verifier (verify.hip, host/rangeproof_host.cpp).
#pragma once
#include <stdint.h>

namespace bp {

struct VDims {
  uint32_t B;   // proofs in the batch
  uint32_t n;   // bit width (upper bound 2^n)
  uint32_t m;   // aggregated values (power of two; commitments padded with O)
  uint32_t nm;  // vector length
  uint32_t k;   // IPP rounds = log2(nm)
};

// rows of the per-batch point array pts[row][proof][64]
enum { VP_A = 0, VP_S = 1, VP_T1 = 2, VP_T2 = 3, VP_L0 = 4 };
// then L_0..L_{k-1}, R_0..R_{k-1}, V_0..V_{m-1}
// rows of the per-batch scalar array sc[row][proof][32]: tBlinding, mu, t, l, r
enum { VS_TAU = 0, VS_MU = 1, VS_T = 2, VS_L = 3, VS_R = 4 };

// lanes cooperating on one proof in the fixed-base kernel: enough threads to fill the chip
inline uint32_t vbatch_lpp(uint32_t B) {
  uint32_t lpp = 1;
  while (lpp < 64 && (uint64_t)B * lpp < 131072) lpp <<= 1;
  return lpp;
}

}  // namespace bp
