// The Fiat-Shamir transcript of the range-proof verifier on the device: the `show` text of points and scalars,
// SHA-256, and the digest as a scalar (`oracle = fromInteger . integerDigest . sha256`,
// /root/reference/Bulletproofs/Utils.hs:29-33,145-174), one lane per proof.  The bytes hashed are exactly the ones
// the host path (transcript.h) hashes; with the challenges formed where the running commitment already lives, a
// pass of the batch verifier needs no host round trip between its kernels (verify.hip, vbatch_run_fsdev_async).
//
// A message is assembled in a buffer of 32-bit words (bytes in memory order, `stride` words between consecutive
// words of one lane: lanes interleave in global memory so that a wave's loads and stores coalesce) and hashed from
// there.  Every transcript opens with `show _r`, whose first 64 characters are one SHA block: the buffer starts
// behind them and the hash starts from that block's state.
#pragma once
#include "dec.h"
#include "fe.h"

namespace bp {

constexpr uint32_t FS_MSG_WORDS = 320;  // per proof: shamirX's message (4 points, 2 scalars, derived style: 842 bytes +
                                        // padding, below 232 words) and, from word FS_U_WORD on, shamirU's (254 bytes)
constexpr uint32_t FS_U_WORD = 232;
// a hashed opening of a message, to be continued later: state (8 words), the bytes behind its last full block (16
// words, zero beyond the text) and their count | full blocks << 8
constexpr uint32_t FS_PRE_WORDS = 25;

BP_HD uint32_t fs_rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}
BP_HD uint32_t fs_bswap(uint32_t x) { return __builtin_bswap32(x); }

BP_HD uint32_t fs_sha_k(int i) {
  constexpr uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
      0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
      0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
      0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
      0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
      0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
      0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
      0xc67178f2};
  return K[i];
}

// one block: w holds the sixteen big-endian message words and is used as the rolling schedule
BP_HD void fs_sha256_compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const uint32_t s0 = fs_rotr(w15, 7) ^ fs_rotr(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = fs_rotr(w2, 17) ^ fs_rotr(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const uint32_t S1 = fs_rotr(e, 6) ^ fs_rotr(e, 11) ^ fs_rotr(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = h + S1 + ch + fs_sha_k(i) + w[i & 15];
    const uint32_t S0 = fs_rotr(a, 2) ^ fs_rotr(a, 13) ^ fs_rotr(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    h = g;
    g = f;
    f = e;
    e = d + t1;
    d = c;
    c = b;
    b = a;
    a = t1 + t2;
  }
  st[0] += a;
  st[1] += b;
  st[2] += c;
  st[3] += d;
  st[4] += e;
  st[5] += f;
  st[6] += g;
  st[7] += h;
}

// `show _r` = "115792089237316195423570985008687907852837564279074904382605163141518161494337" (78 characters):
// the SHA-256 state after its first 64, and the 14 that open every message buffer
constexpr uint32_t FS_R_ABSORBED = 64;
BP_HD void fs_state_after_show_r_block(uint32_t (&st)[8]) {
  constexpr uint32_t MID[8] = {0x3439cba5u, 0xffbd704du, 0xfb571e4cu, 0x2d9ad191u,
                               0x80e5222du, 0xef40b310u, 0xabe0a86eu, 0x2858e6c8u};
#pragma unroll
  for (int i = 0; i < 8; i++) st[i] = MID[i];
}

struct FsMsg {
  uint32_t* w;      // this lane's word 0
  size_t stride;    // words between consecutive words of this lane
  uint32_t len;     // bytes written
  uint32_t cur;     // the bytes of the word being filled
  uint32_t base;    // bytes of the message hashed before word 0 (behind the first block of `show _r`)
};
struct FsMark {  // a position to come back to (the messages of y, z, x share their opening text)
  uint32_t len, cur;
};

BP_HD void fs_put(FsMsg& m, uint32_t byte) {
  m.cur |= byte << (8 * (m.len & 3));
  m.len++;
  if ((m.len & 3) == 0) {
    m.w[(size_t)(m.len / 4 - 1) * m.stride] = m.cur;
    m.cur = 0;
  }
}
BP_HD FsMark fs_mark(const FsMsg& m) { return FsMark{m.len, m.cur}; }
BP_HD void fs_rewind(FsMsg& m, const FsMark& k) {
  m.len = k.len;
  m.cur = k.cur;
}

// the message opens with the 14 characters of `show _r` behind its first block: "41518161494337"
BP_HD FsMsg fs_open(uint32_t* lane_word0, size_t stride) {
  FsMsg m{lane_word0, stride, 0, 0, 0};
  constexpr char T[15] = "41518161494337";
#pragma unroll
  for (int i = 0; i < 14; i++) fs_put(m, (uint32_t)T[i]);
  return m;
}

// decimal digits of v, no leading zeros (zero is "0")
BP_HD void fs_put_dec(FsMsg& m, const u256& v) {
  uint32_t rec[DEC_REC_WORDS];
  u256_to_dec_record(v, rec);
  const uint32_t n = rec[0] & 0xFFu, first = (uint32_t)DEC_REC - n;
#pragma unroll
  for (int pos = 2; pos < DEC_REC; pos++) {
    if ((uint32_t)pos >= first) fs_put(m, (rec[pos >> 2] >> (8 * (pos & 3))) & 0xFFu);
  }
}

// `show (s :: Fr)`: "P 123" (derived Show, style 0) or "123" (style 1)
BP_HD void fs_put_fr(FsMsg& m, int style, const u256& s) {
  if (style == 0) {
    fs_put(m, 'P');
    fs_put(m, ' ');
  }
  fs_put_dec(m, s);
}

// `pointToBS = show`: "O", "A (P x) (P y)" (style 0) or "A x y" (style 1); (0, 0) encodes O
BP_HD void fs_put_point(FsMsg& m, int style, const u256& x, const u256& y) {
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) any |= x.v[i] | y.v[i];
  if (any == 0) {
    fs_put(m, 'O');
    return;
  }
  fs_put(m, 'A');
  fs_put(m, ' ');
  if (style == 0) {
    fs_put(m, '(');
    fs_put(m, 'P');
    fs_put(m, ' ');
  }
  fs_put_dec(m, x);
  if (style == 0) fs_put(m, ')');
  fs_put(m, ' ');
  if (style == 0) {
    fs_put(m, '(');
    fs_put(m, 'P');
    fs_put(m, ' ');
  }
  fs_put_dec(m, y);
  if (style == 0) fs_put(m, ')');
}

// Hashes the full 64-byte blocks written so far into st (no padding) and returns their number.
BP_HD uint32_t fs_absorb_full_blocks(const FsMsg& m, uint32_t (&st)[8]) {
  const uint32_t nblk = m.len / 64;
#pragma unroll 1
  for (uint32_t b = 0; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = fs_bswap(m.w[(size_t)(b * 16 + i) * m.stride]);
    fs_sha256_compress(st, w);
  }
  return nblk;
}

// Pads the message in place (what lies behind its end is overwritten), hashes blocks first_blk.. of the buffer on top
// of st0 (the state behind everything before them) and returns the digest as a scalar: big-endian integer mod r
// (2^256 < 2 r, one subtraction).  The message itself stays: a caller may rewind to a mark and go on writing.
BP_HD u256 fs_finish_fr_from(FsMsg m, const uint32_t (&st0)[8], uint32_t first_blk) {
  const uint64_t bits = (uint64_t)(m.base + m.len + FS_R_ABSORBED) * 8u;
  fs_put(m, 0x80u);
  while (m.len & 3u) fs_put(m, 0u);
  while ((m.len & 63u) != 56u) {
    m.w[(size_t)(m.len / 4) * m.stride] = 0u;
    m.len += 4;
  }
  m.w[(size_t)(m.len / 4) * m.stride] = fs_bswap((uint32_t)(bits >> 32));
  m.w[(size_t)(m.len / 4 + 1) * m.stride] = fs_bswap((uint32_t)bits);
  m.len += 8;
  const uint32_t nblk = m.len / 64;
  uint32_t st[8];
#pragma unroll
  for (int i = 0; i < 8; i++) st[i] = st0[i];
#pragma unroll 1
  for (uint32_t b = first_blk; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = fs_bswap(m.w[(size_t)(b * 16 + i) * m.stride]);
    fs_sha256_compress(st, w);
  }
  u256 dgst;
#pragma unroll
  for (int i = 0; i < 8; i++) dgst.v[i] = st[7 - i];
  return Fr::reduce_once(dgst);
}
BP_HD u256 fs_finish_fr(const FsMsg& m) {
  uint32_t st[8];
  fs_state_after_show_r_block(st);
  return fs_finish_fr_from(m, st, 0);
}

// Saves the hashed opening of m (FS_PRE_WORDS words, `pstride` apart): the state behind its full blocks, the bytes
// behind them, their count and the number of blocks.
BP_HD void fs_save_opening(const FsMsg& m, uint32_t* pre, size_t pstride) {
  uint32_t st[8];
  fs_state_after_show_r_block(st);
  const uint32_t nblk = fs_absorb_full_blocks(m, st);
#pragma unroll
  for (int i = 0; i < 8; i++) pre[(size_t)i * pstride] = st[i];
  const uint32_t left = m.len - 64 * nblk;
#pragma unroll 1
  for (uint32_t i = 0; i < 16; i++) {
    uint32_t word = 0;
    if (4 * i + 4 <= left) word = m.w[(size_t)(nblk * 16 + i) * m.stride];
    else if (4 * i < left) word = m.cur;
    pre[(size_t)(8 + i) * pstride] = word;
  }
  pre[(size_t)24 * pstride] = left | (nblk << 8);
}
// A message that continues a saved opening: st receives the state to finish from
BP_HD FsMsg fs_resume(const uint32_t* pre, size_t pstride, uint32_t* lane_word0, size_t stride, uint32_t (&st)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) st[i] = pre[(size_t)i * pstride];
  const uint32_t info = pre[(size_t)24 * pstride], left = info & 0xFFu, nblk = info >> 8;
  FsMsg m{lane_word0, stride, left, 0, 64 * nblk};
#pragma unroll 1
  for (uint32_t i = 0; i < 16; i++) {
    const uint32_t word = pre[(size_t)(8 + i) * pstride];
    if (4 * i + 4 <= left) m.w[(size_t)i * stride] = word;
    else if (4 * i < left) m.cur = word;
  }
  return m;
}

// ---- the verifier's challenges --------------------------------------------------------------------------------------
struct FsPoint {
  u256 x, y;
};
// y, z, x of MultiRangeProof/Verifier.hs:31-33: shamirY = oracle (show _r <> A <> S), shamirZ = ... <> show y,
// shamirX = show _r <> A <> S <> T1 <> T2 <> show y <> show z  (Utils.hs:145-165).  The three messages share
// show _r <> A <> S: its full blocks are hashed once.
BP_HD void fs_challenges_yzx(int style, const FsPoint& A, const FsPoint& S, const FsPoint& T1, const FsPoint& T2,
                             uint32_t* lane_word0, size_t stride, u256& y, u256& z, u256& x) {
  FsMsg m = fs_open(lane_word0, stride);
  fs_put_point(m, style, A.x, A.y);
  fs_put_point(m, style, S.x, S.y);
  const FsMark as = fs_mark(m);
  uint32_t st[8];
  fs_state_after_show_r_block(st);
  const uint32_t nb = fs_absorb_full_blocks(m, st);
  y = fs_finish_fr_from(m, st, nb);
  fs_rewind(m, as);
  fs_put_fr(m, style, y);
  z = fs_finish_fr_from(m, st, nb);
  fs_rewind(m, as);
  fs_put_point(m, style, T1.x, T1.y);
  fs_put_point(m, style, T2.x, T2.y);
  fs_put_fr(m, style, y);
  fs_put_fr(m, style, z);
  x = fs_finish_fr_from(m, st, nb);
}
// uChallenge = shamirU tBlinding mu t = oracle (show _r <> show tBlinding <> show mu <> show t)  (Utils.hs:172-174)
BP_HD u256 fs_challenge_u(int style, const u256& tau, const u256& mu, const u256& t, uint32_t* lane_word0, size_t stride) {
  FsMsg m = fs_open(lane_word0, stride);
  fs_put_fr(m, style, tau);
  fs_put_fr(m, style, mu);
  fs_put_fr(m, style, t);
  return fs_finish_fr(m);
}
// shamirX' acc L R = oracle (show _r <> L <> R <> acc)  (Utils.hs:167-170) in two parts: the opening show _r <> L <> R is
// known with the proof and hashed ahead of time; the round itself appends the running commitment.
BP_HD void fs_round_opening(int style, const FsPoint& L, const FsPoint& R, uint32_t* lane_word0, size_t stride,
                            uint32_t* pre, size_t pstride) {
  FsMsg m = fs_open(lane_word0, stride);
  fs_put_point(m, style, L.x, L.y);
  fs_put_point(m, style, R.x, R.y);
  fs_save_opening(m, pre, pstride);
}
BP_HD u256 fs_round_challenge(int style, const uint32_t* pre, size_t pstride, const FsPoint& acc, uint32_t* lane_word0,
                              size_t stride) {
  uint32_t st[8];
  FsMsg m = fs_resume(pre, pstride, lane_word0, stride, st);
  fs_put_point(m, style, acc.x, acc.y);
  return fs_finish_fr_from(m, st, 0);
}

}  // namespace bp
