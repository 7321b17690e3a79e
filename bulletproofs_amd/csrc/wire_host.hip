// Canonical byte format and `show` rendering of a range proof (host only).
//
// The reference has no serialisation: `RangeProof` / `InnerProductProof` only derive Show/Eq/Generic/NFData
// (/root/reference/Bulletproofs/RangeProof/Internal.hs:14-38, InnerProductProof/Internal.hs:10-24).  To compare
// proofs byte for byte across implementations this library fixes one:
//
//   tBlinding | mu | t | aCommit | sCommit | t1Commit | t2Commit | k | L_1..L_k | R_1..R_k | l | r
//
// scalars: 32 bytes big-endian, canonical (< r); points: 0x04 | x | y (32 bytes big-endian each, canonical,
// on the curve) or the single byte 0x00 for `O`; k: one byte, the number of IPP rounds.
// bp_rangeproof_show renders the text GHC's derived Show gives for the record (record syntax, fields in
// declaration order), in either candidate format of the element `show` (transcript.h).
#include <string.h>

#include "engine.h"
#include "transcript.h"

using namespace bp;
using namespace bp::host;

namespace {

bool lt_mod(const uint8_t le[32], const uint64_t m[4]) {
  uint64_t v[4];
  memcpy(v, le, 32);
  for (int i = 3; i >= 0; i--) {
    if (v[i] < m[i]) return true;
    if (v[i] > m[i]) return false;
  }
  return false;
}
void put_be(uint8_t* out, const uint8_t le[32]) {
  for (int i = 0; i < 32; i++) out[i] = le[31 - i];
}
bool is_zero64(const uint8_t* p) {
  for (int i = 0; i < 64; i++)
    if (p[i]) return false;
  return true;
}
size_t put_point(uint8_t* out, const uint8_t pt[64]) {
  if (is_zero64(pt)) {
    out[0] = 0x00;
    return 1;
  }
  out[0] = 0x04;
  put_be(out + 1, pt);
  put_be(out + 33, pt + 32);
  return 65;
}

}  // namespace

extern "C" {

size_t bp_rangeproof_wire_size(uint32_t k) { return 3 * 32 + 4 * 65 + 1 + (size_t)2 * k * 65 + 2 * 32; }

// scalars: 5 x 32 B little-endian (tBlinding, mu, t, l, r); points: (4 + 2k) x 64 B (A, S, T1, T2, L.., R..)
int bp_rangeproof_serialize(const uint8_t* scalars, const uint8_t* points, uint32_t k, uint8_t* out, size_t cap,
                            size_t* len) {
  if (!scalars || !points || !out || !len || k > 255) return BP_ERR_ARG;
  if (cap < bp_rangeproof_wire_size(k)) return BP_ERR_ARG;
  for (int i = 0; i < 5; i++)
    if (!lt_mod(scalars + 32 * i, MOD_R)) return BP_ERR_ARG;
  for (uint32_t i = 0; i < 4 + 2 * k; i++) {
    const uint8_t* p = points + (size_t)i * 64;
    if (is_zero64(p)) continue;
    if (!lt_mod(p, MOD_P) || !lt_mod(p + 32, MOD_P)) return BP_ERR_ARG;
  }
  size_t o = 0;
  for (int i = 0; i < 3; i++, o += 32) put_be(out + o, scalars + 32 * i);
  for (int i = 0; i < 4; i++) o += put_point(out + o, points + (size_t)i * 64);
  out[o++] = (uint8_t)k;
  for (uint32_t i = 0; i < 2 * k; i++) o += put_point(out + o, points + (size_t)(4 + i) * 64);
  put_be(out + o, scalars + 96);
  o += 32;
  put_be(out + o, scalars + 128);
  o += 32;
  *len = o;
  return BP_OK;
}

// Strict inverse: rejects non-canonical scalars / coordinates, points off the curve, unknown tags,
// truncated input and trailing bytes.  points_cap: room in `points`, in points.
int bp_rangeproof_deserialize(const uint8_t* in, size_t len, uint32_t* k_out, uint8_t* scalars, uint8_t* points,
                              size_t points_cap) {
  if (!in || !k_out || !scalars || !points) return BP_ERR_ARG;
  size_t o = 0;
  auto get_scalar = [&](uint8_t* dst) {
    if (len - o < 32) return false;
    uint8_t le[32];
    for (int i = 0; i < 32; i++) le[i] = in[o + 31 - i];
    o += 32;
    if (!lt_mod(le, MOD_R)) return false;
    memcpy(dst, le, 32);
    return true;
  };
  auto get_point = [&](uint8_t* dst) {
    if (len - o < 1) return false;
    uint8_t tag = in[o++];
    if (tag == 0x00) {
      memset(dst, 0, 64);
      return true;
    }
    if (tag != 0x04 || len - o < 64) return false;
    affine a;
    uint8_t le[64];
    for (int i = 0; i < 32; i++) {
      le[i] = in[o + 31 - i];
      le[32 + i] = in[o + 63 - i];
    }
    o += 64;
    if (!lt_mod(le, MOD_P) || !lt_mod(le + 32, MOD_P)) return false;
    memcpy(&a, le, 64);
    if (affine_is_inf(a) || !affine_on_curve(a)) return false;
    memcpy(dst, le, 64);
    return true;
  };
  for (int i = 0; i < 3; i++)
    if (!get_scalar(scalars + 32 * i)) return BP_ERR_ARG;
  if (points_cap < 4) return BP_ERR_ARG;
  for (int i = 0; i < 4; i++)
    if (!get_point(points + (size_t)i * 64)) return BP_ERR_ARG;
  if (len - o < 1) return BP_ERR_ARG;
  uint32_t k = in[o++];
  if (points_cap < 4 + (size_t)2 * k) return BP_ERR_ARG;
  for (uint32_t i = 0; i < 2 * k; i++)
    if (!get_point(points + (size_t)(4 + i) * 64)) return BP_ERR_ARG;
  if (!get_scalar(scalars + 96) || !get_scalar(scalars + 128)) return BP_ERR_ARG;
  if (o != len) return BP_ERR_ARG;
  *k_out = k;
  return BP_OK;
}

// `show (proof :: RangeProof Fr PA)` as GHC derives it.  style: 0 / 1 as bp_ctx_set_show_style.
int bp_rangeproof_show(const uint8_t* scalars, const uint8_t* points, uint32_t k, int style, char* out, size_t cap,
                       size_t* len) {
  if (!scalars || !points || !out || !len || style < 0 || style > 1) return BP_ERR_ARG;
  // every number has at most 78 digits; a point at most 2 * 78 + 12 characters
  size_t need = 256 + 5 * 96 + (size_t)(4 + 2 * k) * 192;
  if (cap < need) return BP_ERR_ARG;
  ShowStyle st = (ShowStyle)style;
  char* o = out;
  auto lit = [&](const char* s) {
    size_t n = strlen(s);
    memcpy(o, s, n);
    o += n;
  };
  lit("RangeProof {tBlinding = ");
  o += show_fr(st, scalars, o);
  lit(", mu = ");
  o += show_fr(st, scalars + 32, o);
  lit(", t = ");
  o += show_fr(st, scalars + 64, o);
  const char* names[4] = {", aCommit = ", ", sCommit = ", ", t1Commit = ", ", t2Commit = "};
  for (int i = 0; i < 4; i++) {
    lit(names[i]);
    o += show_point(st, points + (size_t)i * 64, o);
  }
  lit(", productProof = InnerProductProof {lCommits = [");
  for (uint32_t i = 0; i < k; i++) {
    if (i) lit(",");
    o += show_point(st, points + (size_t)(4 + i) * 64, o);
  }
  lit("], rCommits = [");
  for (uint32_t i = 0; i < k; i++) {
    if (i) lit(",");
    o += show_point(st, points + (size_t)(4 + k + i) * 64, o);
  }
  lit("], l = ");
  o += show_fr(st, scalars + 96, o);
  lit(", r = ");
  o += show_fr(st, scalars + 128, o);
  lit("}}");
  *len = (size_t)(o - out);
  return BP_OK;
}

}  // extern "C"
