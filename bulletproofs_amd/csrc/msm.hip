// Pippenger bucket MSM for secp256k1 on gfx950.
//
// Replaces `sumExps` (/root/reference/Bulletproofs/Utils.hs:76-80).
//
// Pipeline (all on the context's stream, no host round trip):
//   1 k_recode        signed c-bit digits per scalar + per-(window,bucket) histogram
//   2 k_scan          exclusive scan of the histogram per window -> bucket starts
//   3 k_scatter       counting-sort scatter: (point index | sign) grouped by bucket
//   4 k_cls_*         counting sort of BUCKETS by population, descending, so that the
//                     64 lanes of a wave own equally long lists (no divergence tail)
//   5 k_accumulate    one lane per bucket (population <= LIGHT_MAX): XYZZ mixed adds
//     k_heavy_*       buckets above LIGHT_MAX are cut into slices, one workgroup per
//                     slice (wave-level shuffle reduction + LDS across waves)
//   6 k_seg_reduce    running-sum reduction of 32-bucket segments, one lane each
//   7 k_window_sum    workgroup tree reduction of the segment values of one window
//   8 k_final         Horner over the windows + affine normalisation
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "engine.h"
#include "devio.h"
#include "fq29.h"
#include "glv.h"
#include "lanes29.h"

namespace bp {

constexpr int LIGHT_MAX = 254;       // buckets with more entries always take the heavy path
constexpr int NBINS = LIGHT_MAX + 2; // bin 0 = heavy, bin k = population LIGHT_MAX+1-k
// When the non-empty buckets do not even fill the chip's lanes, a light lane's chain IS the kernel's duration (65,536
// 64-bit scalars with 14-bit windows: 255 buckets of ~256 entries each walked by one lane, 0.9 ms, beside 32,768 buckets
// of 8).  k_cls_scan then lowers the heavy threshold (bins[2 NBINS + 1]) until at most HEAVY_EXTRA more buckets are
// heavy: their entries are spread over a workgroup each.  A launch that fills the chip keeps LIGHT_MAX: the light
// path does no tree reduction and is the cheaper one per entry.
constexpr uint32_t LIGHT_FILL = 256u * 4u * 64u * 2u;  // lanes of the chip at two waves per SIMD
constexpr uint32_t HEAVY_EXTRA = 512;                  // two slice workgroups per CU
constexpr uint32_t LIGHT_MIN = 24;                     // never send shorter chains to the heavy path
constexpr int SLICE = 2048;          // entries per heavy slice (one 256-thread workgroup)

// ---- 1. digits + histogram --------------------------------------------------
__global__ void __launch_bounds__(256) k_recode(const uint8_t* __restrict__ scalars, uint32_t n, MsmPlan pl,
                                               uint32_t* __restrict__ codes, uint32_t* __restrict__ ranks,
                                               uint32_t* __restrict__ hist) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 s = ld_u256(scalars + (size_t)i * 32);
  uint32_t carry = 0;
  for (int w = 0; w < pl.W; w++) {
    uint32_t code = recode_next(s, pl.c, carry);
    codes[(size_t)w * n + i] = code;
    // the histogram increment also hands this entry its rank inside the bucket, so the scatter pass
    // needs no second round of atomics
    if (code != MSM_SKIP) ranks[(size_t)w * n + i] = atomicAdd(&hist[(size_t)w * pl.nb + (code & ~MSM_NEG)], 1u);
  }
}

// GLV form (glv.h): scalar i becomes the two 129-bit half-scalars k1, k2 with k = k1 + k2*lambda; entry 2i is
// k1 on P_i, entry 2i + 1 is k2 on phi(P_i) = (beta x, y).  Half the windows, hence half the buckets to
// reduce and half the doublings in the final Horner chain; a half-scalar's sign flips the entry's sign flag.
// Scalars [i0, i1) of n (a chunk: the host-scalar entry point recodes chunk by chunk behind the upload).
__global__ void __launch_bounds__(256) k_recode_glv(const uint8_t* __restrict__ scalars, uint32_t n, uint32_t i0, uint32_t i1,
                                                   MsmPlan pl, uint32_t* __restrict__ codes, uint32_t* __restrict__ ranks,
                                                   uint32_t* __restrict__ hist) {
  uint32_t i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i1) return;
  u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)i * 32));
  GlvHalf h[2];
  glv_split(k, h[0], h[1]);
  const size_t ne = (size_t)2 * n;
  const int top_bits = GLV_BITS + 1 - pl.c * (pl.W - 1);
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    u256 s = u256_zero();
#pragma unroll
    for (int q = 0; q < 5; q++) s.v[q] = h[hh].m[q];
    const uint32_t flip = h[hh].neg ? MSM_NEG : 0u;
    const size_t j = (size_t)2 * i + hh;
    uint32_t carry = 0;
    for (int w = 0; w < pl.W; w++) {
      uint32_t code = recode_next(s, pl.c, carry);
      if (code != MSM_SKIP) code ^= flip;
      codes[(size_t)w * ne + j] = code;
      if (w == pl.W - 1 && top_bits <= 5) break;  // handled below, wave-aggregated
      // Structured scalars pile onto bucket 0 (digit +-1): {0,1} scalars in window 0, the carry digit behind the top of
      // 64-bit scalars, small scalars in general.  A million returning atomics on ONE address serialise (measured:
      // 2^20 {0,1} scalars 7.2 ms against 4.9 ms uniform), so the lanes of a wave that hit bucket 0 share one atomic.
      const bool valid = code != MSM_SKIP;
      const uint32_t bucket = code & ~MSM_NEG;
      const bool hot = valid && bucket == 0u;
      const unsigned long long hm = __ballot(hot);
      if (hot) {
        const int leader = __ffsll((long long)hm) - 1;
        uint32_t base = 0;
        if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&hist[(size_t)w * pl.nb], (uint32_t)__popcll(hm));
        base = __shfl(base, leader, 64);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
        ranks[(size_t)w * ne + j] = base + below;
      } else if (valid) {
        ranks[(size_t)w * ne + j] = atomicAdd(&hist[(size_t)w * pl.nb + bucket], 1u);
      }
    }
  }
}

// The top window of the GLV form holds only the last few bits (2 for c = 16): a million entries would
// hammer a handful of counters.  Ranks come from LDS counters per workgroup plus one global atomic per
// workgroup and distinct bucket (at most 32 buckets: top_bits <= 5).
constexpr int RT_PER = 8;  // entries per thread of k_rank_top
__global__ void __launch_bounds__(256) k_rank_top(uint32_t ne, MsmPlan pl, const uint32_t* __restrict__ codes,
                                                 uint32_t* __restrict__ ranks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t cnt[32], base[32];
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int w = pl.W - 1;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t local[RT_PER], bkt[RT_PER];
#pragma unroll
  for (int i = 0; i < RT_PER; i++) {
    const uint32_t j = (blockIdx.x * RT_PER + i) * 256u + threadIdx.x;
    const uint32_t code = j < ne ? codes[(size_t)w * ne + j] : MSM_SKIP;
    const bool valid = code != MSM_SKIP;
    const uint32_t bucket = (code & ~MSM_NEG) & 31u;
    bkt[i] = valid ? bucket : 0xFFFFFFFFu;
    local[i] = 0;
    // one LDS atomic per wave and distinct bucket (there are one or two): the lanes of a bucket rank themselves by ballot
    uint64_t todo = __ballot(valid);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t b = (uint32_t)__shfl((int)bucket, leader, 64);
      const bool mine = valid && bucket == b;
      const uint64_t m = __ballot(mine);
      uint32_t basew = 0;
      if (lane == (uint32_t)leader) basew = atomicAdd(&cnt[b], (uint32_t)__popcll(m));
      basew = (uint32_t)__shfl((int)basew, leader, 64);
      if (mine) local[i] = basew + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32 && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&hist[(size_t)w * pl.nb + threadIdx.x], cnt[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RT_PER; i++) {
    const uint32_t j = (blockIdx.x * RT_PER + i) * 256u + threadIdx.x;
    if (bkt[i] != 0xFFFFFFFFu) ranks[(size_t)w * ne + j] = base[bkt[i]] + local[i];
  }
}

// the same in two steps, so that a loop can issue the load, do its arithmetic, and apply the endomorphism afterwards
__device__ __forceinline__ affine entry_load(const uint8_t* __restrict__ points, uint32_t e, uint32_t glv) {
  uint32_t idx = e & ~MSM_NEG;
  return ld_affine(points + (size_t)(glv ? idx >> 1 : idx) * 64);
}
__device__ __forceinline__ affine entry_finish(const affine& raw, uint32_t e, uint32_t glv) {
  return (glv && (e & 1u)) ? glv_phi(raw) : raw;
}
// point of a sorted entry: plain index, or (index >> 1, endomorphism flag) in GLV form
__device__ __forceinline__ affine entry_point(const uint8_t* __restrict__ points, uint32_t e, uint32_t glv) {
  uint32_t idx = e & ~MSM_NEG;
  if (!glv) return ld_affine(points + (size_t)idx * 64);
  affine p = ld_affine(points + (size_t)(idx >> 1) * 64);
  if (idx & 1u) p = glv_phi(p);
  return p;
}

// ---- two-level bucket sort ------------------------------------------------------------------------------------
// The one-level ranking above spends one device-scope returning atomic per entry (19 M at N = 2^20: 1.1 ms, two thirds
// of it waiting).  Here the entries of a window are first partitioned by the HIGH bits of their bucket into coarse
// bins of 2^FINE_BITS buckets (histogram and placement through LDS counters of a 1024-thread workgroup; one global
// atomic per workgroup and bin), then one workgroup per coarse bin counting-sorts its entries by the low bucket bits
// entirely in LDS and emits the per-bucket counts and offsets the accumulation kernels read.
constexpr int FINE_BITS = 7, FINE = 1 << FINE_BITS;
constexpr int CO_MAXCNT = 17 * 256;   // (window, coarse bin) counters: W <= 17 for c >= 8, nb / 128 <= 256
constexpr int CO_SCALARS = 4096;      // scalars per workgroup of k_recode_coarse (8,192 half-scalar entries per window)
constexpr int CO_TILE = 8192;         // entries per window and workgroup of k_scatter_coarse

__global__ void __launch_bounds__(1024) k_recode_coarse(const uint8_t* __restrict__ scalars, uint32_t n, uint32_t i0, uint32_t i1,
                                                       MsmPlan pl, int wc, int nbin, uint32_t* __restrict__ codes,
                                                       uint32_t* __restrict__ chist) {
  __shared__ uint32_t lh[CO_MAXCNT];
  const int ncnt = wc * nbin;
  for (int k = threadIdx.x; k < ncnt; k += 1024) lh[k] = 0;
  __syncthreads();
  const size_t ne = (size_t)2 * n;
  for (int sidx = 0; sidx < CO_SCALARS / 1024; sidx++) {
    const uint32_t i = i0 + blockIdx.x * CO_SCALARS + sidx * 1024 + threadIdx.x;
    if (i < i1) {
      u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)i * 32));
      GlvHalf h[2];
      glv_split(k, h[0], h[1]);
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        u256 sv = u256_zero();
#pragma unroll
        for (int q = 0; q < 5; q++) sv.v[q] = h[hh].m[q];
        const uint32_t flip = h[hh].neg ? MSM_NEG : 0u;
        const size_t j = (size_t)2 * i + hh;
        uint32_t carry = 0;
        for (int w = 0; w < pl.W; w++) {
          uint32_t code = recode_next(sv, pl.c, carry);
          if (code != MSM_SKIP) code ^= flip;
          codes[(size_t)w * ne + j] = code;
          if (w < wc && code != MSM_SKIP) atomicAdd(&lh[w * nbin + (int)((code & ~MSM_NEG) >> FINE_BITS)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ncnt; k += 1024)
    if (lh[k]) atomicAdd(&chist[k], lh[k]);
}

// cstart = exclusive scan of chist over the cnt (window, bin) counters; ccur = cstart.
// Also cuts every bin into tiles of FINE_TILE entries for the fine sort below: tile_start[k] = first tile of bin k
// (tile_start[cnt] = number of tiles), tile_bin[t] = the bin tile t belongs to.  A uniform 2^20-point sum has ~8,192
// entries a bin (two tiles); structured scalars pile half a million entries onto ONE bin ({0,1} scalars: digit 1 of
// window 0; 64-bit scalars: the carry digit behind their top window), which a single workgroup took 1 ms to sort.
constexpr uint32_t FINE_TILE = 4096;
// every bin owns at least one tile (an empty bin's tile just stores its buckets' starts), so tile_start is strictly
// increasing; the tiles of a call number at most (bins) + (entries) / FINE_TILE
__host__ __device__ inline uint32_t fine_tiles(uint32_t c) { return c ? (c + FINE_TILE - 1) / FINE_TILE : 1u; }
__global__ void __launch_bounds__(1024) k_coarse_scan(const uint32_t* __restrict__ chist, uint32_t cnt,
                                                     uint32_t* __restrict__ cstart, uint32_t* __restrict__ ccur,
                                                     uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_bin,
                                                     uint32_t max_tiles) {
  __shared__ uint32_t part[1024], tpart[1024];
  __shared__ uint32_t ts[CO_MAXCNT + 1];
  const int t = threadIdx.x;
  const uint32_t per = (cnt + 1023) / 1024;
  uint32_t sum = 0, tsum = 0;
  for (uint32_t k = 0; k < per; k++) {
    uint32_t b = t * per + k;
    if (b < cnt) {
      const uint32_t c = chist[b];
      sum += c;
      tsum += fine_tiles(c);
    }
  }
  part[t] = sum;
  tpart[t] = tsum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0, tv = t >= d ? tpart[t - d] : 0;
    __syncthreads();
    part[t] += v;
    tpart[t] += tv;
    __syncthreads();
  }
  uint32_t run = part[t] - sum, trun = tpart[t] - tsum;
  for (uint32_t k = 0; k < per; k++) {
    uint32_t b = t * per + k;
    if (b < cnt) {
      const uint32_t c = chist[b];
      cstart[b] = run;
      ccur[b] = run;
      run += c;
      ts[b] = trun;
      tile_start[b] = trun;
      trun += fine_tiles(c);
    }
  }
  const uint32_t ntiles = tpart[1023];
  if (t == 0) {
    ts[cnt] = ntiles;
    tile_start[cnt] = ntiles;
  }
  __syncthreads();
  // tile -> bin: the bin whose first tile is the last one <= tile
  for (uint32_t tile = t; tile < ntiles && tile < max_tiles; tile += 1024) {
    uint32_t lo = 0, hi = cnt - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (ts[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    tile_bin[tile] = lo;
  }
}

// coarse[pos] = entry index (24 bits) | low bucket bits << 24 | sign, grouped by (window, coarse bin)
__global__ void __launch_bounds__(1024) k_scatter_coarse(const uint32_t* __restrict__ codes, uint32_t ne, MsmPlan pl, int wc,
                                                        int nbin, uint32_t* __restrict__ ccur, uint32_t* __restrict__ coarse) {
  __shared__ uint32_t lcnt[CO_MAXCNT], lbase[CO_MAXCNT];
  const int ncnt = wc * nbin;
  for (int k = threadIdx.x; k < ncnt; k += 1024) lcnt[k] = 0;
  __syncthreads();
  const uint32_t j0 = blockIdx.x * CO_TILE;
  for (int w = 0; w < wc; w++)
    for (int sidx = 0; sidx < CO_TILE / 1024; sidx++) {
      const uint32_t j = j0 + sidx * 1024 + threadIdx.x;
      if (j < ne) {
        const uint32_t code = codes[(size_t)w * ne + j];
        if (code != MSM_SKIP) atomicAdd(&lcnt[w * nbin + (int)((code & ~MSM_NEG) >> FINE_BITS)], 1u);
      }
    }
  __syncthreads();
  for (int k = threadIdx.x; k < ncnt; k += 1024) {
    const uint32_t c = lcnt[k];
    lbase[k] = c ? atomicAdd(&ccur[k], c) : 0u;
    lcnt[k] = 0;
  }
  __syncthreads();
  for (int w = 0; w < wc; w++)
    for (int sidx = 0; sidx < CO_TILE / 1024; sidx++) {
      const uint32_t j = j0 + sidx * 1024 + threadIdx.x;
      if (j < ne) {
        const uint32_t code = codes[(size_t)w * ne + j];
        if (code != MSM_SKIP) {
          const uint32_t bucket = code & ~MSM_NEG;
          const int k = w * nbin + (int)(bucket >> FINE_BITS);
          const uint32_t pos = lbase[k] + atomicAdd(&lcnt[k], 1u);
          coarse[pos] = j | ((bucket & (FINE - 1)) << 24) | (code & MSM_NEG);
        }
      }
    }
}

// Ranks for a counting sort through LDS counters: returns the counter's value before this lane's increment.  When a
// crowd of the wave's lanes meets on one counter (structured scalars: every entry of a tile in the same bucket) the
// crowd shares ONE atomic and ranks itself by ballot; lanes outside a crowd take their own atomic (the uniform case:
// one ballot and a compare of overhead).
__device__ __forceinline__ uint32_t lds_rank(uint32_t* cnt, uint32_t key, bool valid) {
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t todo = __ballot(valid);
  uint32_t r = 0;
  for (int it = 0; it < 4 && todo; it++) {
    const int leader = __builtin_ctzll(todo);
    const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
    const bool same = ((todo >> lane) & 1ull) && key == k0;
    const uint64_t sm = __ballot(same);
    if (__popcll(sm) < 4) break;
    uint32_t base = 0;
    if (lane == (uint32_t)leader) base = atomicAdd(&cnt[k0], (uint32_t)__popcll(sm));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (same) r = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
    todo &= ~sm;
  }
  if ((todo >> lane) & 1ull) r = atomicAdd(&cnt[key], 1u);
  return r;
}

// Fine sort, step 1.  One workgroup per TILE of a (window, coarse bin): histogram of the tile's low bucket bits in LDS,
// then one global atomic per non-empty bucket adds it to hist (the bucket's final population, which the accumulation
// reads) and returns the tile's first rank inside that bucket (tbase).  A bin of one tile gets rank 0 everywhere.
__global__ void __launch_bounds__(256) k_fine_count(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ cstart,
                                                   const uint32_t* __restrict__ chist, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_bin, uint32_t ncnt, MsmPlan pl, int nbin,
                                                   uint32_t* __restrict__ hist, uint32_t* __restrict__ tbase) {
  __shared__ uint32_t cnt[FINE];
  const uint32_t tile = blockIdx.x;
  if (tile >= tile_start[ncnt]) return;
  const uint32_t k = tile_bin[tile], w = k / (uint32_t)nbin, b = k % (uint32_t)nbin;
  const uint32_t s0 = cstart[k] + (tile - tile_start[k]) * FINE_TILE, end = cstart[k] + chist[k];
  const uint32_t s1 = s0 + FINE_TILE < end ? s0 + FINE_TILE : end;
  if (threadIdx.x < FINE) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t p0 = s0; p0 < s1; p0 += 256) {
    const uint32_t p = p0 + threadIdx.x;
    const bool valid = p < s1;
    const uint32_t key = valid ? (coarse[p] >> 24) & (FINE - 1) : 0u;
    (void)lds_rank(cnt, key, valid);
  }
  __syncthreads();
  if (threadIdx.x < FINE) {
    const uint32_t c = cnt[threadIdx.x];
    const size_t bid = (size_t)w * pl.nb + (size_t)b * FINE + threadIdx.x;
    tbase[(size_t)tile * FINE + threadIdx.x] = c ? atomicAdd(&hist[bid], c) : 0u;
  }
}

// Fine sort, step 2.  Every tile scans its bin's bucket populations (hist is complete: the kernel boundary) into bucket
// starts - the bin's first tile also stores them to offs - and places its entries (index | sign) at
// start(bucket) + tbase(tile, bucket) + rank inside the tile.
__global__ void __launch_bounds__(256) k_fine_place(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ cstart,
                                                   const uint32_t* __restrict__ chist, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ tile_bin, uint32_t ncnt, MsmPlan pl, int nbin,
                                                   const uint32_t* __restrict__ hist, const uint32_t* __restrict__ tbase,
                                                   uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t start[FINE], cur[FINE], wtot[2];
  const uint32_t tile = blockIdx.x;
  if (tile >= tile_start[ncnt]) return;
  const uint32_t k = tile_bin[tile], w = k / (uint32_t)nbin, b = k % (uint32_t)nbin;
  const uint32_t first = cstart[k], end = first + chist[k];
  const uint32_t s0 = first + (tile - tile_start[k]) * FINE_TILE;
  const uint32_t s1 = s0 + FINE_TILE < end ? s0 + FINE_TILE : end;
  const size_t bid0 = (size_t)w * pl.nb + (size_t)b * FINE;
  if (threadIdx.x < FINE) {  // exclusive scan of the 128 populations by the first two waves
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t c = hist[bid0 + threadIdx.x];
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
      if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wtot[wv] = incl;
    start[threadIdx.x] = incl - c;
  }
  __syncthreads();
  if (threadIdx.x < FINE) {
    const uint32_t st = first + start[threadIdx.x] + (threadIdx.x >= 64 ? wtot[0] : 0u);
    start[threadIdx.x] = st + tbase[(size_t)tile * FINE + threadIdx.x];
    cur[threadIdx.x] = 0;
    if (tile == tile_start[k]) offs[bid0 + threadIdx.x] = st;
  }
  __syncthreads();
  for (uint32_t p0 = s0; p0 < s1; p0 += 256) {
    const uint32_t p = p0 + threadIdx.x;
    const bool valid = p < s1;
    const uint32_t v = valid ? coarse[p] : 0u;
    const uint32_t key = (v >> 24) & (FINE - 1);
    const uint32_t rk = lds_rank(cur, key, valid);
    if (valid) {
      const uint32_t pos = start[key] + rk;
      if (pos < end) sorted[pos] = (v & 0x00FFFFFFu) | (v & MSM_NEG);
    }
  }
}

// ---- 2. per-window exclusive scan (one workgroup of 1024 per window) -----------
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ hist, uint32_t n, MsmPlan pl, int w0,
                                              uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor) {
  __shared__ uint32_t part[1024];
  int w = w0 + blockIdx.x, t = threadIdx.x;
  int per = (pl.nb + 1023) / 1024;
  const uint32_t* h = hist + (size_t)w * pl.nb;
  const bool vec = (per & 3) == 0 && (pl.nb & 3) == 0;  // a thread's run as uint4 pieces: independent loads in flight together
  uint32_t sum = 0;
  if (vec) {
    const uint4* h4 = (const uint4*)h;
    for (int k = 0; k < per / 4; k++) {
      int b4 = t * (per / 4) + k;
      if (b4 * 4 < pl.nb) {
        uint4 v = h4[b4];
        sum += v.x + v.y + v.z + v.w;
      }
    }
  } else {
    for (int k = 0; k < per; k++) {
      int b = t * per + k;
      if (b < pl.nb) sum += h[b];
    }
  }
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = (uint32_t)((size_t)w * n) + part[t] - sum;  // exclusive
  if (vec) {
    const uint4* h4 = (const uint4*)h;
    uint4* o4 = (uint4*)(offs + (size_t)w * pl.nb);
    uint4* c4 = (uint4*)(cursor + (size_t)w * pl.nb);
    for (int k = 0; k < per / 4; k++) {
      int b4 = t * (per / 4) + k;
      if (b4 * 4 < pl.nb) {
        uint4 v = h4[b4], o;
        o.x = run;
        o.y = o.x + v.x;
        o.z = o.y + v.y;
        o.w = o.z + v.z;
        run = o.w + v.w;
        o4[b4] = o;
        c4[b4] = o;
      }
    }
  } else {
    for (int k = 0; k < per; k++) {
      int b = t * per + k;
      if (b < pl.nb) {
        offs[(size_t)w * pl.nb + b] = run;
        cursor[(size_t)w * pl.nb + b] = run;
        run += h[b];
      }
    }
  }
}

// ---- 3. scatter ------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ ranks,
                                                uint32_t n, MsmPlan pl, int w0, const uint32_t* __restrict__ offs,
                                                uint32_t* __restrict__ sorted) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  int w = w0 + blockIdx.y;
  if (i >= n) return;
  uint32_t code = codes[(size_t)w * n + i];
  if (code == MSM_SKIP) return;
  uint32_t pos = offs[(size_t)w * pl.nb + (code & ~MSM_NEG)] + ranks[(size_t)w * n + i];
  sorted[pos] = i | (code & MSM_NEG);
}

// ---- 4. order buckets by population (descending) -------------------------------
__device__ __forceinline__ int pop_bin(uint32_t cnt) { return cnt > (uint32_t)LIGHT_MAX ? 0 : (LIGHT_MAX + 1 - (int)cnt); }

__global__ void __launch_bounds__(1024) k_cls_hist(const uint32_t* __restrict__ hist, uint32_t total,
                                                  uint32_t* __restrict__ bins) {
  __shared__ uint32_t lh[NBINS];
  for (int k = threadIdx.x; k < NBINS; k += blockDim.x) lh[k] = 0;
  __syncthreads();
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < total) atomicAdd(&lh[pop_bin(hist[t])], 1u);
  __syncthreads();
  for (int k = threadIdx.x; k < NBINS; k += blockDim.x)
    if (lh[k]) atomicAdd(&bins[k], lh[k]);
}
// bins[0..NBINS) counts -> bins[NBINS..2*NBINS) running cursors (exclusive scan); bins[2*NBINS] = #heavy
__global__ void k_cls_scan(uint32_t* bins) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t run = 0;
    for (int k = 0; k < NBINS; k++) {
      bins[NBINS + k] = run;
      run += bins[k];
    }
    // heavy threshold: buckets are ordered by population (descending), so "cnt > thr" is a prefix of `order`
    uint32_t nonempty = bins[0], thr = LIGHT_MAX, nheavy = bins[0];
    for (int pop = 1; pop <= LIGHT_MAX; pop++) nonempty += bins[LIGHT_MAX + 1 - pop];
    if (nonempty < LIGHT_FILL) {
      for (int pop = LIGHT_MAX; pop > (int)LIGHT_MIN; pop--) {
        const uint32_t c = bins[LIGHT_MAX + 1 - pop];
        if (nheavy + c > bins[0] + HEAVY_EXTRA) break;
        nheavy += c;
        thr = (uint32_t)pop - 1;
      }
    }
    bins[2 * NBINS] = nheavy;
    bins[2 * NBINS + 1] = thr;
  }
}
__global__ void __launch_bounds__(1024) k_cls_scatter(const uint32_t* __restrict__ hist, uint32_t total,
                                                     uint32_t* __restrict__ bins, uint32_t* __restrict__ order) {
  __shared__ uint32_t lh[NBINS];
  __shared__ uint32_t lbase[NBINS];
  for (int k = threadIdx.x; k < NBINS; k += blockDim.x) lh[k] = 0;
  __syncthreads();
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  int bin = 0;
  uint32_t rank = 0;
  if (t < total) {
    bin = pop_bin(hist[t]);
    rank = atomicAdd(&lh[bin], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NBINS; k += blockDim.x)
    if (lh[k]) lbase[k] = atomicAdd(&bins[NBINS + k], lh[k]);
  __syncthreads();
  if (t < total) order[lbase[bin] + rank] = t;
}

// ---- 5a. light buckets: one lane per bucket ----------------------------------------
// shift_stride != 0: `points` holds, for window w, the copies 2^(c w) P_i at points + w * shift_stride (bp_msm_points)
__device__ __forceinline__ void light_lane(uint32_t t, const uint8_t* __restrict__ points, const uint32_t* __restrict__ hist,
                                           const uint32_t* __restrict__ offs, const uint32_t* __restrict__ sorted,
                                           const uint32_t* __restrict__ order, uint32_t total, uint32_t glv, uint32_t nb,
                                           size_t shift_stride, uint32_t light_max, xyzz* __restrict__ buckets) {
  if (t >= total) return;
  uint32_t bid = order[t];
  uint32_t cnt = hist[bid];
  points += (size_t)(bid / nb) * shift_stride;
  if (cnt > light_max) return;  // heavy path owns it
  // the bucket's sum on nine 29-bit limbs (fq29.h): started from the first point, one lazy mixed addition per further
  // point; input points may be O, which the lazy formula does not take
  xyzz29 acc = xyzz29_inf();
  if (cnt) {
    const uint32_t* lst = sorted + offs[bid];
    uint32_t e = lst[0];
    affine p = entry_point(points, e, glv);
    acc = xyzz29_from_affine(p, (e & MSM_NEG) != 0);
    affine raw = p;
    if (cnt > 1) {
      e = lst[1];
      raw = entry_load(points, e, glv);
    }
    for (uint32_t k = 1; k < cnt; k++) {
      const affine cur = entry_finish(raw, e, glv);
      const uint32_t ecur = e;
      if (k + 1 < cnt) {  // the next point's load is in flight during the addition; its endomorphism waits for the next step
        e = lst[k + 1];
        raw = entry_load(points, e, glv);
      }
      if (!affine_is_inf(cur)) acc = xyzz29_madd(acc, cur, (ecur & MSM_NEG) != 0);
    }
  }
  st_xyzz(&buckets[bid], xyzz29_to(acc));
}

// ---- 5b. heavy buckets -----------------------------------------------------------------
// heavy_off[h] = first slice of heavy bucket h (h-th entry of `order`); heavy_off[nheavy] = total
__global__ void __launch_bounds__(1024) k_heavy_plan(const uint32_t* __restrict__ hist,
                                                    const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ bins,
                                                    uint32_t* __restrict__ heavy_off) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t base;
  uint32_t nheavy = bins[2 * NBINS];
  int t = threadIdx.x;
  if (t == 0) base = 0;
  __syncthreads();
  for (uint32_t start = 0; start < nheavy; start += 1024) {
    uint32_t h = start + t;
    uint32_t sl = 0;
    if (h < nheavy) sl = (hist[order[h]] + SLICE - 1) / SLICE;
    part[t] = sl;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      uint32_t v = t >= d ? part[t - d] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    if (h < nheavy) heavy_off[h] = base + part[t] - sl;
    __syncthreads();
    if (t == 1023) base += part[1023];
    __syncthreads();
  }
  if (t == 0) heavy_off[nheavy] = base;
}

// slices [first, total) step `stride` of the heavy buckets, one workgroup per slice
__device__ __forceinline__ void heavy_slices(uint32_t first, uint32_t stride, const uint8_t* __restrict__ points,
                                             const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                             const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ order,
                                             const uint32_t* __restrict__ bins, const uint32_t* __restrict__ heavy_off,
                                             uint32_t glv, uint32_t nb, size_t shift_stride, xyzz* __restrict__ partial,
                                             xyzz* lds) {
  uint32_t nheavy = bins[2 * NBINS];
  if (nheavy == 0) return;
  uint32_t total = heavy_off[nheavy];
  for (uint32_t j = first; j < total; j += stride) {
    // largest h with heavy_off[h] <= j
    uint32_t lo = 0, hi = nheavy - 1;
    while (lo < hi) {
      uint32_t mid = (lo + hi + 1) >> 1;
      if (heavy_off[mid] <= j) lo = mid; else hi = mid - 1;
    }
    uint32_t bid = order[lo];
    uint32_t cnt = hist[bid];
    uint32_t s0 = (j - heavy_off[lo]) * SLICE;
    uint32_t s1 = s0 + SLICE < cnt ? s0 + SLICE : cnt;
    const uint32_t* lst = sorted + offs[bid];
    const uint8_t* wpoints = points + (size_t)(bid / nb) * shift_stride;
    xyzz acc = xyzz_inf();
    for (uint32_t k = s0 + threadIdx.x; k < s1; k += blockDim.x) {
      uint32_t e = lst[k];
      affine p = entry_point(wpoints, e, glv);
      bucket_add(acc, p, (e & MSM_NEG) != 0);
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) st_xyzz(&partial[j], acc);
  }
}

// Bucket accumulation in ONE launch: the first heavy_blocks workgroups walk the slices of the heavy buckets (the narrow
// top window of the GLV form always has some), the others run one lane per light bucket; the heavy slices used to
// wait behind the light kernel in two launches of their own.
__global__ void __launch_bounds__(256) k_accumulate(const uint8_t* __restrict__ points, const uint32_t* __restrict__ hist,
                                                   const uint32_t* __restrict__ offs, const uint32_t* __restrict__ sorted,
                                                   const uint32_t* __restrict__ order, const uint32_t* __restrict__ bins,
                                                   const uint32_t* __restrict__ heavy_off, uint32_t total, uint32_t glv,
                                                   uint32_t nb, size_t shift_stride, uint32_t heavy_blocks,
                                                   xyzz* __restrict__ buckets, xyzz* __restrict__ partial) {
  __shared__ xyzz lds[4];
  if (blockIdx.x >= heavy_blocks)
    light_lane((blockIdx.x - heavy_blocks) * blockDim.x + threadIdx.x, points, hist, offs, sorted, order, total, glv, nb,
               shift_stride, bins[2 * NBINS + 1], buckets);
  else
    heavy_slices(blockIdx.x, heavy_blocks, points, hist, offs, sorted, order, bins, heavy_off, glv, nb, shift_stride, partial,
                 lds);
}

// one wave per heavy bucket: sum its slice partials
__global__ void __launch_bounds__(64) k_heavy_final(const uint32_t* __restrict__ order,
                                                   const uint32_t* __restrict__ bins,
                                                   const uint32_t* __restrict__ heavy_off,
                                                   const xyzz* __restrict__ partial, xyzz* __restrict__ buckets) {
  uint32_t nheavy = bins[2 * NBINS];
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    uint32_t a = heavy_off[h], b = heavy_off[h + 1];
    xyzz acc = xyzz_inf();
    for (uint32_t k = a + threadIdx.x; k < b; k += 64) acc = xyzz_add(acc, ld_xyzz(&partial[k]));
    acc = wave_sum(acc);
    if (threadIdx.x == 0) st_xyzz(&buckets[order[h]], acc);
  }
}

// ---- 5c. prepared points: all windows carry weight 1, so their buckets add up index by index ----------------
// buckets[b] += sum_{w >= 1} buckets[w * nb + b]
__global__ void __launch_bounds__(256) k_combine_windows(xyzz* __restrict__ buckets, uint32_t W, uint32_t nb) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  xyzz acc = ld_xyzz(&buckets[b]);
  for (uint32_t w = 1; w < W; w++) acc = xyzz_add(acc, ld_xyzz(&buckets[(size_t)w * nb + b]));
  st_xyzz(&buckets[b], acc);
}

// out[w][i] = 2^(c w) * P_i, affine, for w < W: c doublings between copies, then ONE inversion per point for all its
// copies (Montgomery's trick with the prefix products parked in the output slots).  scratch: [W][n] XYZZ.
__global__ void __launch_bounds__(256) k_msm_shift(const uint8_t* __restrict__ points, uint32_t n, int c, int W,
                                                  xyzz* __restrict__ scratch, uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzz cur = xyzz_from_affine(ld_affine(points + (size_t)i * 64));
  for (int w = 0; w < W; w++) {
    st_xyzz(&scratch[(size_t)w * n + i], cur);
    if (w + 1 < W)
      for (int k = 0; k < c; k++) cur = xyzz_dbl(cur);
  }
  u256 run = u256_one();
  for (int w = 0; w < W; w++) {
    uint8_t* slot = out + ((size_t)w * n + i) * 64;
    st_u256(slot, run);  // product of the earlier copies' ZZZ
    u256 zzz = ld_u256(&scratch[(size_t)w * n + i].ZZZ);
    if (!fq_is_zero(zzz)) run = fq_mul(run, zzz);  // the identity (and anything that doubled into it) stays out
  }
  u256 inv = fq_inv_gcd(run);
  for (int w = W - 1; w >= 0; w--) {
    uint8_t* slot = out + ((size_t)w * n + i) * 64;
    xyzz e = ld_xyzz(&scratch[(size_t)w * n + i]);
    if (fq_is_zero(e.ZZZ)) {
      st_affine(slot, affine_inf());
      continue;
    }
    u256 zi = fq_mul(inv, ld_u256(slot));  // 1 / ZZZ_w
    inv = fq_mul(inv, e.ZZZ);
    st_affine(slot, xyzz_to_affine_with_inv(e, zi));
  }
}

// ---- 6. segment running sums ----------------------------------------------------------
__global__ void __launch_bounds__(256) k_seg_reduce(const xyzz* __restrict__ buckets, MsmPlan pl,
                                                   xyzz* __restrict__ segv) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = (uint32_t)pl.W * pl.nseg;
  if (t >= total) return;
  int sg = t % pl.nseg;
  const xyzz* base = buckets + (size_t)t * pl.seg;  // (w*nseg + sg)*seg = w*nb + sg*seg
  xyzz S, T;
  segment_reduce(pl.seg, [&](int j) { return ld_xyzz(&base[j]); }, S, T);
  st_xyzz(&segv[t], segment_value(pl, sg, S, T));
}

// The same with every segment spread over a quad of lanes (lanes29.h): the 2 seg additions and the multiplication by the
// segment index are one dependent chain (~40 point operations), 3.1 / 2.7 us per addition / doubling on a quad against
// 5.7 / 4.2 us in one lane.  Used while four lanes per segment still fit one wave per SIMD (W nseg <= 16,384: the
// 14-bit plans of 2^14..2^17 terms); at 2^20 (36,864 segments) the quad form puts two waves on every SIMD and gains
// nothing.  Identity operands (empty buckets, S before the first non-empty bucket) are answered by selects.
__global__ void __launch_bounds__(256) k_seg_reduce_quad(const xyzz* __restrict__ buckets, MsmPlan pl, int seg_log,
                                                        int idx_bits, xyzz* __restrict__ segv) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, lane = threadIdx.x & 3u;
  const uint32_t total = (uint32_t)pl.W * pl.nseg;
  if (t >= total) return;  // whole quads leave together
  const uint32_t sg = t % (uint32_t)pl.nseg;
  const xyzz* base = buckets + (size_t)t * pl.seg;
  fe29 S = fe29_zero(), T = fe29_zero();
#pragma unroll 1
  for (int j = pl.seg - 1; j >= 0; j--) {
    S = quad_add_total(S, quad_load(&base[j], lane), lane);
    T = quad_add_total(T, S, lane);
  }
  // segment_value: T + (sg * seg) * S, the small multiple by double-and-add from the top bit of sg
  fe29 m = fe29_zero();
#pragma unroll 1
  for (int i = idx_bits - 1; i >= 0; i--) {
    m = quad_dbl(m, lane);
    const fe29 a = quad_add_total(m, S, lane);
    m = fe29_sel(((sg >> i) & 1u) != 0u, a, m);
  }
#pragma unroll 1
  for (int i = 0; i < seg_log; i++) m = quad_dbl(m, lane);
  T = quad_add_total(T, m, lane);
  quad_store(&segv[t], T, lane);
}

// ---- 7. window sums ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_window_sum(const xyzz* __restrict__ segv, MsmPlan pl,
                                                   xyzz* __restrict__ wsum) {
  __shared__ xyzz lds[4];
  int w = blockIdx.x;
  xyzz acc = xyzz_inf();
  for (int s = threadIdx.x; s < pl.nseg; s += blockDim.x) acc = xyzz_add(acc, ld_xyzz(&segv[(size_t)w * pl.nseg + s]));
  acc = block_sum(acc, lds);
  if (threadIdx.x == 0) st_xyzz(&wsum[w], acc);
}

// ---- 7q. window sums with every point spread over a quad of lanes (lanes29.h) -----------------------------------------
// A quad does an addition in 4 rounds of products instead of 14, on the 29-bit limbs (3.1 us, against 5.7 us in one lane
// on those limbs and 6.4 us on 8 x 32, at one wave per SIMD); each lane loads and stores only its own coordinate.  (The segment step in the same form was measured too:
// four times the lanes put two waves on every SIMD and the kernel took the same 0.23 ms, so it stays one lane per
// segment.)
// out[w][g] = sum of slice g of in[w][0..n): one workgroup of 256 lanes (64 quads, one wave per SIMD: two waves on a SIMD
// would halve each chain's speed) per (slice, window); strided partial sums, then a tree over the quads through LDS.
// Called twice: segment values -> WSQ_GROUPS partial sums per window -> the window sums.
constexpr uint32_t WSQ_GROUPS = 16;
__global__ void __launch_bounds__(256) k_window_sum_quad(const xyzz* __restrict__ in, uint32_t n, uint32_t groups,
                                                        xyzz* __restrict__ out) {
  __shared__ uint32_t tree[64 * 4 * 9];
  const uint32_t q = threadIdx.x >> 2, lane = threadIdx.x & 3u, g = blockIdx.x, w = blockIdx.y;
  const uint32_t per = (n + groups - 1) / groups, lo = g * per, hi = lo + per < n ? lo + per : n;
  const uint32_t cnt = hi > lo ? (hi - lo < 64u ? hi - lo : 64u) : 0u;  // quads that hold a value
  fe29 acc = fe29_zero();
  bool have = false;
#pragma unroll 1
  for (uint32_t s = lo + q; s < hi; s += 64u) {
    const fe29 v = quad_load(&in[(size_t)w * n + s], lane);
    acc = have ? quad_add(acc, v, lane) : v;
    have = true;
  }
#pragma unroll 1
  for (uint32_t m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int i = 0; i < 9; i++) tree[(q * 4 + lane) * 9 + i] = acc.v[i];
    __syncthreads();
    if (q < m && q + m < cnt) {
      fe29 o;
#pragma unroll
      for (int i = 0; i < 9; i++) o.v[i] = tree[((q + m) * 4 + lane) * 9 + i];
      acc = quad_add(acc, o, lane);
    }
    __syncthreads();
  }
  if (q == 0) quad_store(&out[(size_t)w * groups + g], acc, lane);  // an empty slice stores the identity (all zero)
}

// ---- 6'/7'. the same weighted sums by suffix scans (nb <= 65,536) -------------------------------------------------
// sum_b (b + 1) B[b] = sum_b suffix(b), suffix(b) = sum_{b' >= b} B[b'].  A workgroup scans 256 buckets (8 additions
// deep) and sums its suffixes (8 more): tb = sum_t (t + 1) B[256 blk + t] and the workgroup total bt.  Then
//   R_w = sum_blk tb[blk] + 256 * sum_blk blk * bt[blk],
// the second sum being the suffixes of bt from blk = 1 on: one more scan over at most 256 totals.  About 40 additions
// deep in all, against ~75 for the segment running sums above (whose lanes also multiply by the segment index).
__global__ void __launch_bounds__(256) k_bucket_scan(const xyzz* __restrict__ buckets, uint32_t nb, uint32_t nblk,
                                                    xyzz* __restrict__ bt /*[W][nblk]*/, xyzz* __restrict__ tb /*[W][nblk]*/) {
  __shared__ xyzz wt[4], red[4];
  const uint32_t w = blockIdx.y, blk = blockIdx.x, b = blk * 256u + threadIdx.x;
  xyzz v = xyzz_inf();
  if (b < nb) v = ld_xyzz(&buckets[(size_t)w * nb + b]);
  v = block_suffix_scan(v, wt);
  if (threadIdx.x == 0) st_xyzz(&bt[(size_t)w * nblk + blk], v);
  xyzz s = block_sum(v, red);
  if (threadIdx.x == 0) st_xyzz(&tb[(size_t)w * nblk + blk], s);
}
// wsum2[w][0] = 256 * sum_blk blk * bt[w][blk], wsum2[w][1] = sum_blk tb[w][blk]; blockIdx.y selects the half
__global__ void __launch_bounds__(256) k_bucket_sum(const xyzz* __restrict__ bt, const xyzz* __restrict__ tb, uint32_t nblk,
                                                   xyzz* __restrict__ wsum2 /*[W][2]*/) {
  __shared__ xyzz wt[4], red[4];
  const uint32_t w = blockIdx.x, t = threadIdx.x;
  xyzz v = xyzz_inf();
  if (blockIdx.y == 1) {
    if (t < nblk) v = ld_xyzz(&tb[(size_t)w * nblk + t]);
    xyzz s = block_sum(v, red);
    if (t == 0) st_xyzz(&wsum2[2 * w + 1], s);
    return;
  }
  if (t < nblk) v = ld_xyzz(&bt[(size_t)w * nblk + t]);
  v = block_suffix_scan(v, wt);
  xyzz s = block_sum(t >= 1 ? v : xyzz_inf(), red);
  if (t == 0) {
    if (nblk > 1)
      for (int i = 0; i < 8; i++) s = xyzz_dbl(s);
    st_xyzz(&wsum2[2 * w], s);
  }
}
// Horner over the window sums, sum_w 2^(c w) S_w: (W - 1) c DEPENDENT doublings in a single wave whose latency nothing
// else hides (128 of them for 2^20 points: 0.8 ms of a 2.7 ms MSM on the 8 x 32 limbs of ec.h, one lane).  The chain runs
// on the 29-bit limbs of fq29.h, and a doubling is spread over the four lanes of a quad (lanes29.h: three rounds of
// independent products instead of nine dependent ones).  The W - 1 additions between the windows gather the point into
// every lane, add (xyzz29_add, all four lanes compute the same sum) and spread it again.  msm_core.h's window_combine is
// the portable form of the same sum.
// called by the first four lanes of a wave; lane 0 stores
template <class LoadW>
__device__ __forceinline__ void window_combine_store(const MsmPlan& pl, LoadW load, uint8_t* out_affine, xyzz* out_xyzz) {
  const uint32_t lane = threadIdx.x & 3u;
  // windows above the scalars' top set bit are empty (64-bit scalars fill 5 of 9, {0,1} scalars one): O doubled stays
  // O, so the chain starts at the highest window that holds something
  int top = pl.W - 1;
  while (top > 0 && xyzz_is_inf(load(top))) top--;
  fe29 P = quad_spread(xyzz29_from(load(top)), lane);
#pragma unroll 1
  for (int w = top - 1; w >= 0; w--) {
#pragma unroll 1
    for (int i = 0; i < pl.c; i++) P = quad_dbl(P, lane);
    P = quad_spread(xyzz29_add(quad_gather(P), xyzz29_from(load(w))), lane);
  }
  const xyzz r = xyzz29_to(quad_gather(P));
  if (lane != 0) return;
  if (out_xyzz) st_xyzz(out_xyzz, r);
  if (out_affine) st_affine(out_affine, xyzz_to_affine_with_inv(r, fq_inv_gcd(r.ZZZ)));  // O: ZZZ = 0 -> (0, 0)
}
__global__ void __launch_bounds__(64) k_final2(const xyzz* __restrict__ wsum2, MsmPlan pl, uint8_t* __restrict__ out_affine,
                                              xyzz* __restrict__ out_xyzz) {
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  window_combine_store(pl, [&](int w) { return xyzz_add(ld_xyzz(&wsum2[2 * w]), ld_xyzz(&wsum2[2 * w + 1])); },
                       out_affine, out_xyzz);
}

// ---- 8. Horner over windows + normalise ------------------------------------------------------
__global__ void __launch_bounds__(64) k_final(const xyzz* __restrict__ wsum, MsmPlan pl, uint8_t* __restrict__ out_affine,
                        xyzz* __restrict__ out_xyzz) {
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  window_combine_store(pl, [&](int w) { return ld_xyzz(&wsum[w]); }, out_affine, out_xyzz);
}

__global__ void k_store_inf(uint8_t* out_affine, xyzz* out_xyzz) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (out_xyzz) st_xyzz(out_xyzz, xyzz_inf());
  if (out_affine) st_affine(out_affine, affine_inf());
}

// ---- direct path for small N ----------------------------------------------------------------------
// Below a few ten thousand terms the bucket method is dominated by its fixed costs (a dozen launches and
// the single-lane Horner chain of k_final), while one GLV scalar multiplication per lane costs the same
// 129 doubling steps whether 64 or 65,536 lanes run it.  One lane per term: k*P as k1*P + k2*phi(P) with
// the three affine table entries (P, phi P, P + phi P) in registers, selected per bit pair without
// divergent paths; then a shuffle/LDS tree per workgroup and one more workgroup over the partial sums.
__device__ __forceinline__ affine select3(uint32_t idx, const affine& p1, const affine& p2, const affine& p3) {
  affine r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.x.v[i] = idx == 1 ? p1.x.v[i] : (idx == 2 ? p2.x.v[i] : (idx == 3 ? p3.x.v[i] : 0u));
    r.y.v[i] = idx == 1 ? p1.y.v[i] : (idx == 2 ? p2.y.v[i] : (idx == 3 ? p3.y.v[i] : 0u));
  }
  return r;
}

struct SegN {
  uint32_t n[4];
};
// segment s = blockIdx.x / blocks_per_seg owns terms [s * stride, s * stride + n[s])
__global__ void __launch_bounds__(256) k_direct_terms(const uint8_t* __restrict__ scalars,
                                                     const uint8_t* __restrict__ points, uint32_t blocks_per_seg,
                                                     SegN sn, xyzz* __restrict__ partial) {
  __shared__ xyzz lds[4];
  uint32_t seg = blockIdx.x / blocks_per_seg;
  uint32_t local = (blockIdx.x % blocks_per_seg) * blockDim.x + threadIdx.x;
  uint32_t n = seg == 0 ? sn.n[0] : (seg == 1 ? sn.n[1] : (seg == 2 ? sn.n[2] : sn.n[3]));
  size_t i = (size_t)seg * blocks_per_seg * blockDim.x + local;
  xyzz acc = xyzz_inf();
  if (local < n) {
    u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)i * 32));
    affine P = ld_affine(points + (size_t)i * 64);
    GlvHalf h1, h2;
    glv_split(k, h1, h2);
    affine p1 = h1.neg ? affine_neg(P) : P;
    affine p2 = glv_phi(P);
    if (h2.neg) p2 = affine_neg(p2);
    affine p3 = xyzz_to_affine(xyzz_madd(xyzz_from_affine(p1), p2));  // identity stays (0, 0)
    BitReg a = bitreg_init(h1), b = bitreg_init(h2);
#pragma unroll 1
    for (int bit = 0; bit < GLV_BITS; bit++) {
      uint32_t idx = bitreg_next(a) | (bitreg_next(b) << 1);
      acc = xyzz_dbl(acc);
      acc = xyzz_madd(acc, select3(idx, p1, p2, p3));
    }
  }
  xyzz r = block_sum(acc, lds);
  if (threadIdx.x == 0) st_xyzz(&partial[blockIdx.x], r);
}

// one workgroup per segment
__global__ void __launch_bounds__(1024) k_direct_reduce(const xyzz* __restrict__ partial, uint32_t nparts,
                                                       uint8_t* __restrict__ out_affine, xyzz* __restrict__ out_xyzz) {
  __shared__ xyzz lds[16];
  const xyzz* mine = partial + (size_t)blockIdx.x * nparts;
  xyzz acc = xyzz_inf();
  for (uint32_t j = threadIdx.x; j < nparts; j += blockDim.x) acc = xyzz_add(acc, ld_xyzz(&mine[j]));
  xyzz r = block_sum(acc, lds);
  if (threadIdx.x == 0) {
    if (out_xyzz) st_xyzz(out_xyzz + blockIdx.x, r);
    if (out_affine) st_affine(out_affine + (size_t)blockIdx.x * 64, xyzz_to_affine(r));
  }
}

// longest term list of the one-chain-per-term kernels (msm_direct_multi: up to four sums in ONE launch, which is what
// the provers' L / R pairs want), and the size up to which a SINGLE sum takes that form (BP_MSM_DIRECT_MAX, default never)
static size_t msm_direct_max() { return 65536; }
static size_t msm_single_direct_max() {
  static const size_t v = [] {
    const char* e = getenv("BP_MSM_DIRECT_MAX");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0;
  }();
  return v;
}
size_t msm_direct_limit() { return msm_direct_max(); }

int msm_direct_multi(bp_ctx* ctx, const uint8_t* d_scalars, const uint8_t* d_points, uint32_t stride, const uint32_t* n,
                     int nseg, uint8_t* d_out) {
  BP_REQUIRE(nseg >= 1 && nseg <= 4 && stride % 256 == 0 && stride > 0, "msm_direct_multi: bad segment layout");
  SegN sn = {{0, 0, 0, 0}};
  for (int s2 = 0; s2 < nseg; s2++) {
    BP_REQUIRE(n[s2] <= stride, "msm_direct_multi: segment longer than the stride");
    sn.n[s2] = n[s2];
  }
  uint32_t bps = stride / 256;
  if (ctx->msm_ws.reserve((size_t)bps * nseg * sizeof(xyzz)) != 0) {
    ctx->err = "bp_msm: workspace allocation failed";
    return BP_ERR_NOMEM;
  }
  xyzz* partial = (xyzz*)ctx->msm_ws.p;
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_direct_terms, dim3(bps * nseg), dim3(256), 0, st, d_scalars, d_points, bps, sn, partial);
  hipLaunchKernelGGL(k_direct_reduce, dim3(nseg), dim3(bps > 16 ? 1024 : 64), 0, st, partial, bps, d_out,
                     (xyzz*)nullptr);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

// -------------------------------------------------------------------------------------------------
// The MSM's device workspace: the context's grow-only buffer (fixed == nullptr), or a caller-provided block of
// `fixed_bytes` that is never grown (bp_msm_dev_static: no allocation on that path).
struct MsmWsSource {
  void* fixed;
  size_t fixed_bytes;
};
static int msm_ws_get(bp_ctx* ctx, const MsmWsSource& src, size_t need, void** out) {
  if (src.fixed) {
    if (need > src.fixed_bytes) {
      ctx->err = "bp_msm_dev_static: this n / window_bits needs more than the static scratch array holds";
      return BP_ERR_ARG;
    }
    *out = src.fixed;
    return BP_OK;
  }
  if (ctx->msm_ws.reserve(need) != 0) {
    ctx->err = "bp_msm: workspace allocation failed";
    return BP_ERR_NOMEM;
  }
  *out = ctx->msm_ws.p;
  return BP_OK;
}

// Prepared points (bp_msm_points): window w reads the copies 2^(c w) P_i at d_shifted + w * stride, so every window's
// buckets carry weight 1 and the Horner chain over the windows (128 doublings in one wave) disappears.  h_scalars,
// when set, are uploaded in `nchunks` pieces on copy_stream into d_stage, each piece recoded as soon as it has landed.
struct MsmPrep {
  const uint8_t* d_shifted;
  size_t stride;
  int c;
  const uint8_t* h_scalars;
  uint8_t* d_stage;
  hipStream_t copy_stream;
  hipEvent_t* ev;
  int nchunks;
};

// Scalars that still sit in (pinned) HOST memory (bp_msm_dev_hs): the point range is cut into `parts` contiguous pieces;
// all pieces' uploads are queued on copy_stream at once, and piece i's digits, bucket sort and bucket accumulation run
// while piece i + 1 is still crossing PCIe.  Every piece fills its own bucket array (same plan, so the arrays add up
// index by index); one reduction and one Horner chain follow.
struct MsmHostScalars {
  const uint8_t* h_scalars;
  uint8_t* d_stage;        // n x 32 B
  hipStream_t copy_stream;
  hipEvent_t* ev;          // one per part
  int parts;
};

// buckets[i] += sum over the other parts' arrays
__global__ void __launch_bounds__(256) k_add_bucket_arrays(xyzz* __restrict__ buckets, uint32_t count, uint32_t parts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  xyzz acc = ld_xyzz(&buckets[i]);
  for (uint32_t p = 1; p < parts; p++) acc = xyzz_add(acc, ld_xyzz(&buckets[(size_t)p * count + i]));
  st_xyzz(&buckets[i], acc);
}

static int msm_run_ws(bp_ctx* ctx, const MsmWsSource& wsrc, const void* d_scalars, const void* d_points, size_t n,
                      int c_override, void* d_out_affine, void* d_out_xyzz, const MsmPrep* prep = nullptr,
                      const MsmHostScalars* hs = nullptr) {
  hipStream_t st = ctx->stream;
  const bool profile = ctx->profile && !wsrc.fixed;  // event creation allocates: never on the static path
  void* ws = nullptr;
  int rc;
  if (n == 0) {
    hipLaunchKernelGGL(k_store_inf, dim3(1), dim3(64), 0, st, (uint8_t*)d_out_affine, (xyzz*)d_out_xyzz);
    BP_TRY(hipGetLastError());
    return BP_OK;
  }
  if (n >= (1ull << 27)) {
    ctx->err = "bp_msm: n must be below 2^27";
    return BP_ERR_ARG;
  }
  // One GLV double-and-add chain per term (k_direct_terms): since the bucket method's tail runs on quads of lanes it is
  // no longer the faster form at any size (1.5-1.65 ms for every n <= 65,536 against 0.70-0.99 ms).  It
  // remains for a static scratch array too small for the buckets (bp_msm_dev_static) and behind BP_MSM_DIRECT_MAX.
  auto run_direct = [&]() -> int {
    uint32_t nblk = (uint32_t)((n + 255) / 256);
    int rc2;
    if ((rc2 = msm_ws_get(ctx, wsrc, (size_t)nblk * sizeof(xyzz), &ws)) != BP_OK) return rc2;
    xyzz* partial = (xyzz*)ws;
    SegN sn = {{(uint32_t)n, 0, 0, 0}};
    hipLaunchKernelGGL(k_direct_terms, dim3(nblk), dim3(256), 0, st, (const uint8_t*)d_scalars, (const uint8_t*)d_points,
                       nblk, sn, partial);
    hipLaunchKernelGGL(k_direct_reduce, dim3(1), dim3(nblk > 16 ? 1024 : 64), 0, st, partial, nblk,
                       (uint8_t*)d_out_affine, (xyzz*)d_out_xyzz);
    BP_TRY(hipGetLastError());
    return BP_OK;
  };
  const bool may_direct = !prep && !hs && c_override <= 0 && n <= msm_direct_max();
  if (may_direct && n <= msm_single_direct_max()) return run_direct();
  int c = prep ? prep->c : (c_override > 0 ? c_override : msm_choose_c(n));
  if (c < 1 || c > 16) {
    ctx->err = "bp_msm: window_bits must be in 1..16";
    return BP_ERR_ARG;
  }
  static const uint32_t glv_env = getenv("BP_MSM_NO_GLV") ? 0u : 1u;
  const uint32_t glv = prep ? 1u : glv_env;
  const size_t shift_stride = prep ? prep->stride : 0;
  if (prep) d_points = prep->d_shifted;
  MsmPlan pl = msm_plan(c, glv ? GLV_BITS : 256);
  // pieces of the point range (1 unless the scalars come from the host): `per` terms each, a multiple of 256
  const uint32_t parts = hs && hs->parts > 1 ? (uint32_t)hs->parts : 1u;
  const size_t per = parts > 1 ? ((n + parts - 1) / parts + 255) / 256 * 256 : n;
  const size_t ne = glv ? 2 * per : per;  // sorted entries per window (of one piece)
  size_t wn = (size_t)pl.W * ne;
  if (wn >= (1ull << 32)) {
    ctx->err = "bp_msm: windows * n must be below 2^32";
    return BP_ERR_ARG;
  }
  size_t nbt = (size_t)pl.W * pl.nb;
  size_t max_heavy = wn / (LIGHT_MAX + 1) + 1 + HEAVY_EXTRA;
  size_t max_slices = wn / SLICE + max_heavy + 1;

  // workspace layout (sizes computed with a null base first)
  uint32_t *codes, *ranks, *sorted, *hist, *offs, *cursor, *order, *bins, *heavy_off, *chist, *cstart, *ccur;
  uint32_t *tile_start, *tile_bin, *tbase;
  // tiles of the fine sort: one per non-empty (window, bin) plus one per FINE_TILE entries
  const size_t max_tiles = (size_t)pl.W * ((size_t)pl.nb >> FINE_BITS) + wn / FINE_TILE + 1;
  xyzz *partial, *buckets, *segv, *wsum;
  auto carve = [&](void* base) {
    Carver cv(base);
    hist = cv.take<uint32_t>(nbt);          // zeroed for each piece
    bins = cv.take<uint32_t>(2 * NBINS + 8); // zeroed for each piece (contiguous with hist)
    chist = cv.take<uint32_t>(CO_MAXCNT);    // zeroed for each piece: coarse-bin histogram of the two-level sort
    size_t zero_end = cv.off;
    cstart = cv.take<uint32_t>(CO_MAXCNT);
    ccur = cv.take<uint32_t>(CO_MAXCNT);
    tile_start = cv.take<uint32_t>(CO_MAXCNT + 1);
    tile_bin = cv.take<uint32_t>(max_tiles);
    tbase = cv.take<uint32_t>(max_tiles * FINE);
    codes = cv.take<uint32_t>(wn);
    ranks = cv.take<uint32_t>(wn);
    sorted = cv.take<uint32_t>(wn);
    offs = cv.take<uint32_t>(nbt);
    cursor = cv.take<uint32_t>(nbt);
    order = cv.take<uint32_t>(nbt);
    heavy_off = cv.take<uint32_t>(max_heavy + 1);
    partial = cv.take<xyzz>(max_slices);
    buckets = cv.take<xyzz>(nbt * parts);    // one array per piece
    {
      const size_t scan_vals = (size_t)2 * pl.W * (((size_t)pl.nb + 255) / 256);  // k_bucket_scan's bt, tb
      const size_t seg_vals = (size_t)pl.W * pl.nseg;
      segv = cv.take<xyzz>(seg_vals > scan_vals ? seg_vals : scan_vals);
    }
    wsum = cv.take<xyzz>((size_t)(1 + WSQ_GROUPS) * pl.W);  // W sums (2 W for k_bucket_sum) + the quad tail's partial sums
    (void)zero_end;
    return cv.off;
  };
  size_t need = carve(nullptr);
  if (may_direct && wsrc.fixed && need > wsrc.fixed_bytes) return run_direct();
  if ((rc = msm_ws_get(ctx, wsrc, need, &ws)) != BP_OK) return rc;
  carve(ws);
  size_t zero_bytes = (size_t)((char*)(chist + CO_MAXCNT) - (char*)hist);
  // Two-level bucket sort (GLV form, windows of at least FINE_BITS + 1 bits, entry indices below 2^24): the narrow top
  // window, if any, keeps the LDS-ranked path of k_rank_top.  BP_MSM_SORT=atomic selects the one-level atomic ranking.
  static const bool sort_atomic = getenv("BP_MSM_SORT") && !strcmp(getenv("BP_MSM_SORT"), "atomic");
  const bool top_narrow = glv && GLV_BITS + 1 - pl.c * (pl.W - 1) <= 5;
  const int wc = top_narrow ? pl.W - 1 : pl.W;
  const int nbin = pl.nb >> FINE_BITS;
  const bool two_level = glv && !sort_atomic && pl.c > FINE_BITS + 1 && ne < (1u << 24) && wc >= 1 && wc * nbin <= CO_MAXCNT;
  const uint32_t nbt32 = (uint32_t)nbt;
  const uint32_t fgrid = (uint32_t)(max_heavy < 4096 ? max_heavy : 4096);

  if (hs) {  // every piece's upload is queued now; the copy engine runs ahead of the kernels
    for (uint32_t p = 0; p < parts; p++) {
      const size_t i0 = (size_t)p * per, i1 = i0 + per < n ? i0 + per : n;
      if (i0 >= n) break;
      BP_TRY(hipMemcpyAsync(hs->d_stage + i0 * 32, hs->h_scalars + i0 * 32, (i1 - i0) * 32, hipMemcpyHostToDevice,
                            hs->copy_stream));
      BP_TRY(hipEventRecord(hs->ev[p], hs->copy_stream));
    }
    d_scalars = hs->d_stage;
  }
  uint32_t parts_run = 0;
  for (uint32_t part = 0; part < parts; part++) {
    const size_t p0 = (size_t)part * per;
    if (p0 >= n) break;
    parts_run++;
    const uint32_t n32 = (uint32_t)((p0 + per < n ? p0 + per : n) - p0), ne32 = glv ? 2 * n32 : n32;
    const uint8_t* sc_part = d_scalars ? (const uint8_t*)d_scalars + p0 * 32 : nullptr;
    const uint8_t* pt_part = prep ? (const uint8_t*)d_points : (const uint8_t*)d_points + p0 * 64;
    xyzz* bk = buckets + (size_t)part * nbt;
    if (hs) BP_TRY(hipStreamWaitEvent(st, hs->ev[part], 0));
    BP_TRY(hipMemsetAsync(hist, 0, zero_bytes, st));
    if (prep && prep->h_scalars) {
      // upload and recode chunk by chunk: the copy engine moves chunk i + 1 while the recode kernel digests chunk i
      const int nch = prep->nchunks < 1 ? 1 : prep->nchunks;
      const uint32_t cper = (uint32_t)(((n + nch - 1) / nch + 255) / 256 * 256);
      int ci = 0;
      for (uint32_t i0 = 0; i0 < n32; i0 += cper, ci++) {
        const uint32_t i1 = i0 + cper < n32 ? i0 + cper : n32;
        BP_TRY(hipMemcpyAsync(prep->d_stage + (size_t)i0 * 32, prep->h_scalars + (size_t)i0 * 32, (size_t)(i1 - i0) * 32,
                              hipMemcpyHostToDevice, prep->copy_stream));
        BP_TRY(hipEventRecord(prep->ev[ci], prep->copy_stream));
        BP_TRY(hipStreamWaitEvent(st, prep->ev[ci], 0));
        if (two_level)
          hipLaunchKernelGGL(k_recode_coarse, dim3((i1 - i0 + CO_SCALARS - 1) / CO_SCALARS), dim3(1024), 0, st,
                             (const uint8_t*)prep->d_stage, n32, i0, i1, pl, wc, nbin, codes, chist);
        else
          hipLaunchKernelGGL(k_recode_glv, dim3((i1 - i0 + 255) / 256), dim3(256), 0, st, (const uint8_t*)prep->d_stage, n32, i0,
                             i1, pl, codes, ranks, hist);
      }
    } else if (two_level)
      hipLaunchKernelGGL(k_recode_coarse, dim3((n32 + CO_SCALARS - 1) / CO_SCALARS), dim3(1024), 0, st, sc_part, n32, 0u, n32,
                         pl, wc, nbin, codes, chist);
    else if (glv)
      hipLaunchKernelGGL(k_recode_glv, dim3((n32 + 255) / 256), dim3(256), 0, st, sc_part, n32, 0u, n32, pl, codes, ranks, hist);
    if (glv && GLV_BITS + 1 - pl.c * (pl.W - 1) <= 5)
      hipLaunchKernelGGL(k_rank_top, dim3((ne32 + 256 * RT_PER - 1) / (256 * RT_PER)), dim3(256), 0, st, ne32, pl, codes, ranks, hist);
    if (!glv) hipLaunchKernelGGL(k_recode, dim3((n32 + 255) / 256), dim3(256), 0, st, sc_part, n32, pl, codes, ranks, hist);
    if (two_level) {
      // windows [0, wc): coarse partition by the high bucket bits, then the fine sort over tiles of every coarse bin
      const uint32_t ncnt = (uint32_t)(wc * nbin);
      // tiles that can exist: every bin rounds up once, on top of (entries of the windows) / FINE_TILE
      const uint32_t tile_grid = (uint32_t)(ncnt + ((size_t)wc * ne32) / FINE_TILE);
      hipLaunchKernelGGL(k_coarse_scan, dim3(1), dim3(1024), 0, st, chist, ncnt, cstart, ccur, tile_start, tile_bin,
                         (uint32_t)max_tiles);
      hipLaunchKernelGGL(k_scatter_coarse, dim3((ne32 + CO_TILE - 1) / CO_TILE), dim3(1024), 0, st, codes, ne32, pl, wc, nbin,
                         ccur, ranks /* coarse entries */);
      hipLaunchKernelGGL(k_fine_count, dim3(tile_grid), dim3(256), 0, st, ranks, cstart, chist, tile_start, tile_bin, ncnt, pl,
                         nbin, hist, tbase);
      hipLaunchKernelGGL(k_fine_place, dim3(tile_grid), dim3(256), 0, st, ranks, cstart, chist, tile_start, tile_bin, ncnt, pl,
                         nbin, hist, tbase, offs, sorted);
      if (wc < pl.W) {  // the narrow top window: ranks from k_rank_top, one scan and scatter for it alone
        hipLaunchKernelGGL(k_scan, dim3(pl.W - wc), dim3(1024), 0, st, hist, ne32, pl, wc, offs, cursor);
        hipLaunchKernelGGL(k_scatter, dim3((ne32 + 255) / 256, pl.W - wc), dim3(256), 0, st, codes, ranks, ne32, pl, wc, offs,
                           sorted);
      }
    } else {
      hipLaunchKernelGGL(k_scan, dim3(pl.W), dim3(1024), 0, st, hist, ne32, pl, 0, offs, cursor);
      hipLaunchKernelGGL(k_scatter, dim3((ne32 + 255) / 256, pl.W), dim3(256), 0, st, codes, ranks, ne32, pl, 0, offs, sorted);
    }
    hipLaunchKernelGGL(k_cls_hist, dim3((nbt32 + 1023) / 1024), dim3(1024), 0, st, hist, nbt32, bins);
    hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(64), 0, st, bins);
    hipLaunchKernelGGL(k_cls_scatter, dim3((nbt32 + 1023) / 1024), dim3(1024), 0, st, hist, nbt32, bins, order);
    const bool prof_here = profile && part == 0;  // one event pair per call: the first piece's accumulation
    if (prof_here) {  // the previous call's pair has completed on this in-order stream or is read after a sync
      if (!ctx->msm_ev[0]) {
        BP_TRY(hipEventCreate(&ctx->msm_ev[0]));
        BP_TRY(hipEventCreate(&ctx->msm_ev[1]));
      } else if (ctx->prof_pending_light) {
        float ms = 0;
        if (hipEventSynchronize(ctx->msm_ev[1]) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->msm_ev[0], ctx->msm_ev[1]) == hipSuccess) {
          ctx->prof_ms[1] += ms;
          ctx->prof_calls[1] += 1;
        }
      }
    }
    hipLaunchKernelGGL(k_heavy_plan, dim3(1), dim3(1024), 0, st, hist, order, bins, heavy_off);
    if (prof_here) BP_TRY(hipEventRecord(ctx->msm_ev[0], st));
    {
      const uint32_t light_blocks = (nbt32 + 255) / 256;
      const uint32_t hgrid = (uint32_t)(max_slices < 1024 ? max_slices : 1024);
      hipLaunchKernelGGL(k_accumulate, dim3(hgrid + light_blocks), dim3(256), 0, st, pt_part, hist, offs, sorted, order, bins,
                         heavy_off, nbt32, glv, (uint32_t)pl.nb, shift_stride, hgrid, bk, partial);
    }
    if (prof_here) {
      BP_TRY(hipEventRecord(ctx->msm_ev[1], st));
      ctx->prof_pending_light = true;
      ctx->prof_madds[1] += (double)pl.W * ne32;
    }
    hipLaunchKernelGGL(k_heavy_final, dim3(fgrid), dim3(64), 0, st, order, bins, heavy_off, partial, bk);
  }
  if (parts_run > 1)
    hipLaunchKernelGGL(k_add_bucket_arrays, dim3((nbt32 + 255) / 256), dim3(256), 0, st, buckets, nbt32, parts_run);
  if (prep) {  // all windows weigh 1: add their buckets index by index, then reduce ONE window; no Horner chain
    hipLaunchKernelGGL(k_combine_windows, dim3((pl.nb + 255) / 256), dim3(256), 0, st, buckets, (uint32_t)pl.W, (uint32_t)pl.nb);
    pl.W = 1;
  }
  static const bool tail_segments = getenv("BP_MSM_TAIL") && !strcmp(getenv("BP_MSM_TAIL"), "segments");
  // the scan does ~17 additions per bucket against ~2 for the running sums: it pays while the launch stays within
  // about a wave per SIMD (latency-bound), i.e. for one combined window or small plans
  if (pl.nb <= 65536 && (size_t)pl.W * pl.nb <= 65536 && !tail_segments) {
    const uint32_t nblk = ((uint32_t)pl.nb + 255u) / 256u;  // <= 256
    xyzz* bt = segv;  // [W][nblk] twice (the carve sizes segv for both tails)
    xyzz* tb = segv + (size_t)pl.W * nblk;
    hipLaunchKernelGGL(k_bucket_scan, dim3(nblk, pl.W), dim3(256), 0, st, buckets, (uint32_t)pl.nb, nblk, bt, tb);
    hipLaunchKernelGGL(k_bucket_sum, dim3(pl.W, 2), dim3(256), 0, st, bt, tb, nblk, wsum);
    hipLaunchKernelGGL(k_final2, dim3(1), dim3(64), 0, st, wsum, pl, (uint8_t*)d_out_affine, (xyzz*)d_out_xyzz);
    BP_TRY(hipGetLastError());
    return BP_OK;
  }
  uint32_t nsegt = (uint32_t)pl.W * pl.nseg;
  static const bool seg_quad = !(getenv("BP_MSM_SEG_QUAD") && atoi(getenv("BP_MSM_SEG_QUAD")) == 0);
  if (seg_quad && nsegt <= 16384 && pl.nseg > 1) {
    int seg_log = 0, idx_bits = 1;
    while ((1 << seg_log) < pl.seg) seg_log++;
    while ((1 << idx_bits) < pl.nseg) idx_bits++;
    hipLaunchKernelGGL(k_seg_reduce_quad, dim3((nsegt * 4 + 255) / 256), dim3(256), 0, st, buckets, pl, seg_log, idx_bits, segv);
  } else
    hipLaunchKernelGGL(k_seg_reduce, dim3((nsegt + 255) / 256), dim3(256), 0, st, buckets, pl, segv);
  // window sums: 16 slices per window by quads, then the 16 partial sums (0.04 ms against 0.22 ms for one workgroup of
  // single lanes per window).  BP_MSM_QUAD=0 keeps the one-lane kernel.
  static const bool tail_quad = !(getenv("BP_MSM_QUAD") && atoi(getenv("BP_MSM_QUAD")) == 0);
  if (tail_quad && pl.nseg >= 64) {
    xyzz* part = wsum + pl.W;  // [W][WSQ_GROUPS]: wsum is carved with room behind the W sums
    hipLaunchKernelGGL(k_window_sum_quad, dim3(WSQ_GROUPS, pl.W), dim3(256), 0, st, segv, (uint32_t)pl.nseg, WSQ_GROUPS, part);
    hipLaunchKernelGGL(k_window_sum_quad, dim3(1, pl.W), dim3(256), 0, st, part, WSQ_GROUPS, 1u, wsum);
  } else {
    hipLaunchKernelGGL(k_window_sum, dim3(pl.W), dim3(256), 0, st, segv, pl, wsum);
  }
  hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, st, wsum, pl, (uint8_t*)d_out_affine, (xyzz*)d_out_xyzz);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int msm_run(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int c_override, void* d_out_affine,
            void* d_out_xyzz) {
  return msm_run_ws(ctx, MsmWsSource{nullptr, 0}, d_scalars, d_points, n, c_override, d_out_affine, d_out_xyzz);
}

// ---- scalars from host memory (include/bpgpu.h: bp_msm_dev_hs) ---------------------------------------------------------
// Pieces of the point range.  32 MB of scalars take 0.63 ms to cross PCIe at N = 2^20 - a third of the sum's kernel time -
// and nothing but the digit recoding (0.09 ms) can start before the LAST byte has landed unless the sum is cut by point
// range, so that piece i's sort and accumulation run while piece i + 1 is in flight.  Measured on MI355X (bench.py sweep,
// uniform scalars, ms at 2^18 / 2^19 / 2^20): 1 piece 1.36 / 1.85 / 2.79, 2 pieces 1.55 / 1.98 / 4.21, 3 pieces 1.96 /
// 2.16 / 3.06, 4 pieces 3.34 / 2.39 / 3.14.  Copies that run beside kernels and the event hand-offs between the copy
// stream and the compute stream cost more than the overlap returns (the same was found for the prepared-point upload
// and for device-to-host copies in the verifier), so ONE piece is the default: the upload is then a
// plain copy ahead of the kernels on the copy stream.  BP_MSM_HS_PARTS=2..8 selects the piecewise form (read per
// call; tests/test_gpu_msm.py runs both).
static int msm_hs_parts(size_t n) {
  const char* e = getenv("BP_MSM_HS_PARTS");
  const int v = e ? atoi(e) : 1;
  return (v >= 2 && v <= 8 && n >= 4096) ? v : 1;
}

int msm_run_host_scalars(bp_ctx* ctx, const uint8_t* h_scalars, const void* d_points, size_t n, int c_override,
                         void* d_out_affine) {
  if (n == 0) return msm_run(ctx, nullptr, nullptr, 0, c_override, d_out_affine, nullptr);
  const int parts = msm_hs_parts(n);
  if (ctx->msm_stage.bytes < n * 32) {
    BP_TRY(hipStreamSynchronize(ctx->stream));  // a regrowth frees the old buffer
    if (ctx->msm_copy_stream) BP_TRY(hipStreamSynchronize(ctx->msm_copy_stream));
    if (ctx->msm_stage.reserve(n * 32) != 0) {
      ctx->err = "bp_msm_dev_hs: staging allocation failed";
      return BP_ERR_NOMEM;
    }
  }
  if (parts == 1) {
    // one piece: the copy is simply the first command of the sum on the context's own stream - no second stream, no
    // event hand-offs (they cost ~0.1 ms per call, measured) - and the stream's order protects the staging buffer
    BP_TRY(hipMemcpyAsync(ctx->msm_stage.p, h_scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return msm_run_ws(ctx, MsmWsSource{nullptr, 0}, ctx->msm_stage.p, d_points, n, c_override, d_out_affine, nullptr);
  }
  if (!ctx->msm_copy_stream) {
    BP_TRY(hipStreamCreateWithFlags(&ctx->msm_copy_stream, hipStreamNonBlocking));
    for (auto& e : ctx->msm_copy_ev) BP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BP_TRY(hipEventCreateWithFlags(&ctx->msm_stage_free, hipEventDisableTiming));
  }
  // the previous call's kernels may still read the staging buffer: the copies queue behind them
  BP_TRY(hipEventRecord(ctx->msm_stage_free, ctx->stream));
  BP_TRY(hipStreamWaitEvent(ctx->msm_copy_stream, ctx->msm_stage_free, 0));
  MsmHostScalars hs;
  hs.h_scalars = h_scalars;
  hs.d_stage = (uint8_t*)ctx->msm_stage.p;
  hs.copy_stream = ctx->msm_copy_stream;
  hs.ev = ctx->msm_copy_ev;
  hs.parts = parts;
  return msm_run_ws(ctx, MsmWsSource{nullptr, 0}, nullptr, d_points, n, c_override, d_out_affine, nullptr, nullptr, &hs);
}

void msm_host_scalars_release(bp_ctx* ctx) {
  if (ctx->msm_copy_stream) {
    (void)hipStreamSynchronize(ctx->msm_copy_stream);
    (void)hipStreamDestroy(ctx->msm_copy_stream);
    ctx->msm_copy_stream = nullptr;
  }
  for (auto& e : ctx->msm_copy_ev)
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  if (ctx->msm_stage_free) (void)hipEventDestroy(ctx->msm_stage_free);
  ctx->msm_stage_free = nullptr;
  ctx->msm_stage.release();
}

// ---- the entry point that allocates nothing (include/bpgpu.h: bp_msm_dev_static) -------------------------------
// Its workspace is this file-scope array: every GPU the library runs on holds one copy until the process ends.  In
// host code the NAME `scratch` is only a handle for hipGetSymbolAddress - never pass it to a kernel or a memset; the
// per-device address comes from msm_static_init (called by bp_ctx_create with the device current) and is what the
// kernels receive, exactly as they receive ctx->msm_ws.p on the allocating path.
constexpr size_t MSM_STATIC_BYTES = (size_t)32 << 20;
static __device__ __align__(256) unsigned char scratch[MSM_STATIC_BYTES];
constexpr int MSM_STATIC_MAX_DEV = 64;
static void* g_scratch_ptr[MSM_STATIC_MAX_DEV];  // per device: each GPU has its own copy at its own address
static std::mutex g_scratch_mu;

int msm_static_init(int device) {
  if (device < 0 || device >= MSM_STATIC_MAX_DEV) return BP_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return BP_OK;
  void* ptr = nullptr;
  if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return BP_ERR_DEVICE;
  g_scratch_ptr[device] = ptr;
  return BP_OK;
}

size_t msm_static_bytes() { return MSM_STATIC_BYTES; }

int msm_run_static(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int c_override, void* d_out) {
  void* ptr = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (ctx->device >= 0 && ctx->device < MSM_STATIC_MAX_DEV) ptr = g_scratch_ptr[ctx->device];
  }
  if (!ptr) {
    ctx->err = "bp_msm_dev_static: the scratch array of this device was not resolved at bp_ctx_create";
    return BP_ERR_STATE;
  }
  return msm_run_ws(ctx, MsmWsSource{ptr, MSM_STATIC_BYTES}, d_scalars, d_points, n, c_override, d_out, nullptr);
}

// ---- bp_msm_points: a fixed point set prepared once (include/bpgpu.h) ---------------------------------------------
struct MsmPoints {
  bp_ctx* ctx;
  size_t n;
  int c, W;
  DevBuf shifted;   // [W][n] affine copies 2^(c w) P_i
  DevBuf ws;        // the bucket method's workspace for this n (sized once)
  DevBuf stage;     // n scalars uploaded from the host
  hipStream_t copy_stream;
  hipEvent_t ev[8];
  hipEvent_t ev_start;  // orders a new upload behind the previous call's readers of the staging buffer
  uint8_t* d_out;   // 64 B result slot
};

int msm_points_create(bp_ctx* ctx, const uint8_t* h_points, const void* d_points, size_t n, MsmPoints** out) {
  *out = nullptr;
  if (n < 1 || n >= (1ull << 24)) {
    ctx->err = "bp_msm_points_create: n must be in 1..2^24";
    return BP_ERR_ARG;
  }
  MsmPoints* mp = new MsmPoints();
  mp->ctx = ctx;
  mp->n = n;
  mp->c = msm_choose_c(n);
  MsmPlan pl = msm_plan(mp->c, GLV_BITS);
  mp->W = pl.W;
  mp->copy_stream = nullptr;
  for (auto& e : mp->ev) e = nullptr;
  mp->ev_start = nullptr;
  auto fail = [&](int rc, const char* msg) {
    ctx->err = msg;
    mp->shifted.release();
    mp->ws.release();
    mp->stage.release();
    for (auto& e : mp->ev)
      if (e) (void)hipEventDestroy(e);
    if (mp->ev_start) (void)hipEventDestroy(mp->ev_start);
    if (mp->copy_stream) (void)hipStreamDestroy(mp->copy_stream);
    delete mp;
    return rc;
  };
  const size_t copies = (size_t)pl.W * n;
  if (mp->shifted.reserve(copies * 64) != 0 || mp->stage.reserve(n * 32 + 256) != 0)
    return fail(BP_ERR_NOMEM, "bp_msm_points_create: allocation failed");
  DevBuf tmp;  // XYZZ scratch of the shift kernel (+ the uploaded points)
  if (tmp.reserve(copies * sizeof(xyzz) + n * 64) != 0) return fail(BP_ERR_NOMEM, "bp_msm_points_create: scratch allocation failed");
  hipStream_t st = ctx->stream;
  uint8_t* d_in = (uint8_t*)tmp.p + copies * sizeof(xyzz);
  hipError_t e = hipMemcpyAsync(d_in, h_points ? (const void*)h_points : d_points, n * 64,
                                h_points ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_msm_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, (uint32_t)n, mp->c, mp->W,
                       (xyzz*)tmp.p, (uint8_t*)mp->shifted.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  tmp.release();
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&mp->copy_stream, hipStreamNonBlocking);
  for (auto& ev : mp->ev)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&mp->ev_start, hipEventDisableTiming);
  if (e != hipSuccess) return fail(BP_ERR_DEVICE, "bp_msm_points_create: device error while preparing the points");
  mp->d_out = (uint8_t*)mp->stage.p + n * 32;
  *out = mp;
  return BP_OK;
}

void msm_points_destroy(MsmPoints* mp) {
  if (!mp) return;
  (void)hipSetDevice(mp->ctx->device);
  (void)hipStreamSynchronize(mp->ctx->stream);
  (void)hipStreamSynchronize(mp->copy_stream);
  mp->shifted.release();
  mp->ws.release();
  mp->stage.release();
  for (auto& e : mp->ev)
    if (e) (void)hipEventDestroy(e);
  if (mp->ev_start) (void)hipEventDestroy(mp->ev_start);
  (void)hipStreamDestroy(mp->copy_stream);
  delete mp;
}

// d_scalars (device) or h_scalars (host; pinned memory overlaps the upload with the recoding); result to d_out (device)
// and/or h_out (host, synchronous).
int msm_points_run(MsmPoints* mp, const void* d_scalars, const uint8_t* h_scalars, void* d_out, uint8_t* h_out) {
  bp_ctx* ctx = mp->ctx;
  MsmPrep pr;
  pr.d_shifted = (const uint8_t*)mp->shifted.p;
  pr.stride = mp->n * 64;
  pr.c = mp->c;
  pr.h_scalars = h_scalars;
  pr.d_stage = (uint8_t*)mp->stage.p;
  pr.copy_stream = mp->copy_stream;
  pr.ev = mp->ev;
  static const int env_chunks = [] {
    const char* e = getenv("BP_MSM_UPLOAD_CHUNKS");
    int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 0;
  }();
  // One piece by default: since the two-level sort the recoding kernel takes ~0.08 ms at 2^20, so there is little left to
  // hide the copy behind, and the per-piece events cost more than they save (measured inside bench.py: 4.66 ms with 8
  // pieces against 2.94 ms with one).  BP_MSM_UPLOAD_CHUNKS=2..8 still selects the pipelined form.
  pr.nchunks = env_chunks ? env_chunks : 1;
  if (h_scalars) {  // the previous call's recode kernels may still read the staging buffer
    BP_TRY(hipEventRecord(mp->ev_start, ctx->stream));
    BP_TRY(hipStreamWaitEvent(mp->copy_stream, mp->ev_start, 0));
  }
  // the handle owns its workspace: grow-only, sized by the first call
  struct Own {
    bp_ctx* ctx;
    DevBuf saved;
    DevBuf& mine;
    Own(bp_ctx* c, DevBuf& m) : ctx(c), saved(c->msm_ws), mine(m) { ctx->msm_ws = mine; }
    ~Own() {
      mine = ctx->msm_ws;
      ctx->msm_ws = saved;
    }
  } own(ctx, mp->ws);
  void* out = d_out ? d_out : (void*)mp->d_out;
  int rc = msm_run_ws(ctx, MsmWsSource{nullptr, 0}, d_scalars, nullptr, mp->n, 0, out, nullptr, &pr);
  if (rc != BP_OK) return rc;
  if (h_out) {
    BP_TRY(hipMemcpyAsync(h_out, out, 64, hipMemcpyDeviceToHost, ctx->stream));
    BP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return BP_OK;
}

}  // namespace bp
