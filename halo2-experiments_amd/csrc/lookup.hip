// lookup.hip -- the lookup argument's permuted columns (upstream halo2_proofs plonk/lookup/prover.rs:
// permute_expression_pair, at the tag pinned by /root/reference/Cargo.toml:10; reached from the reference through
// create_proof, /root/reference/src/circuits/utils.rs:40-48 -- the MerkleSumTree circuit's range checks are lookups).
//
// Upstream, per lookup: A' = the compressed input column's usable rows sorted by the field's `Ord` (the canonical
// integer); a BTreeMap counts the table's values; walking A', the first occurrence of every value puts that value into
// S' at the same row and takes one instance out of the map (a value missing from the table is the error
// ConstraintSystemFailure); the leftover table values, in ascending order, then fill the rows of repeated inputs
// from the LAST such row backwards.  The result is a deterministic function of the two multisets, so the device form
// only has to reproduce it:
//   1  keys      Montgomery words -> canonical integers (one product by 2^-256), padded to a power of two with keys above r
//   2  sort      bitonic network on 256-bit keys: stages with a partner inside a 2048-key tile run in LDS, the others
//                are one launch per stage -- (log n)(log n + 1)/2 compare-exchange stages, independent of the values
//   3  mark      per row of A': first occurrence?  then its value's first position in the sorted table (binary search)
//                is marked used -- distinct values mark distinct positions; not found raises the error flag
//   4  compact   exclusive scans of "repeated row" and "table position unused" -> the two lists upstream walks
//   5  fill      S'[row] = A'[row] at first occurrences, else the (m - 1 - rank)-th leftover value; back to Montgomery
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "g1.h"
#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

constexpr int LK_THREADS = 256;
constexpr uint32_t LK_TILE_LOG = 11, LK_TILE = 1u << LK_TILE_LOG;     // keys per LDS tile: 2048 x 32 B = 64 KiB

struct Key {
  uint32_t w[8];
};

__device__ __forceinline__ Key key_load(const uint32_t* p, uint64_t i) {
  const uint4* q = reinterpret_cast<const uint4*>(p + i * 8);
  const uint4 a = q[0], b = q[1];
  return Key{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void key_store(uint32_t* p, uint64_t i, const Key& k) {
  uint4* q = reinterpret_cast<uint4*>(p + i * 8);
  q[0] = make_uint4(k.w[0], k.w[1], k.w[2], k.w[3]);
  q[1] = make_uint4(k.w[4], k.w[5], k.w[6], k.w[7]);
}
// -1 / 0 / +1 as a < / == / > b (little-endian words: the top word decides first)
__device__ __forceinline__ int key_cmp(const Key& a, const Key& b) {
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    if (a.w[k] != b.w[k]) return a.w[k] < b.w[k] ? -1 : 1;
  }
  return 0;
}

struct LkFr {
  uint32_t l[9];
};

// A batch of lookups: pair p has its columns in ptr.in[2p] (input) and ptr.in[2p + 1] (table), its outputs in
// ptr.out[2p] / ptr.out[2p + 1]; by value, LK_MAX_PAIRS pairs per launch chain
constexpr int LK_MAX_PAIRS = 8;
// the key-bit words of lk_convert_kernel: per key array LK_KB_SLOTS slots of (lo, hi), one 64-byte line each (a slot is shared by
// the workgroups with blockIdx.x % LK_KB_SLOTS == slot: no word is hot), OR-ed on the host
constexpr uint32_t LK_KB_SLOTS = 32, LK_KB_STRIDE = 16;
constexpr size_t LK_KB_WORDS = (size_t)2 * LK_MAX_PAIRS * LK_KB_SLOTS * LK_KB_STRIDE;
struct LkPtrs {
  const uint32_t* in[2 * LK_MAX_PAIRS];
  uint32_t* out[2 * LK_MAX_PAIRS];
};

// Montgomery words -> canonical integers: one product by `c` (2^-256 in the form the raw words need); rows >= live get
// the all-ones key (above r: sorts behind every field element).  grid.y = key array (2 per pair).
// kbits slots of array a (LK_KB_SLOTS) |= word 0 of its live keys, resp. |= their words 1 .. 7: the host picks the sort by them (one OR
// per wave, not per key).
__global__ __launch_bounds__(LK_THREADS) void lk_convert_kernel(LkPtrs ptr, uint32_t* keys, uint64_t pitch, uint64_t live, uint64_t total, LkFr c,
                                                                uint32_t* __restrict__ kbits) {
  const uint64_t i = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;
  uint32_t lo_bits = 0, hi_bits = 0;
  if (i < total) {
    uint32_t* out = keys + (size_t)blockIdx.y * pitch;
    if (i >= live) {
      key_store(out, i, Key{{~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    } else {
      const Key k = key_load(ptr.in[blockIdx.y], i);
      Fr cc;
#pragma unroll
      for (int j = 0; j < 9; ++j) cc.l[j] = c.l[j];
      HM_DECLARE(cc, 1.0);
      const Fr y = fe_canonical(fe_mul(fe_unpack<FrParams>(k.w), cc));
      Key o;
      fe_pack(o.w, y);
      key_store(out, i, o);
      lo_bits = o.w[0];
      hi_bits = o.w[1] | o.w[2] | o.w[3] | o.w[4] | o.w[5] | o.w[6] | o.w[7];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_bits |= __shfl_xor(lo_bits, off, 64);
    hi_bits |= __shfl_xor(hi_bits, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    // OR is monotone: a wave whose bits are already in its slot skips the atomic (a stale read only costs an atomic that was not
    // needed).  With ONE pair of words per array every wave of the launch hit the same line: 131 000 serialised atomics at
    // 2^18 rows x 16 arrays were 0.6 of the kernel's 0.63 ms, and the plain reads of the test alone still 0.2.
    uint32_t* slot = kbits + ((size_t)blockIdx.y * LK_KB_SLOTS + blockIdx.x % LK_KB_SLOTS) * LK_KB_STRIDE;
    const uint32_t cur_lo = __atomic_load_n(slot, __ATOMIC_RELAXED), cur_hi = __atomic_load_n(slot + 1, __ATOMIC_RELAXED);
    if (lo_bits & ~cur_lo) atomicOr(slot, lo_bits);
    if (hi_bits & ~cur_hi) atomicOr(slot + 1, hi_bits);
  }
}

// ---------------------------------------------------------------------------------------------
// Sorting SMALL keys (every live key of the chain below 2^B, B <= LK_COUNT_BITS: range-check and byte-table lookups -- the
// reference's MerkleSumTree lookups are those, /root/reference/src/chips/merkle_sum_tree.rs -- hold values of 8 .. 16 bits)
// needs no comparison network: the keys carry no payload, so the sorted array is "value v repeated count[v] times".  Three
// launches whatever n: histogram (one atomic per distinct value and wave), exclusive scan of the 2^B counters, expansion (every
// output position finds its value by a binary search over the scanned counters).  The 256-bit bitonic network costs
// (log n)(log n + 1) / 2 stages whatever the values.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t LK_COUNT_BITS = 20;           // at most 2^20 counters (4 MiB) per key array
__global__ __launch_bounds__(LK_THREADS) void lk_count_kernel(const uint32_t* __restrict__ keys, uint64_t pitch, uint64_t live,
                                                              uint32_t* __restrict__ counters, uint32_t nbins) {
  const uint64_t i = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;
  keys += (size_t)blockIdx.y * pitch;
  counters += (size_t)blockIdx.y * nbins;
  const bool on = i < live;
  const uint32_t v = on ? keys[i * 8] : 0u;
  uint64_t todo = __ballot(on);
  while (todo) {                                   // one atomic per distinct value of the wave (constant columns: one per wave)
    const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
    const uint32_t lv = (uint32_t)__builtin_amdgcn_readlane((int)v, leader);
    const uint64_t same = __ballot(on && v == lv) & todo;
    if ((threadIdx.x & 63) == leader) atomicAdd(counters + lv, (uint32_t)__popcll(same));
    todo &= ~same;
  }
}
// exclusive scan of one array's counters in place (one workgroup of 1024 lanes per array)
__global__ __launch_bounds__(1024) void lk_count_scan_kernel(uint32_t* __restrict__ counters, uint32_t nbins) {
  __shared__ uint32_t s[1024];
  counters += (size_t)blockIdx.x * nbins;
  const uint32_t t = threadIdx.x, per = (nbins + 1023) / 1024;
  const uint32_t lo = t * per < nbins ? t * per : nbins, hi = lo + per < nbins ? lo + per : nbins;
  uint32_t sum = 0;
  for (uint32_t b = lo; b < hi; ++b) sum += counters[b];
  s[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t a = t >= off ? s[t - off] : 0u;
    __syncthreads();
    s[t] += a;
    __syncthreads();
  }
  uint32_t run = s[t] - sum;
  for (uint32_t b = lo; b < hi; ++b) {
    const uint32_t c = counters[b];
    counters[b] = run;
    run += c;
  }
}
// keys[i] = the largest v with start[v] <= i for i < live (empty values share their successor's start: the LAST of equal
// starts is the value that owns position i), the all-ones padding key behind
__global__ __launch_bounds__(LK_THREADS) void lk_count_expand_kernel(uint32_t* __restrict__ keys, uint64_t pitch, uint64_t live, uint64_t total,
                                                                     const uint32_t* __restrict__ start, uint32_t nbins) {
  const uint64_t i = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;
  if (i >= total) return;
  keys += (size_t)blockIdx.y * pitch;
  start += (size_t)blockIdx.y * nbins;
  if (i >= live) {
    key_store(keys, i, Key{{~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    return;
  }
  uint32_t lo = 0, hi = nbins;                     // invariant: start[lo] <= i, and (hi == nbins or start[hi] > i)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (start[mid] <= (uint32_t)i) lo = mid; else hi = mid;
  }
  key_store(keys, i, Key{{lo, 0u, 0u, 0u, 0u, 0u, 0u, 0u}});
}

// one compare-exchange stage of the bitonic network with partner distance j >= LK_TILE (phase length k); blockIdx.y
// selects one of the arrays sorted side by side (the input and the table column: `pitch` words apart)
__device__ __forceinline__ void lk_cmpx(Key& a, Key& b, bool asc) {
  const int c = key_cmp(a, b);
  if (asc ? c > 0 : c < 0) {
    const Key t = a;
    a = b;
    b = t;
  }
}
__global__ __launch_bounds__(LK_THREADS) void lk_sort_global_kernel(uint32_t* keys, uint64_t n, uint64_t k, uint64_t j, uint64_t pitch) {
  keys += (size_t)blockIdx.y * pitch;
  const uint64_t t = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;      // one lane per pair
  if (t >= n / 2) return;
  const uint64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
  Key a = key_load(keys, i), b = key_load(keys, l);
  const Key a0 = a;
  lk_cmpx(a, b, (i & k) == 0);
  if (key_cmp(a, a0) != 0) {
    key_store(keys, i, a);
    key_store(keys, l, b);
  }
}
// two consecutive stages (distances j and j / 2, both >= LK_TILE) in one launch: a lane owns the four keys that differ
// in those two index bits; the direction bit (i & k) is the same for all four
__global__ __launch_bounds__(LK_THREADS) void lk_sort_global2_kernel(uint32_t* keys, uint64_t n, uint64_t k, uint64_t j, uint64_t pitch) {
  keys += (size_t)blockIdx.y * pitch;
  const uint64_t t = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;      // one lane per quadruple
  if (t >= n / 4) return;
  const uint64_t j2 = j >> 1;
  const uint64_t i0 = ((t & ~(j2 - 1)) << 2) | (t & (j2 - 1));
  Key q0 = key_load(keys, i0), q1 = key_load(keys, i0 | j2), q2 = key_load(keys, i0 | j), q3 = key_load(keys, i0 | j | j2);
  const bool asc = (i0 & k) == 0;
  lk_cmpx(q0, q2, asc);
  lk_cmpx(q1, q3, asc);
  lk_cmpx(q0, q1, asc);
  lk_cmpx(q2, q3, asc);
  key_store(keys, i0, q0);
  key_store(keys, i0 | j2, q1);
  key_store(keys, i0 | j, q2);
  key_store(keys, i0 | j | j2, q3);
}

// every stage with partner distance < LK_TILE of the phases k_first .. k_last (k_first = 2: the whole sort of a tile),
// for one tile in LDS (structure of arrays: word w of key e at lds[w * LK_TILE + e])
__global__ __launch_bounds__(LK_THREADS) void lk_sort_tile_kernel(uint32_t* keys, uint64_t n, uint64_t k_first, uint64_t k_last, uint64_t pitch) {
  extern __shared__ uint32_t lds[];
  keys += (size_t)blockIdx.y * pitch;
  const uint64_t base = (uint64_t)blockIdx.x * LK_TILE;
  const uint32_t tile = n < LK_TILE ? (uint32_t)n : LK_TILE;
  for (uint32_t e = threadIdx.x; e < tile; e += LK_THREADS) {
    const Key q = key_load(keys, base + e);
#pragma unroll
    for (int w = 0; w < 8; ++w) lds[w * LK_TILE + e] = q.w[w];
  }
  __syncthreads();
  for (uint64_t k = k_first; k <= k_last; k <<= 1) {
    uint32_t j = (uint32_t)(k / 2 < tile ? k / 2 : tile / 2);
    for (; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < tile / 2; t += LK_THREADS) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        Key a, b;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          a.w[w] = lds[w * LK_TILE + i];
          b.w[w] = lds[w * LK_TILE + l];
        }
        const bool asc = ((base + i) & k) == 0;
        const int c = key_cmp(a, b);
        if (asc ? c > 0 : c < 0) {
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            lds[w * LK_TILE + i] = b.w[w];
            lds[w * LK_TILE + l] = a.w[w];
          }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t e = threadIdx.x; e < tile; e += LK_THREADS) {
    Key q;
#pragma unroll
    for (int w = 0; w < 8; ++w) q.w[w] = lds[w * LK_TILE + e];
    key_store(keys, base + e, q);
  }
}

// Per-pair work arrays: every array below holds `stride` words per pair (stride >= rows), the block sums `sblocks` per pair.
// flags[i] = 1 for a repeated input row (same value as the row before), 0 for a first occurrence; first occurrences mark
// used[lower_bound(table, value)] = 1, or raise the pair's `missing` word when the table does not hold the value
__global__ __launch_bounds__(LK_THREADS) void lk_mark_kernel(const uint32_t* keys, uint64_t pitch, uint64_t rows, uint64_t stride,
                                                             uint32_t* repeated, uint32_t* used, uint32_t* small) {
  const uint32_t p = blockIdx.y;
  const uint32_t *a_sorted = keys + (size_t)(2 * p) * pitch, *t_sorted = keys + (size_t)(2 * p + 1) * pitch;
  repeated += (size_t)p * stride;
  used += (size_t)p * stride;
  const uint64_t i = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;
  if (i >= rows) return;
  const Key v = key_load(a_sorted, i);
  const bool first = i == 0 || key_cmp(v, key_load(a_sorted, i - 1)) != 0;
  repeated[i] = first ? 0u : 1u;
  if (!first) return;
  uint64_t lo = 0, hi = rows;                       // lower_bound: first position with key >= v
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (key_cmp(key_load(t_sorted, mid), v) < 0) lo = mid + 1;
    else hi = mid;
  }
  if (lo < rows && key_cmp(key_load(t_sorted, lo), v) == 0) used[lo] = 1u;
  else atomicExch(small + p * 4 + 2, 1u);
}

// exclusive scan of 0/1 flags (optionally inverted), three launches: block sums, one-block scan of those, apply; grid.y = pair
constexpr int SC_ITEMS = 8;
constexpr int SC_BLOCK = LK_THREADS * SC_ITEMS;
__device__ __forceinline__ uint32_t lk_block_scan(uint32_t v, uint32_t* s, uint32_t& total) {   // exclusive, 256 lanes
  const uint32_t t = threadIdx.x;
  s[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < LK_THREADS; off <<= 1) {
    const uint32_t a = t >= off ? s[t - off] : 0u;
    __syncthreads();
    s[t] += a;
    __syncthreads();
  }
  total = s[LK_THREADS - 1];
  const uint32_t r = s[t] - v;
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(LK_THREADS) void lk_scan_sums_kernel(const uint32_t* flags, uint64_t n, uint64_t stride, uint32_t invert,
                                                                  uint32_t* sums, uint32_t sblocks) {
  __shared__ uint32_t s[LK_THREADS];
  flags += (size_t)blockIdx.y * stride;
  sums += (size_t)blockIdx.y * sblocks;
  const uint64_t b0 = (uint64_t)blockIdx.x * SC_BLOCK + (uint64_t)threadIdx.x * SC_ITEMS;
  uint32_t v = 0;
  for (int k = 0; k < SC_ITEMS; ++k)
    if (b0 + k < n) v += flags[b0 + k] ^ invert;
  uint32_t total;
  (void)lk_block_scan(v, s, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup per pair; the pair's grand total goes to small[4 p + which]
__global__ __launch_bounds__(LK_THREADS) void lk_scan_top_kernel(uint32_t* sums, uint32_t nblocks, uint32_t sblocks, uint32_t* small, uint32_t which) {
  __shared__ uint32_t s[LK_THREADS];
  sums += (size_t)blockIdx.x * sblocks;
  const uint32_t per = (nblocks + LK_THREADS - 1) / LK_THREADS;
  const uint32_t lo = threadIdx.x * per < nblocks ? threadIdx.x * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
  uint32_t v = 0;
  for (uint32_t i = lo; i < hi; ++i) v += sums[i];
  uint32_t total;
  uint32_t run = lk_block_scan(v, s, total);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t c = sums[i];
    sums[i] = run;
    run += c;
  }
  if (threadIdx.x == 0) small[blockIdx.x * 4 + which] = total;
}
// positions[rank] = index, for every index whose (possibly inverted) flag is set -- the compacted list in index order
__global__ __launch_bounds__(LK_THREADS) void lk_scan_apply_kernel(const uint32_t* flags, uint64_t n, uint64_t stride, uint32_t invert,
                                                                   const uint32_t* sums, uint32_t sblocks, uint32_t* rank_of, uint32_t* positions) {
  __shared__ uint32_t s[LK_THREADS];
  flags += (size_t)blockIdx.y * stride;
  sums += (size_t)blockIdx.y * sblocks;
  if (rank_of) rank_of += (size_t)blockIdx.y * stride;
  positions += (size_t)blockIdx.y * stride;
  const uint64_t b0 = (uint64_t)blockIdx.x * SC_BLOCK + (uint64_t)threadIdx.x * SC_ITEMS;
  uint32_t f[SC_ITEMS], v = 0;
  for (int k = 0; k < SC_ITEMS; ++k) {
    f[k] = b0 + k < n ? flags[b0 + k] ^ invert : 0u;
    v += f[k];
  }
  uint32_t total;
  uint32_t run = sums[blockIdx.x] + lk_block_scan(v, s, total);
  for (int k = 0; k < SC_ITEMS; ++k) {
    if (b0 + k < n) {
      if (rank_of) rank_of[b0 + k] = run;
      if (f[k]) positions[run] = (uint32_t)(b0 + k);
    }
    run += f[k];
  }
}

// S'[i] = A'[i] at first occurrences; the repeated row of rank r (ascending) takes the leftover table value of rank
// m - 1 - r (upstream pops the repeated rows from the back while walking the leftovers upwards).  Both columns return
// to Montgomery words.  small[4 p + ..]: [0] repeated rows, [1] leftovers, [2] missing, [3] mismatch.
__global__ __launch_bounds__(LK_THREADS) void lk_fill_kernel(const uint32_t* keys, uint64_t pitch, uint64_t rows, uint64_t stride,
                                                             const uint32_t* repeated, const uint32_t* rep_rank, const uint32_t* left_pos,
                                                             uint32_t* small, LkFr c, LkPtrs ptr) {
  const uint32_t p = blockIdx.y;
  const uint32_t *a_sorted = keys + (size_t)(2 * p) * pitch, *t_sorted = keys + (size_t)(2 * p + 1) * pitch;
  repeated += (size_t)p * stride;
  rep_rank += (size_t)p * stride;
  left_pos += (size_t)p * stride;
  const uint64_t i = (uint64_t)blockIdx.x * LK_THREADS + threadIdx.x;
  if (i >= rows) return;
  const Key a = key_load(a_sorted, i);
  Key sv = a;
  if (repeated[i]) {
    const uint32_t m = small[p * 4 + 0];
    if (small[p * 4 + 1] != m) {                   // cannot happen when every first occurrence found its table value
      atomicExch(small + p * 4 + 3, 1u);
      return;
    }
    sv = key_load(t_sorted, left_pos[m - 1 - rep_rank[i]]);
  }
  Fr cc;
#pragma unroll
  for (int j = 0; j < 9; ++j) cc.l[j] = c.l[j];
  HM_DECLARE(cc, 1.0);
  Key o;
  fe_pack(o.w, fe_canonical(fe_mul(fe_unpack<FrParams>(a.w), cc)));
  key_store(ptr.out[2 * p], i, o);
  fe_pack(o.w, fe_canonical(fe_mul(fe_unpack<FrParams>(sv.w), cc)));
  key_store(ptr.out[2 * p + 1], i, o);
}

// ---------------------------------------------------------------------------------------------
// sorts `arrays` key arrays of n_pow2 keys each, `pitch` words apart, side by side (grid.y)
static int lk_sort(uint32_t* d_keys, uint64_t n_pow2, uint32_t arrays, uint64_t pitch, hipStream_t stream) {
  if (n_pow2 < 2) return HM_OK;
  const uint32_t tiles = (uint32_t)((n_pow2 + LK_TILE - 1) / LK_TILE);
  const size_t lds = (size_t)8 * LK_TILE * 4;
  const uint64_t in_tile = n_pow2 < LK_TILE ? n_pow2 : (uint64_t)LK_TILE;
  hipLaunchKernelGGL(lk_sort_tile_kernel, dim3(tiles, arrays), dim3(LK_THREADS), lds, stream, d_keys, n_pow2, (uint64_t)2, in_tile, pitch);
  for (uint64_t k = (uint64_t)LK_TILE * 2; k <= n_pow2; k <<= 1) {
    uint64_t j = k / 2;
    for (; j >= 2 * (uint64_t)LK_TILE; j >>= 2)                    // two global stages per launch while two remain
      hipLaunchKernelGGL(lk_sort_global2_kernel, dim3((uint32_t)((n_pow2 / 4 + LK_THREADS - 1) / LK_THREADS), arrays), dim3(LK_THREADS), 0,
                         stream, d_keys, n_pow2, k, j, pitch);
    if (j >= LK_TILE)
      hipLaunchKernelGGL(lk_sort_global_kernel, dim3((uint32_t)((n_pow2 / 2 + LK_THREADS - 1) / LK_THREADS), arrays), dim3(LK_THREADS), 0,
                         stream, d_keys, n_pow2, k, j, pitch);
    hipLaunchKernelGGL(lk_sort_tile_kernel, dim3(tiles, arrays), dim3(LK_THREADS), lds, stream, d_keys, n_pow2, k, k, pitch);
  }
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

// the same for keys below 2^bits: three launches (see lk_count_kernel); counters: arrays x 2^bits words, zeroed here
static int lk_sort_small(uint32_t* d_keys, uint64_t live, uint64_t n_pow2, uint32_t arrays, uint64_t pitch, uint32_t bits, uint32_t* d_counters,
                         hipStream_t stream) {
  const uint32_t nbins = 1u << bits;
  HM_HIP_CHECK(hipMemsetAsync(d_counters, 0, (size_t)arrays * nbins * 4, stream));
  hipLaunchKernelGGL(lk_count_kernel, dim3((uint32_t)((live + LK_THREADS - 1) / LK_THREADS), arrays), dim3(LK_THREADS), 0, stream,
                     (const uint32_t*)d_keys, pitch, live, d_counters, nbins);
  hipLaunchKernelGGL(lk_count_scan_kernel, dim3(arrays), dim3(1024), 0, stream, d_counters, nbins);
  hipLaunchKernelGGL(lk_count_expand_kernel, dim3((uint32_t)((n_pow2 + LK_THREADS - 1) / LK_THREADS), arrays), dim3(LK_THREADS), 0, stream,
                     d_keys, pitch, live, n_pow2, (const uint32_t*)d_counters, nbins);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

static int lk_compact(const uint32_t* d_flags, uint64_t n, uint64_t stride, uint32_t pairs, uint32_t invert, uint32_t* d_sums, uint32_t* d_small,
                      uint32_t which, uint32_t* d_rank, uint32_t* d_positions, hipStream_t stream) {
  const uint32_t blocks = (uint32_t)((n + SC_BLOCK - 1) / SC_BLOCK);
  hipLaunchKernelGGL(lk_scan_sums_kernel, dim3(blocks, pairs), dim3(LK_THREADS), 0, stream, d_flags, n, stride, invert, d_sums, blocks);
  hipLaunchKernelGGL(lk_scan_top_kernel, dim3(pairs), dim3(LK_THREADS), 0, stream, d_sums, blocks, blocks, d_small, which);
  hipLaunchKernelGGL(lk_scan_apply_kernel, dim3(blocks, pairs), dim3(LK_THREADS), 0, stream, d_flags, n, stride, invert, (const uint32_t*)d_sums,
                     blocks, d_rank, d_positions);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

// `pairs` lookups of the same length through one launch chain.  d_inputs[p], d_tables[p]: columns of which the first
// `rows` entries are read; d_out_inputs[p], d_out_tables[p]: rows [0, rows) written.  missing[p] (optional) tells which
// lookup failed.  Returns HM_ERR_NOT_FOUND when an input value is missing from its table (upstream:
// Error::ConstraintSystemFailure).
int lookup_permute_run(DeviceCtx& ctx, const void* const* d_inputs, const void* const* d_tables, size_t pairs, uint64_t rows,
                       void* const* d_out_inputs, void* const* d_out_tables, int* missing, hipStream_t stream) {
  if (missing)
    for (size_t p = 0; p < pairs; ++p) missing[p] = 0;
  if (rows == 0 || pairs == 0) return HM_OK;
  if (rows >= ((uint64_t)1 << 31)) return hm_fail(HM_ERR_BAD_ARG, "lookup permute: too many rows");
  uint64_t n2 = 1;
  while (n2 < rows) n2 <<= 1;
  if (!ctx.lookup_attr_set) {
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lk_sort_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(8 * LK_TILE * 4)));
    ctx.lookup_attr_set = true;
  }
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  const uint32_t blocks = (uint32_t)((rows + SC_BLOCK - 1) / SC_BLOCK);
  const uint32_t P = (uint32_t)(pairs < (size_t)LK_MAX_PAIRS ? pairs : (size_t)LK_MAX_PAIRS);      // pairs per chain
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const uint64_t stride = (rows + 63) & ~(uint64_t)63, pitch = n2 * 8;
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += align(bytes); return o; };
  const size_t o_keys = carve((size_t)2 * P * n2 * 32), o_rep = carve((size_t)P * stride * 4), o_used = carve((size_t)P * stride * 4),
               o_rank = carve((size_t)P * stride * 4), o_left = carve((size_t)P * stride * 4), o_reppos = carve((size_t)P * stride * 4),
               o_sums = carve((size_t)P * blocks * 4 * 2), o_small = carve((size_t)LK_MAX_PAIRS * 16), o_kbits = carve(LK_KB_WORDS * 4);
  // the counting sort of small keys wants 2^bits counters per key array; they are carved only when the sort can be used at all
  // (more keys than a few LDS tiles: below that the bitonic network is one or two launches), sized for the worst case it accepts
  static const bool count_sort_on = [] { const char* v = std::getenv("HALO2_MI355X_LOOKUP_COUNT_SORT"); return !(v && *v == '0'); }();   // A/B
  const bool may_count = count_sort_on && n2 > 4 * (uint64_t)LK_TILE;
  const size_t o_counters = carve(may_count ? (size_t)2 * P * ((size_t)4 << LK_COUNT_BITS) : 4);
  uint8_t* ws = (uint8_t*)slot->scratch.ensure(off);
  if (!ws) return hm_fail(HM_ERR_HIP, "lookup permute: scratch allocation failed");
  uint32_t *keys = (uint32_t*)(ws + o_keys), *rep = (uint32_t*)(ws + o_rep), *used = (uint32_t*)(ws + o_used);
  uint32_t *rank = (uint32_t*)(ws + o_rank), *left = (uint32_t*)(ws + o_left), *reppos = (uint32_t*)(ws + o_reppos);
  uint32_t *sums = (uint32_t*)(ws + o_sums), *small = (uint32_t*)(ws + o_small), *kbits = (uint32_t*)(ws + o_kbits);
  uint32_t* counters = (uint32_t*)(ws + o_counters);
  // the raw words are the internal form of v / 32; times the internal form of 32 * 2^-256 (the integer 32 ... see
  // msm_s_digits_kernel) they become the canonical integer v (as Fr::to_repr); and back with 2^256
  LkFr to_canon, to_mont;
  {
    const host::Fr4 one_int = {{1, 0, 0, 0}};                       // the integer 1 read as Montgomery words = 2^-256
    host::fr_to_internal9(one_int, to_canon.l);
    const host::Fr4 r2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};   // 2^512 mod r as words = the element 2^256
    host::fr_to_internal9(r2, to_mont.l);
  }
  int result = HM_OK;
  for (size_t p0 = 0; p0 < pairs; p0 += P) {
    const uint32_t cnt = (uint32_t)(pairs - p0 < (size_t)P ? pairs - p0 : (size_t)P);
    LkPtrs ptr;
    std::memset(&ptr, 0, sizeof ptr);
    for (uint32_t p = 0; p < cnt; ++p) {
      ptr.in[2 * p] = (const uint32_t*)d_inputs[p0 + p];
      ptr.in[2 * p + 1] = (const uint32_t*)d_tables[p0 + p];
      ptr.out[2 * p] = (uint32_t*)d_out_inputs[p0 + p];
      ptr.out[2 * p + 1] = (uint32_t*)d_out_tables[p0 + p];
    }
    HM_HIP_CHECK(hipMemsetAsync(used, 0, (size_t)cnt * stride * 4, stream));
    HM_HIP_CHECK(hipMemsetAsync(small, 0, (size_t)LK_MAX_PAIRS * 16, stream));
    const uint32_t cb = (uint32_t)((n2 + LK_THREADS - 1) / LK_THREADS), rb = (uint32_t)((rows + LK_THREADS - 1) / LK_THREADS);
    HM_HIP_CHECK(hipMemsetAsync(kbits, 0, LK_KB_WORDS * 4, stream));
    hipLaunchKernelGGL(lk_convert_kernel, dim3(cb, 2 * cnt), dim3(LK_THREADS), 0, stream, ptr, keys, pitch, rows, n2, to_canon, kbits);
    uint32_t small_bits = 0;                                       // != 0: every live key of the chain is below 2^small_bits
    if (may_count) {                                               // (one small copy and a wait: the call synchronises at its end anyway)
      static thread_local std::vector<uint32_t> h_bits_v(LK_KB_WORDS);
      uint32_t* h_bits = h_bits_v.data();
      HM_HIP_CHECK(hipMemcpyAsync(h_bits, kbits, LK_KB_WORDS * 4, hipMemcpyDeviceToHost, stream));
      HM_HIP_CHECK(hipStreamSynchronize(stream));
      uint32_t lo_or = 0, hi_or = 0;
      for (uint32_t a = 0; a < 2 * cnt; ++a)
        for (uint32_t sl = 0; sl < LK_KB_SLOTS; ++sl) {
          lo_or |= h_bits[((size_t)a * LK_KB_SLOTS + sl) * LK_KB_STRIDE];
          hi_or |= h_bits[((size_t)a * LK_KB_SLOTS + sl) * LK_KB_STRIDE + 1];
        }
      uint32_t need = 1;
      while (need < 32 && (lo_or >> need) != 0) ++need;
      if (hi_or == 0 && need <= LK_COUNT_BITS) small_bits = need;
    }
    int rc = small_bits ? lk_sort_small(keys, rows, n2, 2 * cnt, pitch, small_bits, counters, stream)
                        : lk_sort(keys, n2, 2 * cnt, pitch, stream);            // every column of the chain side by side
    if (rc != HM_OK) return rc;
    hipLaunchKernelGGL(lk_mark_kernel, dim3(rb, cnt), dim3(LK_THREADS), 0, stream, (const uint32_t*)keys, pitch, rows, stride, rep, used, small);
    rc = lk_compact(rep, rows, stride, cnt, 0, sums, small, 0, rank, reppos, stream);
    if (rc == HM_OK) rc = lk_compact(used, rows, stride, cnt, 1, sums + (size_t)P * blocks, small, 1, nullptr, left, stream);
    if (rc != HM_OK) return rc;
    hipLaunchKernelGGL(lk_fill_kernel, dim3(rb, cnt), dim3(LK_THREADS), 0, stream, (const uint32_t*)keys, pitch, rows, stride, (const uint32_t*)rep,
                       (const uint32_t*)rank, (const uint32_t*)left, small, to_mont, ptr);
    HM_HIP_CHECK(hipGetLastError());
    uint32_t h_small[4 * LK_MAX_PAIRS];
    HM_HIP_CHECK(hipMemcpyAsync(h_small, small, sizeof h_small, hipMemcpyDeviceToHost, stream));
    HM_HIP_CHECK(hipStreamSynchronize(stream));
    for (uint32_t p = 0; p < cnt; ++p) {
      if (h_small[4 * p + 2]) {
        if (missing) missing[p0 + p] = 1;
        if (result == HM_OK) result = HM_ERR_NOT_FOUND;
      } else if (h_small[4 * p + 3] && result == HM_OK) {
        result = HM_ERR_INTERNAL;
      }
    }
  }
  const int rrc = aux_release(ctx, slot, stream);
  if (result == HM_ERR_NOT_FOUND) return hm_fail(HM_ERR_NOT_FOUND, "lookup permute: an input value is missing from the table (ConstraintSystemFailure)");
  if (result == HM_ERR_INTERNAL) return hm_fail(HM_ERR_INTERNAL, "lookup permute: leftover / repeated counts differ");
  return rrc;
}

}  // namespace hm
