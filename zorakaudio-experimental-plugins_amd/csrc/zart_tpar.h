// zart_tpar.h -- wavefront primitives of the time-parallel kernels that zajit/tpar.py generates (one wavefront per
// instance, lane = frame): DPP moves of doubles, prefix scans of affine maps, lane broadcast.
//
// A recurrence y[t] = a[t] y[t-1] + b[t] is the composition of the per-frame maps f_t(y) = a[t] y + b[t]; composition is
// associative, so the wavefront computes all 64 prefixes f_t o ... o f_0 in six DPP steps (row_shr 1/2/4/8 inside the
// 16-lane rows, row_bcast15 and row_bcast31 across them -- no LDS, no lane masks). Lanes without a source keep the identity
// map (a = 1, b = 0), which is what update_dpp's `old` operand is for. The 2 x 2 form serves biquads and other coupled
// pairs. The scheme is the one of the hand-written DDT kernel (kernels/ddt_fast.hip.h:98-108), generalised from a constant
// pole to per-frame coefficients so that sample-and-hold (a in {0, 1}), gated integrators and smoothed coefficients scan too.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZT_ROW_SHR(n) (0x110 | (n))
#define ZT_WAVE_SHR1 0x138
#define ZT_ROW_BCAST15 0x142
#define ZT_ROW_BCAST31 0x143

__device__ __forceinline__ double zt_readlane(double v, int l) {      // l must be wave-uniform
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_readlane(t.x, l);
  t.y = __builtin_amdgcn_readlane(t.y, l);
  return __builtin_bit_cast(double, t);
}

// value of an arbitrary lane, per lane (LDS crossbar, no LDS memory); src must be 0..63
__device__ __forceinline__ double zt_bperm(double v, int src) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_ds_bpermute(src << 2, t.x);
  t.y = __builtin_amdgcn_ds_bpermute(src << 2, t.y);
  return __builtin_bit_cast(double, t);
}

// A value every lane computed identically, declared wave-uniform: it can live in scalar registers (or, spilled, in single
// lanes of a vector register) instead of occupying a vector register pair in all 64 lanes.
__device__ __forceinline__ double zt_uniform(double v) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_readfirstlane(t.x);
  t.y = __builtin_amdgcn_readfirstlane(t.y);
  return __builtin_bit_cast(double, t);
}

// lanes selected by ROWS that have a source lane under CTRL receive its value; every other lane receives `keep`
template <int CTRL, int ROWS>
__device__ __forceinline__ double zt_dpp(double v, double keep) {
  int2 t = __builtin_bit_cast(int2, v);
  const int2 k = __builtin_bit_cast(int2, keep);
  t.x = __builtin_amdgcn_update_dpp(k.x, t.x, CTRL, ROWS, 0xF, false);
  t.y = __builtin_amdgcn_update_dpp(k.y, t.y, CTRL, ROWS, 0xF, false);
  return __builtin_bit_cast(double, t);
}

// the same with every row enabled, lanes without a source receive 0: no `old` value has to be put into the destination first
template <int CTRL>
__device__ __forceinline__ double zt_dppz(double v) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_mov_dpp(t.x, CTRL, 0xF, 0xF, true);
  t.y = __builtin_amdgcn_mov_dpp(t.y, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, t);
}

// "This frame left the variable alone" (zajit/tpar.py HOLD): a variable whose incoming value no path of @sample can read has no
// state-in node; where a frame does not write it, its value is this marker -- a quiet NaN that no arithmetic produces (payload
// in the low word, which a float -> double conversion leaves zero) and that only ever moves through selects.
#define ZT_HOLD __builtin_bit_cast(double, (uint64_t)0x7FF800005A5A5A5AULL)
__device__ __forceinline__ bool zt_is_hold(double v) { return __builtin_bit_cast(uint64_t, v) == 0x7FF800005A5A5A5AULL; }

// wave-wide minimum / maximum of a 64-bit integer (rare paths: conditional writes)
__device__ __forceinline__ int64_t zt_wave_min_i64(int64_t v) {
  for (int off = 1; off < 64; off <<= 1) { const int64_t o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ int64_t zt_wave_max_i64(int64_t v) {
  for (int off = 1; off < 64; off <<= 1) { const int64_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}
// A chunk's delay-line write covers [s0, s0 + k) and, after a ring's wrap, [s1, s1 + m). Do two writes touch one cell?
__device__ __forceinline__ bool zt_range_meet(int64_t a, int64_t na, int64_t b, int64_t nb) {
  return na > 0 && nb > 0 && a < b + nb && b < a + na;
}
__device__ __forceinline__ bool zt_spans_meet(int64_t s0a, int64_t ka, int64_t s1a, int64_t ma, int64_t s0b, int64_t kb, int64_t s1b,
                                              int64_t mb) {
  return zt_range_meet(s0a, ka, s0b, kb) || zt_range_meet(s0a, ka, s1b, mb) || zt_range_meet(s1a, ma, s0b, kb) ||
         zt_range_meet(s1a, ma, s1b, mb);
}
__device__ __forceinline__ bool zt_span_hits(int64_t s, int64_t n, int64_t lo, int64_t hi) { return n > 0 && s <= hi && lo < s + n; }
// Per-trip cells of a uniform loop: address expression 0 visits a0 + k * s0 (range [lo0, hi0]), expression 1 a1 + k * s1. They
// never name one cell if their ranges are disjoint, or they walk in step through interleaved records (same stride, first
// addresses that differ by less than a multiple of it). An empty range (no trips) meets nothing.
__device__ __forceinline__ bool zt_sites_ok(int64_t a0, int64_t s0, int64_t lo0, int64_t hi0, int64_t a1, int64_t s1, int64_t lo1,
                                            int64_t hi1) {
  if (hi0 < lo0 || hi1 < lo1) return true;
  if (hi0 < lo1 || hi1 < lo0) return true;
  if (s0 != s1 || s0 == 0) return false;
  return (a1 - a0) % s0 != 0;
}

// LDS doubles for the per-trip cells of a block's uniform loops (band states: CMD 26 cells x 32 bands)
#ifndef ZT_CELL_DOUBLES
#define ZT_CELL_DOUBLES 2048
#endif

// an integer small enough that sums and products of a few of them stay exact in a double (loop counters run in strips: i0 + k * step)
__device__ __forceinline__ bool zt_small_int(double x) { return x == floor(x) && fabs(x) < 1.0e12; }
// ... that is a power of two (ring lengths written as masks: (pos + 1) & (N - 1))
__device__ __forceinline__ bool zt_pow2(double x) { const long long v = (long long)x; return v >= 1 && (v & (v - 1)) == 0; }

// LDS doubles for the staged windows of the rings a uniform loop gathers from (zajit/tpar.py RingGroup): 24 KB keeps four
// wavefronts per CU beside the other tables; a window that does not fit leaves its loop on the gathers from memory
#ifndef ZT_RING_DOUBLES
#define ZT_RING_DOUBLES 3072
#endif

// All address expressions of a loop against each other (zs: [5][n] = first address, stride, previous, lowest, highest; bit j of
// `stored`: expression j is stored to): one pair per lane and pass, not inlined -- with 26 expressions (CMD) the 325 inline pair
// tests, each with a 64-bit remainder, were most of the kernel's code.
static __device__ __attribute__((noinline)) bool zt_sites_all_ok(const long long* zs, int n, unsigned long long stored, int lane) {
  bool bad = false;
  for (int q = lane; q < n * n; q += 64) {
    const int j = q / n, k = q - j * n;
    if (k <= j || !(((stored >> j) | (stored >> k)) & 1ull)) continue;
    bad |= !zt_sites_ok(zs[j], zs[n + j], zs[3 * n + j], zs[4 * n + j], zs[k], zs[n + k], zs[3 * n + k], zs[4 * n + k]);
  }
  return __ballot(bad) == 0ull;
}

// value of the previous lane; lane 0 receives `first` (the value carried in from the previous chunk)
__device__ __forceinline__ double zt_shift1(double v, double first) { return zt_dpp<ZT_WAVE_SHR1, 0xF>(v, first); }

// ---- y[t] = a[t] y[t-1] + b[t]: on return (a, b) of lane t is the map from the state before the chunk to the state after
// frame t, i.e. y[t] = a * y_in + b --------------------------------------------------------------------------------------
#define ZT_SCAN1_ROW(CTRL)                                        \
  {                                                              \
    const double as_ = zt_dpp<CTRL, 0xF>(a, 1.0);                \
    const double bs_ = zt_dppz<CTRL>(b);                         \
    b = __builtin_fma(a, bs_, b);                                \
    a = a * as_;                                                 \
  }
#define ZT_SCAN1_STEP(CTRL, ROWS)                                \
  {                                                              \
    const double as_ = zt_dpp<CTRL, ROWS>(a, 1.0);               \
    const double bs_ = zt_dpp<CTRL, ROWS>(b, 0.0);               \
    b = __builtin_fma(a, bs_, b);                                \
    a = a * as_;                                                 \
  }
__device__ __forceinline__ void zt_scan1(double& a, double& b) {
  ZT_SCAN1_ROW(ZT_ROW_SHR(1))
  ZT_SCAN1_ROW(ZT_ROW_SHR(2))
  ZT_SCAN1_ROW(ZT_ROW_SHR(4))
  ZT_SCAN1_ROW(ZT_ROW_SHR(8))
  ZT_SCAN1_STEP(ZT_ROW_BCAST15, 0xA)
  ZT_SCAN1_STEP(ZT_ROW_BCAST31, 0xC)
}
#undef ZT_SCAN1_STEP
#undef ZT_SCAN1_ROW

// ---- the same recurrence with a coefficient that is constant over the launch: only b moves through the lanes. The powers
// a^2, a^4, a^8, a^16 are wave-uniform, the position-dependent weight w = a^((lane & 15) + 1) of the two cross-row steps is a
// per-lane constant of the coefficient (one LDS row per distinct coefficient, zt_pow_row fills it once per launch). The state
// carried into the chunk is injected at lane 0 (b[0] += a * y_in), so the scan itself delivers y[t] -- the DDT kernel's scheme.
struct ZtPow { double a2, a4, a8, a16; };
__device__ __forceinline__ double zt_pow_row(double a, int lane) {   // a^((lane & 15) + 1)
  double r = 1.0, p = a;
  const int e = (lane & 15) + 1;
#pragma unroll
  for (int k = 0; k < 5; ++k) { r = ((e >> k) & 1) ? r * p : r; p = p * p; }
  return r;
}
__device__ __forceinline__ double zt_scan1_sum(double b, double y_in, int lane);
__device__ __forceinline__ double zt_scan1_inv(double b, double a, const ZtPow& q, double w, double y_in, int lane) {
  // (a is wave-uniform.) 0 and 1 are what a variable assigned under a launch-constant condition turns into -- `mode ? x = e`
  // is x = mode ? e : x, i.e. a = 0 or a = 1 for the whole launch -- and neither needs the weighted scan
  if (a == 0.0) return b;
  if (a == 1.0) return zt_scan1_sum(b, y_in, lane);
  b = lane == 0 ? __builtin_fma(a, y_in, b) : b;
  b = __builtin_fma(a, zt_dppz<ZT_ROW_SHR(1)>(b), b);
  b = __builtin_fma(q.a2, zt_dppz<ZT_ROW_SHR(2)>(b), b);
  b = __builtin_fma(q.a4, zt_dppz<ZT_ROW_SHR(4)>(b), b);
  b = __builtin_fma(q.a8, zt_dppz<ZT_ROW_SHR(8)>(b), b);
  b = __builtin_fma(w, zt_dpp<ZT_ROW_BCAST15, 0xA>(b, 0.0), b);
  b = __builtin_fma(lane >= 48 ? w * q.a16 : w, zt_dpp<ZT_ROW_BCAST31, 0xC>(b, 0.0), b);
  return b;
}
// a == 1: a running sum
__device__ __forceinline__ double zt_scan1_sum(double b, double y_in, int lane) {
  b = lane == 0 ? b + y_in : b;
  b += zt_dppz<ZT_ROW_SHR(1)>(b);
  b += zt_dppz<ZT_ROW_SHR(2)>(b);
  b += zt_dppz<ZT_ROW_SHR(4)>(b);
  b += zt_dppz<ZT_ROW_SHR(8)>(b);
  b += zt_dpp<ZT_ROW_BCAST15, 0xA>(b, 0.0);
  b += zt_dpp<ZT_ROW_BCAST31, 0xC>(b, 0.0);
  return b;
}

// ---- two coupled states: y[t] = A[t] y[t-1] + b[t], A = [a00 a01; a10 a11] ---------------------------------------------
// D coupled states (round 4: three -- a gate, its smoother and the smoother's previous value; a play position, its direction and
// an "active" latch): the maps y -> A y + b of the 64 frames composed by the same six DPP steps, D x D + D values per lane.
template <int D> struct ZtMapN { double a[D][D]; double b[D]; };
template <int D, int CTRL, int ROWS> __device__ __forceinline__ void zt_scanN_step(ZtMapN<D>& m) {
  ZtMapN<D> s, r;
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) s.a[i][j] = zt_dpp<CTRL, ROWS>(m.a[i][j], i == j ? 1.0 : 0.0);     // (no source lane: the identity map)
    s.b[i] = zt_dpp<CTRL, ROWS>(m.b[i], 0.0);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double acc = m.a[i][0] * s.a[0][j];
#pragma unroll
      for (int k = 1; k < D; ++k) acc = __builtin_fma(m.a[i][k], s.a[k][j], acc);
      r.a[i][j] = acc;
    }
    double acc = m.b[i];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) acc = __builtin_fma(m.a[i][k], s.b[k], acc);
    r.b[i] = acc;
  }
  m = r;
}
template <int D> __device__ __forceinline__ void zt_scanN(ZtMapN<D>& m) {
  zt_scanN_step<D, ZT_ROW_SHR(1), 0xF>(m);
  zt_scanN_step<D, ZT_ROW_SHR(2), 0xF>(m);
  zt_scanN_step<D, ZT_ROW_SHR(4), 0xF>(m);
  zt_scanN_step<D, ZT_ROW_SHR(8), 0xF>(m);
  zt_scanN_step<D, ZT_ROW_BCAST15, 0xA>(m);
  zt_scanN_step<D, ZT_ROW_BCAST31, 0xC>(m);
}
// the states such a scan implies at the END of each frame, given the carried-in state c[]
template <int D> __device__ __forceinline__ double zt_mapN_apply(const ZtMapN<D>& m, int i, const double* c) {
  double acc = m.b[i];
#pragma unroll
  for (int k = D - 1; k >= 0; --k) acc = __builtin_fma(m.a[i][k], c[k], acc);
  return acc;
}

struct ZtMap2 { double a00, a01, a10, a11, b0, b1; };
#define ZT_SCAN2_STEP(CTRL, ROWS)                                                             \
  {                                                                                           \
    const double s00 = zt_dpp<CTRL, ROWS>(m.a00, 1.0), s01 = zt_dpp<CTRL, ROWS>(m.a01, 0.0);   \
    const double s10 = zt_dpp<CTRL, ROWS>(m.a10, 0.0), s11 = zt_dpp<CTRL, ROWS>(m.a11, 1.0);   \
    const double t0 = zt_dpp<CTRL, ROWS>(m.b0, 0.0), t1 = zt_dpp<CTRL, ROWS>(m.b1, 0.0);       \
    ZtMap2 r;                                                                                 \
    r.a00 = __builtin_fma(m.a00, s00, m.a01 * s10);                                           \
    r.a01 = __builtin_fma(m.a00, s01, m.a01 * s11);                                           \
    r.a10 = __builtin_fma(m.a10, s00, m.a11 * s10);                                           \
    r.a11 = __builtin_fma(m.a10, s01, m.a11 * s11);                                           \
    r.b0 = __builtin_fma(m.a00, t0, __builtin_fma(m.a01, t1, m.b0));                          \
    r.b1 = __builtin_fma(m.a10, t0, __builtin_fma(m.a11, t1, m.b1));                          \
    m = r;                                                                                    \
  }
__device__ __forceinline__ void zt_scan2(ZtMap2& m) {
  ZT_SCAN2_STEP(ZT_ROW_SHR(1), 0xF)
  ZT_SCAN2_STEP(ZT_ROW_SHR(2), 0xF)
  ZT_SCAN2_STEP(ZT_ROW_SHR(4), 0xF)
  ZT_SCAN2_STEP(ZT_ROW_SHR(8), 0xF)
  ZT_SCAN2_STEP(ZT_ROW_BCAST15, 0xA)
  ZT_SCAN2_STEP(ZT_ROW_BCAST31, 0xC)
}
#undef ZT_SCAN2_STEP

// ---- rand(): MT19937 as za_mt_next (zart.h) runs it, readable at any position of a chunk ----------------------------------
// A launch keeps two generations of the generator side by side in LDS (mt[0..624) = the instance's current table, mt[624..1248)
// = the next one). The word for the k-th call of the launch sits at position pos0 + k of that pair, so every lane can fetch the
// words of its frame once the per-lane call counts are known (they are an ordinary prefix sum). A new generation is produced
// by the whole wavefront in three lane-parallel phases: element k needs new[k - 227] from k = 227 on, so [0, 227), [227, 454)
// and [454, 623) are each parallel inside; element 623 closes the ring. At most 624 words may be consumed per chunk.
#define ZT_MT_N 624
#define ZT_MT_M 397
__device__ __forceinline__ uint32_t zt_mt_tw(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
__device__ __forceinline__ void zt_mt_twist(const uint32_t* cur, uint32_t* nxt, int lane) {
  for (int k = lane; k < 227; k += 64) nxt[k] = cur[k + ZT_MT_M] ^ zt_mt_tw(cur[k], cur[k + 1]);
  __syncthreads();
  for (int k = 227 + lane; k < 454; k += 64) nxt[k] = nxt[k - 227] ^ zt_mt_tw(cur[k], cur[k + 1]);
  __syncthreads();
  for (int k = 454 + lane; k < 623; k += 64) nxt[k] = nxt[k - 227] ^ zt_mt_tw(cur[k], cur[k + 1]);
  __syncthreads();
  if (lane == 0) nxt[623] = nxt[396] ^ zt_mt_tw(cur[623], nxt[0]);
  __syncthreads();
}
// Start of a launch: returns pos0, the position of the launch's first word. An instance that never called rand() (index 0) is
// seeded here the way za_mt_next seeds it on first use, positioned at the end of that generation.
__device__ __forceinline__ int zt_mt_begin(uint32_t* mt, const uint32_t* gmt, int64_t gstride, uint32_t gindex, int lane) {
  int pos0;
  if (gindex == 0) {
    if (lane == 0) {
      uint32_t prev = 0x4141F00Du;
      mt[0] = prev;
      for (int k = 1; k < ZT_MT_N; ++k) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k; mt[k] = prev; }
    }
    pos0 = ZT_MT_N;
  } else {
    for (int k = lane; k < ZT_MT_N; k += 64) mt[k] = gmt[k * gstride];
    pos0 = (int)gindex;
  }
  __syncthreads();
  zt_mt_twist(mt, mt + ZT_MT_N, lane);
  return pos0;
}
__device__ __forceinline__ double zt_mt_word(const uint32_t* mt, int pos0, double index) {
  int p = pos0 + (int)index;
  p = p < 0 ? 0 : (p > 2 * ZT_MT_N - 1 ? 2 * ZT_MT_N - 1 : p);      // (lanes past the end of a launch hold arbitrary counts)
  uint32_t y = mt[p];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return (double)y;
}
// End of a chunk, `total` words consumed so far in the launch (wave-uniform): once the last consumed word lies in the next
// generation that one becomes current and its successor is produced.
__device__ __forceinline__ void zt_mt_retire(uint32_t* mt, int& pos0, int total, int lane) {
  if (pos0 + total > ZT_MT_N) {
    __syncthreads();
    for (int k = lane; k < ZT_MT_N; k += 64) mt[k] = mt[ZT_MT_N + k];
    __syncthreads();
    zt_mt_twist(mt, mt + ZT_MT_N, lane);
    pos0 -= ZT_MT_N;
  }
}
// End of the launch: what za_mt_next would have left -- the generation of the last consumed word and the index after it; an
// instance that consumed nothing keeps its table (and stays unseeded if it was).
__device__ __forceinline__ void zt_mt_end(const uint32_t* mt, int pos0, int total, uint32_t* gmt, int64_t gstride, uint32_t* gindex, int lane) {
  if (total <= 0) return;
  for (int k = lane; k < ZT_MT_N; k += 64) gmt[k * gstride] = mt[k];
  if (lane == 0) *gindex = (uint32_t)(pos0 + total);
}

// ---- coupled pair with a matrix that is constant over the launch (biquads): only b moves through the lanes. Per distinct
// matrix the launch keeps A^2, A^4, A^8 (wave-uniform, 12 doubles of LDS) and two per-lane weight matrices: WA = A^((lane & 15)
// + 1) for the row_bcast15 step (rows 1 and 3) and WB = A^(lane - 31) for the row_bcast31 step (rows 2 and 3). The state carried
// into the chunk is injected at lane 0.
struct ZtMat2 { double m00, m01, m10, m11; };
__device__ __forceinline__ ZtMat2 zt_mat_mul(const ZtMat2& x, const ZtMat2& y) {
  ZtMat2 r;
  r.m00 = __builtin_fma(x.m00, y.m00, x.m01 * y.m10); r.m01 = __builtin_fma(x.m00, y.m01, x.m01 * y.m11);
  r.m10 = __builtin_fma(x.m10, y.m00, x.m11 * y.m10); r.m11 = __builtin_fma(x.m10, y.m01, x.m11 * y.m11);
  return r;
}
__device__ __forceinline__ ZtMat2 zt_mat_pow(ZtMat2 p, int e) {      // e >= 0
  ZtMat2 r = {1.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int k = 0; k < 6; ++k) { if ((e >> k) & 1) r = zt_mat_mul(r, p); p = zt_mat_mul(p, p); }
  return r;
}
// fills tab[0..12) with A^2, A^4, A^8 and tab[12 + j * 64 + lane], j = 0..7, with WA (4 rows) and WB (4 rows)
__device__ __forceinline__ void zt_mat_table(double* tab, const ZtMat2& a, int lane) {
  const ZtMat2 a2 = zt_mat_mul(a, a), a4 = zt_mat_mul(a2, a2), a8 = zt_mat_mul(a4, a4);
  if (lane == 0) {
    tab[0] = a2.m00; tab[1] = a2.m01; tab[2] = a2.m10; tab[3] = a2.m11;
    tab[4] = a4.m00; tab[5] = a4.m01; tab[6] = a4.m10; tab[7] = a4.m11;
    tab[8] = a8.m00; tab[9] = a8.m01; tab[10] = a8.m10; tab[11] = a8.m11;
  }
  const ZtMat2 wa = zt_mat_pow(a, (lane & 15) + 1), wb = zt_mat_pow(a, lane >= 32 ? lane - 31 : 0);
  double* t = tab + 12 + lane;
  t[0] = wa.m00; t[64] = wa.m01; t[128] = wa.m10; t[192] = wa.m11;
  t[256] = wb.m00; t[320] = wb.m01; t[384] = wb.m10; t[448] = wb.m11;
}
#define ZT_MAT_TABLE_DOUBLES (12 + 8 * 64)
#define ZT_SCAN2I_ROW(CTRL, M)                                                                          \
  {                                                                                                     \
    const double t0 = zt_dppz<CTRL>(b0), t1 = zt_dppz<CTRL>(b1);                                        \
    b0 = __builtin_fma((M).m00, t0, __builtin_fma((M).m01, t1, b0));                                    \
    b1 = __builtin_fma((M).m10, t0, __builtin_fma((M).m11, t1, b1));                                    \
  }
#define ZT_SCAN2I_STEP(CTRL, ROWS, M)                                                                   \
  {                                                                                                     \
    const double t0 = zt_dpp<CTRL, ROWS>(b0, 0.0), t1 = zt_dpp<CTRL, ROWS>(b1, 0.0);                    \
    b0 = __builtin_fma((M).m00, t0, __builtin_fma((M).m01, t1, b0));                                    \
    b1 = __builtin_fma((M).m10, t0, __builtin_fma((M).m11, t1, b1));                                    \
  }
// tab: the matrix's table as filled by zt_mat_table; `zo` = opaque 0 (keeps the reads inside the caller's iteration)
__device__ __forceinline__ void zt_scan2_inv(double& b0, double& b1, const ZtMat2& a, const double* tab, int zo, double y0_in,
                                             double y1_in, int lane) {
  if (lane == 0) {
    b0 = __builtin_fma(a.m00, y0_in, __builtin_fma(a.m01, y1_in, b0));
    b1 = __builtin_fma(a.m10, y0_in, __builtin_fma(a.m11, y1_in, b1));
  }
  const double* u = tab + zo;
  const ZtMat2 a2 = {u[0], u[1], u[2], u[3]}, a4 = {u[4], u[5], u[6], u[7]}, a8 = {u[8], u[9], u[10], u[11]};
  ZT_SCAN2I_ROW(ZT_ROW_SHR(1), a)
  ZT_SCAN2I_ROW(ZT_ROW_SHR(2), a2)
  ZT_SCAN2I_ROW(ZT_ROW_SHR(4), a4)
  ZT_SCAN2I_ROW(ZT_ROW_SHR(8), a8)
  const double* w = u + 12 + lane;
  const ZtMat2 wa = {w[0], w[64], w[128], w[192]}, wb = {w[256], w[320], w[384], w[448]};
  ZT_SCAN2I_STEP(ZT_ROW_BCAST15, 0xA, wa)
  ZT_SCAN2I_STEP(ZT_ROW_BCAST31, 0xC, wb)
}
#undef ZT_SCAN2I_ROW
#undef ZT_SCAN2I_STEP
