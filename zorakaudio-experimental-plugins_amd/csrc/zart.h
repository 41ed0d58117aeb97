// zart.h -- per-instance runtime that zajit-generated section code executes against.
//
// One header, two compilers: hipcc --offload-arch=gfx950 (device functions, the product) and g++ (the CPU
// restatement under oracle/, test infrastructure). It restates the *lowering semantics* of the reference's
// AOT emitter so generated code can stay a thin expression tree:
//   value model / truthiness / ordered compares ..... dsp_jsfx_aot.py:3725, 4349-4352, 4315-4317
//   i32 integer ops (| & ~ % << >>) .................. dsp_jsfx_aot.py:4107-4114, 4355-4381
//   mem[] addressing (trunc(base+idx+1e-5), clamp) ... dsp_jsfx_aot.py:4062-4103
//   slider(i)/spl(i) ................................. dsp_jsfx_aot.py:3789-3837
//   loop() count ..................................... dsp_jsfx_aot.py:5663-5709
//   min/max/sign/sqr/invsqrt ......................... dsp_jsfx_aot.py:5223-5277
//   rand (MT19937, per instance) ..................... dsp_jsfx_aot.py:3880-4060, 5294-5323
//   memset ........................................... dsp_jsfx_aot.py:5439-5495
//   memcpy / convolve_c .............................. src/JSFXJuceProcessor.cpp:1341-1413
//   sliderchange / slider_automate / slider_show ..... src/JSFXJuceProcessor.cpp:2448-2475, dsp_jsfx_aot.py:5393-5437
//
// Differences from the reference that are deliberate: the arena is fixed-capacity (no realloc on
// device); reads past the capacity yield 0 exactly like a fresh zero-filled growth would, writes past it are dropped
// and latch ZA_ERR_MEM_OVERFLOW (+ the size that would have been needed) so the host fails loudly.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZA_FN __device__ inline __attribute__((always_inline))
// (FFT leaves are built with -DZA_INLINE_ALL, zajit/build.py: the transform code inlined into its kernels, so that the kernels'
//  register cap covers it. Never together with outlined user functions, whose whole point is not to be inlined.)
#if defined(ZA_INLINE_ALL) && !(defined(ZA_OUTLINE_FNS) && ZA_OUTLINE_FNS)
#define ZA_NOINLINE __device__ inline __attribute__((always_inline))
#else
#define ZA_NOINLINE __device__ __attribute__((noinline))
#endif
#else
#define ZA_FN static inline __attribute__((always_inline))
#define ZA_NOINLINE static __attribute__((noinline))
#endif
// The section functions za_section_{init,slider,block,sample}. A script with a very large variable table (ZA_BIG_STATE: Texture's
// 1087, Sample's 4058) runs from a state object in memory whatever is done, so inlining its sections into every
// kernel that runs them -- process, the time-parallel kernel's @block / event-frame functions, its serial tail, prepare, slider --
// buys nothing and costs the device compiler the same several-hundred-kilobyte body three to five times: they are real calls there.
#if defined(ZA_BIG_STATE) && ZA_BIG_STATE && defined(__HIPCC__) && !defined(ZA_INLINE_ALL)
#define ZA_SECTION_FN __device__ __attribute__((noinline))
#else
#define ZA_SECTION_FN ZA_FN
#endif
#if defined(ZA_OUTLINE_FNS) && ZA_OUTLINE_FNS
#define ZA_UFN ZA_NOINLINE     // user functions of a very large script: real calls (zajit/codegen.py)
#else
#define ZA_UFN ZA_FN
#endif

enum : uint32_t {
  ZA_ERR_MEM_OVERFLOW = 1u,   // a store addressed mem[] beyond the arena capacity
  ZA_ERR_LOOP_CAP = 2u,       // a loop()/while exceeded ZA_LOOP_CAP iterations and was cut
  ZA_ERR_UNSUPPORTED = 4u,    // a host-only builtin was reached on the device
  ZA_ERR_FFT_ARGS = 8u,       // informational: fft call ignored (bad size / page crossing), as the reference does
};

#ifndef ZA_LOOP_CAP
#define ZA_LOOP_CAP (int64_t(1) << 26)
#endif

#define ZA_STRING_BASE 1099511627776.0 /* 2^40: opaque string-literal handles, dsp_jsfx_aot.py:3681-3695 */

struct ZaGmemView;   // zart_gmem.h
struct ZaPoolView;   // zart_pool.h
struct ZaFileView;   // zart_file.h
struct ZaBusView;    // zart_msg.h

// The state of one instance while a section runs is split in two. ZaEnv is everything the runtime's out-of-line builtins (FFT,
// rand refill, msg_*, gmem block moves, file_*, pool export) may touch; ZaState adds the script's own variables, sliders and
// sample registers. An out-of-line builtin is entered through an inline stub (ZA_OUTCALL below) that hands it a COPY of the
// ZaEnv part and copies it back afterwards, so the address of the ZaState object itself never leaves the kernel: on the device
// that is what lets the compiler keep v[] / spl[] / sl[] in registers. (With `S& s` passed to one real call anywhere in a
// kernel, the whole object lives in scratch memory for the whole kernel and every variable access of the per-sample code is
// a scratch round trip: ~0.8 us per frame measured on an FFT leaf's otherwise empty @sample, tools/ring_io.py.)
template <bool LM>
struct ZaEnv {
  using Env = ZaEnv<LM>;
  static constexpr bool kLm = LM;   // this instantiation routes mem[0, lm_words) to LDS (device process kernel only)
  double srate, samplesblock, midi_bus, ext_midi_bus;
  double* mem;           // element a lives at mem[a * mem_stride]
  int64_t mem_stride;
  int64_t mem_cap;
  int64_t mem_high;      // 1 + highest element stored so far (the oracle's write-trace high-water mark)
  int64_t mem_need;      // capacity that would have satisfied every store seen
  uint32_t* mt;          // MT19937 words, word k at mt[k * mt_stride]
  int64_t mt_stride;
  uint32_t mti;
  uint32_t err;
  uint64_t pend_change, pend_automate, pend_automate_end;
  uint64_t vis_mask;
  int32_t vis_init;
  int32_t block_size;
  const ZaGmemView* gmem;
  const ZaPoolView* pool;
  uint64_t instance_id;
  double sink;
  double memtop;
  uint32_t gmem_attached; // this instance has called gmem_attach() (or the leaf declares options:gmem=)
  double* fft;           // FFT builtin scratch (natural-order work area), element a at fft[a * fft_stride]
  int64_t fft_stride;
  int64_t fft_cap;       // doubles available (0 when the leaf has no FFT builtins)
  uint32_t replica;      // this lane duplicates another lane's instance (zab_generic.hip.h): it must not raise requests of its own
  const ZaBusView* bus;  // message bus of the engine (zart_msg.h); null when the leaf has no msg_*() calls
  uint32_t inst_index;   // this instance's row in the bus tables (instance_id - first id of the engine)
  const ZaFileView* files; // file slots of the engine (zart_file.h); null when the leaf has no file builtins
  int64_t* fh;           // this instance's file handle words, word k at fh[k * fh_stride]
  int64_t fh_stride;
  uint32_t rep_i, rep_n, rep_stride;   // replica lanes of this instance (zab_generic.hip.h): my index, how many, lane stride
  uint32_t lm_words;     // LDS window over mem[0, lm_words) for the length of a launch (device only; 0 = none)
  uint32_t lm_stride;    // word a of this lane at za_lmem[a * lm_stride + lm_off]
  uint32_t lm_off;
};

template <int NV, bool LM = false>
struct ZaState : ZaEnv<LM> {
  double v[NV];
  double sl[64];
  double spl[64];
};

// body of an inline stub around an out-of-line builtin `call` (which names the environment copy `e`)
#define ZA_OUTCALL(call) typename S::Env e = s; auto r_ = (call); static_cast<typename S::Env&>(s) = e; return r_
#define ZA_OUTCALL_VOID(call) typename S::Env e = s; (call); static_cast<typename S::Env&>(s) = e

// ---------------------------------------------------------------------------------------------
// scalars
// ---------------------------------------------------------------------------------------------
ZA_FN bool za_truthy(double x) { return x < 0.0 || x > 0.0; }             // ordered x != 0
ZA_FN double za_b(bool c) { return c ? 1.0 : 0.0; }
ZA_FN double za_ne(double a, double b) { return (a < b || a > b) ? 1.0 : 0.0; }  // ordered !=
ZA_FN double za_not(double a) { return a == 0.0 ? 1.0 : 0.0; }
ZA_FN double za_neg(double a) { return 0.0 - a; }

ZA_FN int64_t za_f2i64(double x) {
  // fptosi: truncation toward zero. Out-of-range / NaN inputs are undefined in the reference's IR; pin them to the
  // x86 "integer indefinite" result (INT64_MIN) so CPU and GPU builds of this header agree with each other.
  if (!(x > -9.2233720368547758e18 && x < 9.2233720368547758e18)) return INT64_MIN;
  return (int64_t)x;
}
ZA_FN int32_t za_i32(double x) {
#if defined(__HIPCC__)
  // One v_cvt_i32_f64 when the operand fits 32 bits (the usual case for | & ~ << >> %); the wrap-around of wider values
  // needs the 64-bit conversion, ~9 VALU instructions. The empty asm keeps that a branch: as a select both would always run.
  if (__builtin_expect(x > -2147483648.0 && x < 2147483647.0, 1)) return (int32_t)x;
  asm volatile("" ::: "memory");
#endif
  return (int32_t)(uint32_t)(uint64_t)za_f2i64(x);
}
ZA_FN double za_or(double a, double b) { return (double)(za_i32(a) | za_i32(b)); }
ZA_FN double za_and(double a, double b) { return (double)(za_i32(a) & za_i32(b)); }
ZA_FN double za_xor(double a, double b) { return (double)(za_i32(a) ^ za_i32(b)); }
ZA_FN double za_shl(double a, double b) { return (double)(int32_t)((uint32_t)za_i32(a) << (za_i32(b) & 31)); }
ZA_FN double za_shr(double a, double b) { return (double)(za_i32(a) >> (za_i32(b) & 31)); }
ZA_FN double za_mod(double a, double b) {
  int32_t l = za_i32(a), r = za_i32(b);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  // The device has no integer divide: l % r is ~40 instructions (0.15 us per frame for one ring counter, tools/ring_io.py).
  // Ring indices are `(pos + 1) % N` and `x % 2^k`: both have exact shortcuts (same result as srem for 0 <= l, 0 < r).
  if (l >= 0 && r > 0) {
    if ((uint32_t)l < (uint32_t)r) return (double)l;
    if (((uint32_t)r & ((uint32_t)r - 1u)) == 0u) return (double)(l & (r - 1));
    if (l == r) return 0.0;
    asm volatile("" ::: "memory");                         // (keeps the division below a branch, not a select)
  }
#endif
  if (r == 0 || (l == INT32_MIN && r == -1)) return 0.0;  // srem traps/UB in the reference; defined as 0 here
  return (double)(l % r);
}
ZA_FN double za_min(double a, double b) { return a < b ? a : b; }   // olt select: NaN -> second operand
ZA_FN double za_max(double a, double b) { return a > b ? a : b; }
ZA_FN double za_sign(double a) { return a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0); }
ZA_FN double za_sqr(double a) { return a * a; }
ZA_FN double za_invsqrt(double a) {
  float f = (float)a;
  int32_t bits;
  __builtin_memcpy(&bits, &f, 4);
  bits = (int32_t)0x5f3759df - (bits >> 1);
  float y;
  __builtin_memcpy(&y, &bits, 4);
  double y0 = (double)y;
  return y0 * (1.5 - (0.5 * a) * (y0 * y0));
}
#if defined(__HIPCC__)
// Conversions whose result is only meaningful below 2^31 (arena addresses: mem_cap < 2^31 - 1024, zabatch.hip; loop counts:
// ZA_LOOP_CAP = 2^26): clamp in f64, then ONE 32-bit convert, no branch. NaN and negatives end at -1 -> 0 like the
// reference's clamp; anything at or above `hi` stays out of range for the caller's own check (arena bound / loop cap).
ZA_FN int64_t za_f2i_low(double x, double hi) {
  const int32_t a = (int32_t)fmin(fmax(x, -1.0), hi);
  return a < 0 ? 0 : a;
}
ZA_FN int64_t za_loopcount(double n) { return za_f2i_low(n, 134217728.0); }
#else
ZA_FN int64_t za_loopcount(double n) { int64_t c = za_f2i64(n); return c < 0 ? 0 : c; }
#endif

// ---------------------------------------------------------------------------------------------
// accumulation loops shared by the replica lanes of an instance (zajit/emit.py e_Loop)
// ---------------------------------------------------------------------------------------------
ZA_FN bool za_coop_int(double x) { return x == floor(x) && fabs(x) < 1.0e15; }   // counters stay exact under i0 + k * step
// Fewest trips (elements) for which a loop (memcpy / memset) is shared by the replica lanes: two per lane, but no more than 64 --
// with one instance per wavefront (64 replica lanes) a 100-tap FIR must not fall back to 64 identical copies of the serial loop
// (DOT at 512 instances, one per wavefront: 462 ms against 52 ms at two).
template <class S>
ZA_FN int64_t za_coop_min(const S& s) { return s.rep_n >= 32u ? 64 : 2 * (int64_t)s.rep_n; }
#if defined(__HIPCC__)
#define ZA_COOP_ON(s) ((s).rep_n > 1u)
template <class S>
ZA_FN double za_coop_sum(S& s, double x) {          // every replica lane ends with the same bits (fp addition commutes)
  for (uint32_t off = s.rep_stride; off < 64u; off <<= 1) x += __shfl_xor(x, (int)off, 64);
  return x;
}
#else
#define ZA_COOP_ON(s) false
template <class S>
ZA_FN double za_coop_sum(S& s, double x) { (void)s; return x; }
#endif

// Elementwise ("map") loops shared by the replica lanes of an instance (zajit/emit.py _map_plan): one row per arena access of
// a trip. kind 0 load / 1 store at a0 + sig * trip; kind 2 a load somewhere in [a0, a0 + ext).
struct ZaMapAcc { double a0, ext; int32_t sig, kind, ord; };   // ord: position of the access within a trip
// Are the c trips independent -- no store of one trip at an address another trip touches? Conservative: a store and another
// access either never meet (disjoint address ranges), or walk in step and meet only inside one trip (same stride, first
// addresses equal) or never (first addresses differ by less than a multiple of the stride: re / im interleave).
ZA_FN bool za_map_ok(const ZaMapAcc* A, int n, int64_t c) {
  if (c < 2) return false;
  double lo[24], hi[24];
  for (int j = 0; j < n; ++j) {
    if (!za_coop_int(A[j].a0)) return false;
    if (A[j].kind == 2) {
      if (!za_coop_int(A[j].ext) || A[j].ext < 1.0) return false;
      lo[j] = A[j].a0; hi[j] = A[j].a0 + A[j].ext - 1.0;
    } else {
      const double e = A[j].a0 + (double)A[j].sig * (double)(c - 1);
      lo[j] = A[j].a0 < e ? A[j].a0 : e; hi[j] = A[j].a0 < e ? e : A[j].a0;
    }
    if (lo[j] < 0.0 || hi[j] > 2147483000.0) return false;
  }
  for (int j = 0; j < n; ++j) {
    if (A[j].kind != 1) continue;
    if (A[j].sig == 0) return false;                         // every trip stores to the same cell
    for (int k = 0; k < n; ++k) {
      if (k == j) continue;
      if (hi[j] < lo[k] || hi[k] < lo[j]) continue;          // never meet
      if (A[k].kind == 2 || A[k].sig != A[j].sig) return false;
      const double d = A[k].a0 - A[j].a0;
      if (d == 0.0) {                                        // the same cell, in the same trip only
        if (A[k].kind == 0 && A[k].ord > A[j].ord) return false;   // ... read back after the store
        continue;
      }
      if (fmod(d, (double)A[j].sig) != 0.0) continue;        // interleaved, never the same cell
      return false;
    }
  }
  return true;
}
// trips of `while (v < bound) ( ...; v += step; )`, step > 0
ZA_FN int64_t za_map_trips(double bound, double v, double step) {
  const double d = (bound - v) / step;
  if (!(d > 0.0)) return 0;
  if (!(d < 1.0e9)) return (int64_t)1 << 40;
  return (int64_t)ceil(d);
}
#if defined(__HIPCC__)
// after the shared trips: what the replica lanes stored becomes visible to each other, and the per-lane bookkeeping of those
// stores (high-water mark, overflow report) is merged so that every lane -- the primary one, which is written back -- has it
template <class S>
ZA_FN void za_map_sync(S& s) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (uint32_t off = s.rep_stride; off < 64u; off <<= 1) {
    const int64_t h = __shfl_xor(s.mem_high, (int)off, 64), nd = __shfl_xor(s.mem_need, (int)off, 64);
    const uint32_t e = (uint32_t)__shfl_xor((int)s.err, (int)off, 64);
    s.mem_high = h > s.mem_high ? h : s.mem_high;
    s.mem_need = nd > s.mem_need ? nd : s.mem_need;
    s.err |= e;
  }
}
#else
template <class S>
ZA_FN void za_map_sync(S& s) { (void)s; }
#endif

// ---------------------------------------------------------------------------------------------
// mem[]
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
ZA_FN int64_t za_addr1(double x) { return za_f2i_low(x + 1.0e-5, 2147483520.0); }
#else
ZA_FN int64_t za_addr1(double x) { int64_t a = za_f2i64(x + 1.0e-5); return a < 0 ? 0 : a; }
#endif
ZA_FN int64_t za_addr(double base, double idx) { return za_addr1(base + idx); }

// LDS WINDOW (device, leaves whose only mem[] users are the functions of this section): a serial script is bound by the
// latency of its dependent mem[] accesses -- ~1 us each from HBM/L2 against ~0.1 us from LDS. When every instance's
// arena footprint after @init fits, the process kernel keeps mem[0, lm_words) of its instances in LDS for the whole
// launch (zab_generic.hip.h loads it on entry and writes the stored part back on exit); addresses above it, and every
// other kernel, use the arena in HBM. Addresses are >= 0 here (za_addr1 clamps).
#if defined(__HIPCC__) && defined(ZA_USES_LMEM) && ZA_USES_LMEM
extern __shared__ double za_lmem[];
#define ZA_LM_HIT(s, a) (S::kLm && (uint64_t)(a) < (uint64_t)(s).lm_words)
#define ZA_LM_REF(s, a) za_lmem[(uint32_t)(a) * (s).lm_stride + (s).lm_off]
#else
#define ZA_LM_HIT(s, a) false
#define ZA_LM_REF(s, a) (s).sink
#endif

// The arena load is unconditional (address clamped into the arena, result discarded when out of range): without a
// branch per access the compiler can issue the loads of neighbouring iterations of a read-only loop together instead of
// paying one memory latency per iteration.
template <class S>
ZA_FN double za_ld(S& s, int64_t a) {
  if (ZA_LM_HIT(s, a)) return ZA_LM_REF(s, a);
#if defined(ZA_LD_BRANCH)
  return a < s.mem_cap ? s.mem[a * s.mem_stride] : 0.0;
#else
  const bool in = a < s.mem_cap;
  const double v = s.mem[(in ? a : 0) * s.mem_stride];
  return in ? v : 0.0;
#endif
}

template <class S>
ZA_FN void za_note_store(S& s, int64_t end) {
  if (end > s.mem_high) s.mem_high = end;
  if (end > s.mem_cap) { s.err |= ZA_ERR_MEM_OVERFLOW; if (end > s.mem_need) s.mem_need = end; }
}
template <class S>
ZA_FN double za_st(S& s, int64_t a, double v) {
  za_note_store(s, a + 1);
  if (ZA_LM_HIT(s, a)) { ZA_LM_REF(s, a) = v; return v; }
  if (a < s.mem_cap && !s.replica) s.mem[a * s.mem_stride] = v;   // (replica lanes would store the same value again)
  return v;
}

template <class S>
ZA_FN double* za_mem_ptr(S& s, int64_t a) {    // lvalue for builtins with output arguments
  za_note_store(s, a + 1);
  if (ZA_LM_HIT(s, a)) return (double*)&ZA_LM_REF(s, a);
  return a < s.mem_cap ? &s.mem[a * s.mem_stride] : &s.sink;
}

// slider(i) 1-based, spl(i) 0-based; out of range reads 0, writes are ignored (value still returned).
ZA_FN double za_dyn_ld64(const double* arr, double i, int off) {
  int64_t k = za_f2i64(i + 1.0e-5) - off;
  return (k >= 0 && k < 64) ? arr[k] : 0.0;
}
ZA_FN void za_dyn_st64(double* arr, double i, int off, double v) {
  int64_t k = za_f2i64(i + 1.0e-5) - off;
  if (k >= 0 && k < 64) arr[k] = v;
}

template <class S>
ZA_FN double za_memset(S& s, double dest, double value, double len) {
  int64_t d = za_addr1(dest);
  int64_t n = za_f2i64(len);
  if (n < 0) n = 0;
  if (n > 0) za_note_store(s, d + n);
  int64_t e = d + n;
  if (e > s.mem_cap) e = s.mem_cap;
  int64_t i = d;
#if defined(__HIPCC__)
  if (ZA_COOP_ON(s) && e - d >= za_coop_min(s)) {    // the instance's replica lanes share the range
    for (i = d + s.rep_i; i < e; i += s.rep_n) {
      if (ZA_LM_HIT(s, i)) ZA_LM_REF(s, i) = value; else s.mem[i * s.mem_stride] = value;
    }
    za_map_sync(s);
    return dest;
  }
#endif
  for (; i < e && ZA_LM_HIT(s, i); ++i) ZA_LM_REF(s, i) = value;
  for (; i < e; ++i) s.mem[i * s.mem_stride] = value;
  return dest;
}

ZA_FN int64_t za_round_idx(double v) { return za_f2i64(v + (v >= 0.0 ? 1.0e-5 : -1.0e-5)); }

template <class S>
ZA_FN double za_memcpy(S& s, double destD, double srcD, double lenD) {
  int64_t d = za_round_idx(destD), r = za_round_idx(srcD), n = za_round_idx(lenD);
  if (d < 0) d = 0;
  if (r < 0) r = 0;
  if (n <= 0) return 0.0;
  if (d + n < d || r + n < r) return 0.0;
  int64_t need = (d + n > r + n) ? d + n : r + n;
  if (need > s.mem_cap) {           // the reference grows here; a fixed arena cannot -> fail loudly, copy nothing
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if (need > s.mem_need) s.mem_need = need;
    return 0.0;
  }
  if (d + n > s.mem_high) s.mem_high = d + n;
  const int64_t st = s.mem_stride;
#define ZA_CP1(i) do { const double x_ = ZA_LM_HIT(s, r + (i)) ? ZA_LM_REF(s, r + (i)) : s.mem[(r + (i)) * st];                    \
                       if (ZA_LM_HIT(s, d + (i))) ZA_LM_REF(s, d + (i)) = x_; else s.mem[(d + (i)) * st] = x_; } while (0)
#if defined(__HIPCC__)
  if (ZA_COOP_ON(s) && n >= za_coop_min(s) && d != r) {
    // The replica lanes share the range, eight elements per lane and batch, ALL of a batch's loads before its stores (they
    // are one wave: the stores wait for every lane's loads). That is also what makes overlapping ranges safe: moving down
    // (d < r) the batches ascend, so a cell is always read by the batch that owns it or an earlier one before the batch
    // that overwrites it stores -- the serial memmove order; moving up (d > r) they descend. (Trip by trip the device
    // compiler must assume that a store aliases the next load and pays a memory latency per element: an STFT's
    // `memcpy(ola, ola + HOP, N - HOP)` was most of its time.)
    const int64_t R = s.rep_n, B = 8 * R, nb = (n + B - 1) / B;
    for (int64_t q = 0; q < nb; ++q) {
      const int64_t b0 = (d < r ? q : nb - 1 - q) * B + s.rep_i;
      double v_[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = b0 + u * R, ic = i < n ? i : 0;
        v_[u] = ZA_LM_HIT(s, r + ic) ? ZA_LM_REF(s, r + ic) : s.mem[(r + ic) * st];
      }
      // (every load of the batch has returned before its first store is issued -- not just each store's own)
      asm volatile("" : "+v"(v_[0]), "+v"(v_[1]), "+v"(v_[2]), "+v"(v_[3]), "+v"(v_[4]), "+v"(v_[5]), "+v"(v_[6]), "+v"(v_[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = b0 + u * R;
        if (i < n) { if (ZA_LM_HIT(s, d + i)) ZA_LM_REF(s, d + i) = v_[u]; else s.mem[(d + i) * st] = v_[u]; }
      }
    }
    za_map_sync(s);
    return 0.0;
  }
#endif
  if (d <= r) for (int64_t i = 0; i < n; ++i) ZA_CP1(i);
  else for (int64_t i = n - 1; i >= 0; --i) ZA_CP1(i);
#undef ZA_CP1
  return 0.0;
}

// ---------------------------------------------------------------------------------------------
// rand()
// ---------------------------------------------------------------------------------------------
// The seeding / 624-word refill is the out-of-line part (by value: no state object behind it); the draw itself is inline.
inline ZA_NOINLINE void za_mt_refill(uint32_t* mt, int64_t st, uint32_t i) {
  const int N = 624, M = 397;
  if (i == 0) {
    uint32_t prev = 0x4141F00Du;
    mt[0] = prev;
    for (int k = 1; k < N; ++k) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k; mt[k * st] = prev; }
  }
  for (int k = 0; k < N; ++k) {
    uint32_t a = mt[k * st], b = mt[((k + 1) % N) * st];
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    mt[k * st] = mt[((k + M) % N) * st] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
  }
}
template <class S>
ZA_FN uint32_t za_mt_next(S& s) {
  uint32_t i = s.mti;
  if (i == 0 || i >= 624u) {
    za_mt_refill(s.mt, s.mt_stride, i);
    i = 0;
  }
  s.mti = i + 1;
  uint32_t y = s.mt[i * s.mt_stride];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}
template <class S>
ZA_FN double za_rand(S& s, double mx) {
  double m = floor(mx);
  if (m < 1.0) m = 1.0;
  return ((double)za_mt_next(s) * (1.0 / 4294967295.0)) * m;
}

// ---------------------------------------------------------------------------------------------
// slider bookkeeping
// ---------------------------------------------------------------------------------------------
ZA_FN int64_t za_change_mask(double m) {
  // (m > 0) ? llround(m) : 0, non-positive results select nothing (src/JSFXJuceProcessor.cpp:2448-2475)
  if (!(m > 0.0)) return 0;
  if (m >= 9.2233720368547758e18) return 0;
  int64_t k = (int64_t)(m + 0.5);
  return k > 0 ? k : 0;
}
template <class S> ZA_FN double za_sliderchange(S& s, double m) {
  int64_t k = za_change_mask(m);
  if (k <= 0) return 0.0;
  s.pend_change |= (uint64_t)k;
  return 1.0;
}
template <class S> ZA_FN double za_slider_automate(S& s, double m, double end) {
  int64_t k = za_change_mask(m);
  if (k <= 0) return 0.0;
  s.pend_change |= (uint64_t)k;
  if (end != 0.0) s.pend_automate_end |= (uint64_t)k; else s.pend_automate |= (uint64_t)k;
  return 1.0;
}
ZA_FN uint64_t za_mask_arg(double m) {     // slider_show: fptoui(max(m, 0))
  if (!(m > 0.0)) return 0;
  if (m >= 18446744073709551615.0) return ~0ull;
  return (uint64_t)m;
}
template <class S> ZA_FN void za_vis_init(S& s) { if (!s.vis_init) { s.vis_mask = ~0ull; s.vis_init = 1; } }
template <class S> ZA_FN double za_slider_show1(S& s, double m) {
  za_vis_init(s);
  uint64_t k = (m < 0.0) ? 0 : za_mask_arg(m);
  return (double)(s.vis_mask & k);
}
template <class S> ZA_FN double za_slider_show2(S& s, double m, double mode) {
  za_vis_init(s);
  uint64_t k = (m < 0.0) ? 0 : za_mask_arg(m);
  if (mode == 0.0) s.vis_mask &= ~k;
  else if (mode == -1.0) s.vis_mask ^= k;
  else s.vis_mask |= k;
  return (double)(s.vis_mask & k);
}
template <class S> ZA_FN double za_slider_next_chg(S& s, double idx, double* out) {
  // reference: writes the current value, reports no further change point (src/JSFXJuceProcessor.cpp:137-152)
  int32_t k = (int32_t)za_f2i64(idx + 1.0e-5) - 1;
  if (out) *out = (k >= 0 && k < 64) ? s.sl[k] : 0.0;
  return 0.0;
}
template <class S> ZA_FN double za_unsupported(S& s) { s.err |= ZA_ERR_UNSUPPORTED; return 0.0; }
