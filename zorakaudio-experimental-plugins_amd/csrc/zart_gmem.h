// zart_gmem.h -- gmem[] and gmem_* builtins for generated section code.
//
// Restates the reference's shared-cell semantics (not its shared-memory plumbing):
//   cell index ............ floor(idx + 1e-5); non-finite or <= 0 -> 0 ............ src/DspJsfxGmem.cpp:67-77
//   load / store .......... relaxed 64-bit cells holding the double's bit pattern; out of range load -> 0, store
//                           ignored and returns 0; an in-range store returns the value, bumps the page's sequence,
//                           records the writer id and bumps the global sequence ..... src/DspJsfxGmem.cpp:178-207
//   bulk get/put/fill/zero/copy: integer args = llround + clamp to int; negative / out-of-range starts -> 0 cells;
//                           count clipped to the segment; one page bump per page touched .. DspJsfxGmem.cpp:209-309,
//                           src/DspJsfxRuntime.cpp:95-104,533-556
//   gmem_seq(page) ........ page < 0 -> global sequence ........................... src/DspJsfxGmem.cpp:311-318
//   gmem_size / gmem_page / gmem_attach(_size) ..................................... src/DspJsfxRuntimeBuiltins.cpp:142-231
// Default segment: 1 Mi cells in pages of 1024 (src/DspJsfxGmem.h:17-18).
//
// Placement: one segment per engine (= per GPU), shared by all its instances -- the reference's rule that instances
// sharing a namespace must see one memory is met by co-location. Segment names are not distinguished:
// every gmem_attach() of an engine's instances lands on the engine's segment.
#pragma once

#include "zart.h"

#define ZA_GMEM_DEFAULT_CELLS (1024ull * 1024ull)
#define ZA_GMEM_PAGE_CELLS 1024ull

struct ZaGmemView {
  unsigned long long* cells;        // [cell_count] bit patterns
  unsigned long long* page_seq;     // [page_count]
  unsigned long long* page_writer;  // [page_count]
  unsigned long long* global_seq;   // [1]
  uint64_t cell_count;
  uint64_t page_count;
};

#if defined(__HIPCC__)
#define ZA_ALOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZA_ASTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZA_AADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define ZA_ALOAD(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#define ZA_ASTORE(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define ZA_AADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#endif

ZA_FN uint64_t za_gmem_cell(double idx) {
  if (!(idx > 0.0) || !(idx < 1.0e300)) return 0;          // NaN, +-inf, <= 0
  const double t = floor(idx + 1.0e-5);
  if (t <= 0.0) return 0;
  if (t >= 18446744073709551615.0) return ~0ull;
  return (uint64_t)t;
}
ZA_FN int32_t za_gmem_int(double v) {                        // clampIntArg: llround with int saturation, NaN -> 0
  if (!(v == v) || v > 1.0e300 || v < -1.0e300) return 0;
  if (v <= -2147483648.0) return INT32_MIN;
  if (v >= 2147483647.0) return INT32_MAX;
  return (int32_t)(v < 0.0 ? -(double)(int64_t)(-v + 0.5) : (double)(int64_t)(v + 0.5));
}
ZA_FN double za_bits2d(unsigned long long b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
ZA_FN unsigned long long za_d2bits(double d) { unsigned long long b; __builtin_memcpy(&b, &d, 8); return b; }

template <class S> ZA_FN const ZaGmemView* za_gmem_view(S& s) { return (s.gmem && s.gmem_attached) ? s.gmem : nullptr; }

// One store = one bump of its page's sequence and of the global one. On the device the lanes of a wavefront that store in the same
// instruction (the lane-per-instance kernels: 64 instances running the same `gmem[k] = v`) add their counts TOGETHER: one atomic
// of popcount(lanes) per wavefront instead of 64 of 1 on the same word. The sums -- all a sequence is -- are the same; nothing can
// observe the order of relaxed increments. Measured on 3DPannerManager (256 instances, ~800 stores per block each, every one of
// them on the same two words): 1121 -> 907 ms per 48 000 frames; what remains is the latency of ~10 000 dependent L2 round trips per block and lane.
template <class S> ZA_FN void za_gmem_bump(S& s, const ZaGmemView* g, uint64_t page) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long act = __ballot(1);                                  // the lanes storing in this instruction
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int lead = (int)__ffsll((long long)act) - 1;
  const unsigned long long n = (unsigned long long)__popcll(act);
  const uint64_t page0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(page >> 32)) << 32)
                         | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)page);
  const bool same = __ballot(page == page0) == act;                            // (wave-uniform)
  if (same) {
    if (lane == lead) {
      if (page < g->page_count) {
        ZA_ASTORE(&g->page_writer[page], (unsigned long long)s.instance_id);   // (one of the writers: the reference keeps the last)
        ZA_AADD(&g->page_seq[page], n);
      }
      ZA_AADD(g->global_seq, n);
    }
    return;
  }
  if (page < g->page_count) {
    ZA_ASTORE(&g->page_writer[page], (unsigned long long)s.instance_id);
    ZA_AADD(&g->page_seq[page], 1ull);
  }
  if (lane == lead) ZA_AADD(g->global_seq, n);
#else
  if (page < g->page_count) {
    ZA_ASTORE(&g->page_writer[page], (unsigned long long)s.instance_id);
    ZA_AADD(&g->page_seq[page], 1ull);
  }
  ZA_AADD(g->global_seq, 1ull);
#endif
}

template <class S> ZA_FN double za_gmem_load(S& s, double idx) {
  const ZaGmemView* g = za_gmem_view(s);
  if (!g) return 0.0;
  const uint64_t c = za_gmem_cell(idx);
  return c < g->cell_count ? za_bits2d(ZA_ALOAD(&g->cells[c])) : 0.0;
}
template <class S> ZA_FN double za_gmem_store(S& s, double idx, double v) {
  const ZaGmemView* g = za_gmem_view(s);
  if (!g) return 0.0;
  const uint64_t c = za_gmem_cell(idx);
  if (c >= g->cell_count) return 0.0;
  ZA_ASTORE(&g->cells[c], za_d2bits(v));
  za_gmem_bump(s, g, c / ZA_GMEM_PAGE_CELLS);
  return v;
}

template <class S> ZA_FN double za_gmem_attach(S& s, double /*nameHandle*/) {
  if (!s.gmem) return 0.0;
  s.gmem_attached = 1;
  return 1.0;
}
template <class S> ZA_FN double za_gmem_attach_size(S& s, double nameHandle, double /*cells*/) { return za_gmem_attach(s, nameHandle); }
template <class S> ZA_FN double za_gmem_size(S& s) {
  const ZaGmemView* g = za_gmem_view(s);
  return g ? (double)g->cell_count : 0.0;
}
template <class S> ZA_FN double za_gmem_page(S& /*s*/, double idx) { return (double)(za_gmem_cell(idx) / ZA_GMEM_PAGE_CELLS); }
template <class S> ZA_FN double za_gmem_seq(S& s, double page) {
  const ZaGmemView* g = za_gmem_view(s);
  if (!g) return 0.0;
  const int32_t p = za_gmem_int(page);
  if (p < 0) return (double)ZA_ALOAD(g->global_seq);
  return (uint64_t)p < g->page_count ? (double)ZA_ALOAD(&g->page_seq[p]) : 0.0;
}

// mem[dstBase ..] <- cells[srcIdx ..]
template <class S> ZA_NOINLINE double za_gmem_get_o(S& s, double dstBaseD, double srcIdxD, double countD) {
  const ZaGmemView* g = za_gmem_view(s);
  const int32_t dst = za_gmem_int(dstBaseD), src = za_gmem_int(srcIdxD), cnt = za_gmem_int(countD);
  if (!g || cnt <= 0 || dst < 0 || src < 0 || (uint64_t)src >= g->cell_count) return 0.0;
  uint64_t n = (uint64_t)cnt;
  if (n > g->cell_count - (uint64_t)src) n = g->cell_count - (uint64_t)src;
  if ((int64_t)dst + (int64_t)n > s.mem_cap) {             // the reference grows mem here
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if ((int64_t)dst + (int64_t)n > s.mem_need) s.mem_need = (int64_t)dst + (int64_t)n;
    return 0.0;
  }
  za_note_store(s, (int64_t)dst + (int64_t)n);
  for (uint64_t i = 0; i < n; ++i) s.mem[((int64_t)dst + (int64_t)i) * s.mem_stride] = za_bits2d(ZA_ALOAD(&g->cells[(uint64_t)src + i]));
  return (double)(int32_t)n;
}
template <class S> ZA_FN double za_gmem_get(S& s, double dstBaseD, double srcIdxD, double countD) { ZA_OUTCALL(za_gmem_get_o(e, dstBaseD, srcIdxD, countD)); }
// cells[dstIdx ..] <- mem[srcBase ..]
template <class S> ZA_NOINLINE double za_gmem_put_o(S& s, double dstIdxD, double srcBaseD, double countD) {
  const ZaGmemView* g = za_gmem_view(s);
  const int32_t dst = za_gmem_int(dstIdxD), src = za_gmem_int(srcBaseD), cnt = za_gmem_int(countD);
  if (!g || cnt <= 0 || dst < 0 || src < 0) return 0.0;
  // reference: src + n must lie inside the *allocated* heap (memN); the fixed arena's capacity plays that role
  if ((int64_t)src + (int64_t)cnt > s.mem_cap || (uint64_t)dst >= g->cell_count) return 0.0;
  uint64_t n = (uint64_t)cnt;
  if (n > g->cell_count - (uint64_t)dst) n = g->cell_count - (uint64_t)dst;
  uint64_t last = ~0ull;
  for (uint64_t i = 0; i < n; ++i) {
    ZA_ASTORE(&g->cells[(uint64_t)dst + i], za_d2bits(s.mem[((int64_t)src + (int64_t)i) * s.mem_stride]));
    const uint64_t pg = ((uint64_t)dst + i) / ZA_GMEM_PAGE_CELLS;
    if (pg != last) { za_gmem_bump(s, g, pg); last = pg; }
  }
  return (double)(int32_t)n;
}
template <class S> ZA_FN double za_gmem_put(S& s, double dstIdxD, double srcBaseD, double countD) { ZA_OUTCALL(za_gmem_put_o(e, dstIdxD, srcBaseD, countD)); }
template <class S> ZA_NOINLINE double za_gmem_fill_o(S& s, double dstIdxD, double value, double countD) {
  const ZaGmemView* g = za_gmem_view(s);
  const int32_t dst = za_gmem_int(dstIdxD), cnt = za_gmem_int(countD);
  if (!g || cnt <= 0 || dst < 0 || (uint64_t)dst >= g->cell_count) return 0.0;
  uint64_t n = (uint64_t)cnt;
  if (n > g->cell_count - (uint64_t)dst) n = g->cell_count - (uint64_t)dst;
  const unsigned long long bits = za_d2bits(value);
  uint64_t last = ~0ull;
  for (uint64_t i = 0; i < n; ++i) {
    ZA_ASTORE(&g->cells[(uint64_t)dst + i], bits);
    const uint64_t pg = ((uint64_t)dst + i) / ZA_GMEM_PAGE_CELLS;
    if (pg != last) { za_gmem_bump(s, g, pg); last = pg; }
  }
  return (double)(int32_t)n;
}
template <class S> ZA_FN double za_gmem_fill(S& s, double dstIdxD, double value, double countD) { ZA_OUTCALL(za_gmem_fill_o(e, dstIdxD, value, countD)); }
template <class S> ZA_FN double za_gmem_zero(S& s, double dstIdxD, double countD) { return za_gmem_fill(s, dstIdxD, 0.0, countD); }
// cells[dst ..] <- cells[src ..] as if through a temporary (overlap-safe)
template <class S> ZA_NOINLINE double za_gmem_copy_o(S& s, double dstIdxD, double srcIdxD, double countD) {
  const ZaGmemView* g = za_gmem_view(s);
  const int32_t dst = za_gmem_int(dstIdxD), src = za_gmem_int(srcIdxD), cnt = za_gmem_int(countD);
  if (!g || cnt <= 0 || dst < 0 || src < 0 || (uint64_t)dst >= g->cell_count || (uint64_t)src >= g->cell_count) return 0.0;
  uint64_t n = (uint64_t)cnt;
  if (n > g->cell_count - (uint64_t)dst) n = g->cell_count - (uint64_t)dst;
  if (n > g->cell_count - (uint64_t)src) n = g->cell_count - (uint64_t)src;
  uint64_t last = ~0ull;
  const bool backward = (uint64_t)dst > (uint64_t)src;    // memmove direction gives the temporary-copy result
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t i = backward ? n - 1 - j : j;
    ZA_ASTORE(&g->cells[(uint64_t)dst + i], ZA_ALOAD(&g->cells[(uint64_t)src + i]));
  }
  for (uint64_t i = 0; i < n; ++i) {                       // page bumps in ascending order, once per page
    const uint64_t pg = ((uint64_t)dst + i) / ZA_GMEM_PAGE_CELLS;
    if (pg != last) { za_gmem_bump(s, g, pg); last = pg; }
  }
  return (double)(int32_t)n;
}
template <class S> ZA_FN double za_gmem_copy(S& s, double dstIdxD, double srcIdxD, double countD) { ZA_OUTCALL(za_gmem_copy_o(e, dstIdxD, srcIdxD, countD)); }
