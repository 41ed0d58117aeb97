// zart_fft.h -- JSFX FFT builtins for generated section code: fft / ifft / fft_real / ifft_real / fft_permute /
// fft_ipermute / convolve_c over mem[], with the reference's conventions (not its code):
//   * argument handling, size limits 16..32768, "region must not cross a 65536-double page, else silently do nothing",
//     overlap-safe convolve_c ................................. src/JSFXJuceProcessor.cpp:1076-1447
//   * WDL_fft(buf, N, 0): unscaled forward DFT (e^-i), result stored in WDL_fft_permute order; WDL_fft(buf, N, 1):
//     input in that order, unscaled inverse (e^+i), natural-order output ............. src/WDL/fft.h:55-57
//   * permutation: position i holds natural bin (N - f(i)) mod N with f the DJBFFT split-radix frequency recursion
//     (src/WDL/fft.c:990-1019); fft_permute gathers natural[k] = buf[perm[k]] (JSFXJuceProcessor.cpp:1230-1264)
//   * WDL_real_fft: N reals -> N/2 packed bins in permute order of N/2, bins scaled by 2, DC in [0].re and Nyquist in
//     [0].im; inverse is the exact reverse, unscaled (so fft_real -> ifft_real multiplies by 2N) ... src/WDL/fft.h:59-63,
//     fft.c:1086-1176 (the "two for one" real transform)
// Known answers: tests/golden/wdl_fft.npz (produced by the reference WDL build).
//
// The transform itself is this repo's own: an iterative radix-2 decimation-in-time FFT (natural order), then a scatter
// into the WDL order. Two execution forms with identical arithmetic per butterfly (hence identical bits):
//   * wave-cooperative (device, n <= 4096 complex): the generic kernels run one INSTANCE per lane, so a call site is
//     reached by up to 64 lanes, each wanting its own transform. The lanes that reached the call take the requests one
//     by one (ballot / readlane broadcast of the requester's base, size and mem pointer) and work on each together:
//     the buffer is staged in a 64 KiB LDS array, every pass's n/2 butterflies are spread over the participating
//     lanes, and the result is written back in the order the builtin defines. fft / ifft / fft_permute / fft_ipermute.
//   * serial (CPU port; device for n > 4096 and for the real transforms): one lane, per-instance scratch in HBM.
// (A 64 x 64 DFT-as-GEMM on the f64 MFMA pipe was priced and rejected: 4 real 64^3 GEMMs per stage x 2 stages = 4.2 Mflop
// against 0.25 Mflop for the radix-2 FFT of 4096 points, for a matrix pipe that is only 2x the f64 vector rate.)
#pragma once

#include "zart.h"

#define ZA_FFT_MIN 16
#define ZA_FFT_MAX 32768
#define ZA_FFT_PAGE 65536

// cos/sin(2*pi*j/ZA_FFT_MAX), j < ZA_FFT_MAX/2; filled once per process by za_fft_table_init (host) / init kernel (device)
#if defined(__HIPCC__)
#define ZA_FFT_COOP_MAX 4096
// Complex points a wave keeps in LDS at a time (a module build knob). 1024 (default) = 16 KiB + 8 KiB of twiddles, six
// wavefronts per CU; complex transforms of 2048 / 4096 points then run SLICED: 1024-point blocks through LDS, the last one
// or two radix-2 stages from registers, the pieces meeting in the instance's HBM scratch (za_fft_coop); real transforms
// above 2048 reals take the serial path. -DZA_FFT_LDS_POINTS=4096 = the whole of the largest cooperative transform in LDS
// (64 KiB + 32 KiB: ONE wavefront per CU; round 1's layout, kept as the fixtures' *_full builds). Every size any leaf of
// the catalog asks for (1024 complex, 2048 real) fits the small buffer. Measured (MI355X, 2048 buffers, fft + permute +
// ipermute + ifft): 1024 points 357 -> 126 us, 4096 points 1421 -> 940 us (tools/fft_bench.py).
#ifndef ZA_FFT_LDS_POINTS
#define ZA_FFT_LDS_POINTS 1024
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
extern "C" int zab_module_fft_lds_points(void) { return ZA_FFT_LDS_POINTS; }   // read by the runtime (wavefronts per batch)
#endif
__device__ double za_fft_cos[ZA_FFT_MAX / 2];
__device__ double za_fft_sin[ZA_FFT_MAX / 2];
__device__ double za_fft_twc[ZA_FFT_LDS_POINTS];     // (cos, sin)(2 pi j / ZA_FFT_LDS_POINTS), j < ZA_FFT_LDS_POINTS / 2
__device__ double za_fft_tw4[ZA_FFT_COOP_MAX];       // (cos, sin)(2 pi j / ZA_FFT_COOP_MAX), j < ZA_FFT_COOP_MAX / 2: the sliced transforms' second stage
#ifdef ZA_FFT_STAMPS
__device__ int za_fft_stamp_once;
#endif
__device__ uint16_t za_fft_perm[2 * ZA_FFT_COOP_MAX];     // [n + i] = natural bin stored at position i of an n-point transform
__device__ uint16_t za_fft_iperm[2 * ZA_FFT_COOP_MAX];    // [n + k] = position that holds natural bin k
#else
static double za_fft_cos[ZA_FFT_MAX / 2];
static double za_fft_sin[ZA_FFT_MAX / 2];
static void za_fft_table_init(void) {
  static int done;
  if (done) return;
  for (int j = 0; j < ZA_FFT_MAX / 2; ++j) {
    za_fft_cos[j] = cos(6.283185307179586476925286766559 * (double)j / (double)ZA_FFT_MAX);
    za_fft_sin[j] = sin(6.283185307179586476925286766559 * (double)j / (double)ZA_FFT_MAX);
  }
  done = 1;
}
#endif

// DJBFFT frequency of buffer position i in an n-point transform (iterative form of the split-radix recursion).
ZA_FN uint32_t za_fft_freq(uint32_t i, uint32_t n) {
  uint32_t mul = 1, add = 0;           // result = ((f(i', n') * mul) + add), accumulated from the outside in
  // f(i,n): n<=2 -> i;  i<n/2 -> 2 f(i, n/2);  i-n/2 < n/4 -> 4 f(., n/4) + 1;  else 4 f(., n/4) - 1   (mod n)
  const uint32_t mask = n - 1;
  while (n > 2) {
    uint32_t m = n >> 1;
    if (i < m) { mul <<= 1; n = m; continue; }
    i -= m; m >>= 1;
    if (i < m) { add += mul; mul <<= 2; n = m; continue; }
    i -= m;
    add -= mul; mul <<= 2; n = m;
  }
  return (i * mul + add) & mask;
}
ZA_FN uint32_t za_fft_bin_of_pos(uint32_t i, uint32_t n) { return (n - za_fft_freq(i, n)) & (n - 1); }
ZA_FN uint32_t za_bitrev(uint32_t v, int bits) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
  v = (v >> 16) | (v << 16);
  return v >> (32 - bits);
}
ZA_FN int za_log2(uint32_t n) { int b = 0; while ((1u << b) < n) ++b; return b; }

#if defined(__HIPCC__)
// fills the twiddle and permutation tables once per process (launched by the module before its first prepare)
extern "C" __global__ void za_fft_table_kernel() {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ZA_FFT_MAX / 2) {
    double sn, cs;
    sincos(6.283185307179586476925286766559 * (double)j / (double)ZA_FFT_MAX, &sn, &cs);
    za_fft_cos[j] = cs;
    za_fft_sin[j] = sn;
    if (j % (ZA_FFT_MAX / ZA_FFT_LDS_POINTS) == 0) {      // the in-LDS passes' twiddles, (cos, sin) side by side: 8 KB that stay in L1
      za_fft_twc[2 * (j / (ZA_FFT_MAX / ZA_FFT_LDS_POINTS))] = cs;
      za_fft_twc[2 * (j / (ZA_FFT_MAX / ZA_FFT_LDS_POINTS)) + 1] = sn;
    }
    if (j % (ZA_FFT_MAX / ZA_FFT_COOP_MAX) == 0) {
      za_fft_tw4[2 * (j / (ZA_FFT_MAX / ZA_FFT_COOP_MAX))] = cs;
      za_fft_tw4[2 * (j / (ZA_FFT_MAX / ZA_FFT_COOP_MAX)) + 1] = sn;
    }
  }
  if (j >= 2 && j < 2 * ZA_FFT_COOP_MAX) {      // j = n + i with n the largest power of two <= j (real transforms use sizes from 8)
    uint32_t n = 2;
    while (2 * n <= (uint32_t)j) n <<= 1;
    const uint32_t bin = za_fft_bin_of_pos((uint32_t)j - n, n);
    za_fft_perm[j] = (uint16_t)bin;
    za_fft_iperm[n + bin] = (uint16_t)((uint32_t)j - n);
  }
}
#endif

ZA_FN bool za_fft_pow2(int64_t n) { return n >= ZA_FFT_MIN && n <= ZA_FFT_MAX && (n & (n - 1)) == 0; }
ZA_FN bool za_fft_in_page(int64_t base, int64_t span) {
  if (base < 0 || span <= 0) return false;
  return (base / ZA_FFT_PAGE) == ((base + span - 1) / ZA_FFT_PAGE);
}

// Validate a region of `span` doubles at base; the reference grows mem here, a fixed arena reports overflow.
template <class S>
ZA_FN bool za_fft_region(S& s, double baseD, int64_t span, int64_t& base, int64_t scratch_need) {
  int64_t b = za_round_idx(baseD);
  if (b < 0) b = 0;
  if (!za_fft_in_page(b, span)) return false;
  if (b + span > s.mem_cap) {
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if (b + span > s.mem_need) s.mem_need = b + span;
    return false;
  }
  if (s.fft_cap < scratch_need) { s.err |= ZA_ERR_UNSUPPORTED; return false; }   // scratch not provisioned for this size
  if (b + span > s.mem_high) s.mem_high = b + span;
  base = b;
  return true;
}

#define ZA_M(a) s.mem[(a) * s.mem_stride]
#define ZA_F(a) s.fft[(a) * s.fft_stride]

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#ifndef ZA_FFT_WAVES_PER_EU
#define ZA_FFT_WAVES_PER_EU 2
#endif
// Twiddles of the in-LDS passes, (cos, sin)(2 pi j / ZA_FFT_LDS_POINTS), j < ZA_FFT_LDS_POINTS / 2. Round 3: read from the compact
// device table za_fft_twc (8 KB, resident in a CU's L1: every wavefront of the CU reads the same table) instead of a copy in each
// wavefront's LDS -- the copy was a third of a transform's LDS footprint (16 + 8 KB: six wavefronts per CU; 16 KB: two per SIMD,
// what the register cap allows), and a pass asks for its twiddles together with its points, before it needs either.
// -DZA_FFT_TW_LDS=1 keeps round 2's LDS copy (staged by the first cooperative transform of a workgroup; LDS is neither cleared nor
// private between launches, so every kernel of an FFT leaf resets the flag on entry -- ZA_KERNEL_ENTRY in zab_generic.hip.h).
#ifndef ZA_FFT_TW_LDS
#define ZA_FFT_TW_LDS 0
#endif
#if ZA_FFT_TW_LDS
__shared__ double za_fft_tw[ZA_FFT_LDS_POINTS];
__shared__ int za_fft_tw_ready;
#define ZA_KERNEL_ENTRY() do { za_fft_tw_ready = 0; __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define ZA_KERNEL_ENTRY() (void)0
#endif
enum { ZA_COOP_FFT = 0, ZA_COOP_IFFT = 1, ZA_COOP_PERMUTE = 2, ZA_COOP_IPERMUTE = 3, ZA_COOP_FFT_REAL = 4, ZA_COOP_IFFT_REAL = 5,
       ZA_COOP_CONVOLVE = 6,
       ZA_COOP_FFT_NAT = 7,      // fft(b, n); fft_permute(b, n)    -> natural order in, natural-order spectrum out
       ZA_COOP_IFFT_NAT = 8,     // fft_ipermute(b, n); ifft(b, n)  -> natural-order spectrum in, natural order out
       ZA_COOP_FFT_REAL_NAT = 9,     // fft_real(b, n); fft_permute(b, n / 2)    -> packed spectrum, bin k at position k
       ZA_COOP_IFFT_REAL_NAT = 10 }; // fft_ipermute(b, n / 2); ifft_real(b, n) -> the same layout in, n reals out
__device__ __forceinline__ int64_t za_readlane64(int64_t v, int l) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffff), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}
// The radix-2 passes of an nlp-point transform whose points sit bit-reversed in `buf` (LDS), butterflies spread over the nact
// participating lanes -- taken R passes at a time: a lane holds the 2^R points that R consecutive passes combine among
// themselves, runs those passes on them in registers and writes them back once. Every butterfly is the radix-2 one, with the
// twiddle and the order of operations the one-pass-at-a-time form uses, hence the same bits; what changes is the traffic: LDS
// is what bounds these transforms (a butterfly moves 80 bytes per pass for 10 flops), and a point now makes one round trip
// per three passes (46 reads and writes per 12 butterflies instead of 120) behind a third of the barriers.
// LDS layout of the transform buffer: element i (16 bytes) sits at ZA_B(i) doubles. The bank a wavefront's 16-lane phase hits is
// decided by the low four bits of the element index; the fused passes stride through the buffer so that, in the first two of
// them, those bits take only four values per phase (a lane owns 4 consecutive elements / lanes step 16 elements): four-way
// conflicts on every access, 65 % of the LDS pipeline's active cycles (SQ_LDS_BANK_CONFLICT, profiles/r03_fft_sq.txt). XOR-ing
// bits 4-5 of the index into bits 0-1 and 2-3 makes the low four bits a bijection of the lanes in every pass's pattern
// (varying bits {2..5}, {0,1,4,5}, {0..3}) and leaves each 16-element row a permutation of itself.
#ifndef ZA_FFT_NO_SWIZZLE
#define ZA_B(i) (2 * ((int)(i) ^ ((((int)(i) >> 4) & 3) * 5)))
#else
#define ZA_B(i) (2 * (int)(i))
#endif
template <int R>
__device__ __forceinline__ void za_fft_lds_pass(double* buf, const double* tw, int nlp, int h0, int sign, int rank, int nact) {
  constexpr int Q = 1 << R;                      // points per item
  constexpr int U = R == 3 ? 1 : (R == 2 ? 2 : 4);   // items per trip: all their LDS reads are issued before the first store
  const int items = nlp >> R;
  const int hb = 31 - __builtin_clz((unsigned)h0);
  for (int idx0 = rank; idx0 < items; idx0 += U * nact) {
    double xr[U][Q], xi[U][Q], wr[U][Q - 1], wi[U][Q - 1];
    int base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = idx0 + u * nact;
      const int ix = idx < items ? idx : 0;
      const int j = ix & (h0 - 1);
      const int b0 = ((ix >> hb) << (hb + R)) + j;        // points b0 + k * h0, k < Q
      base[u] = idx < items ? b0 : -1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int step = ZA_FFT_LDS_POINTS >> (hb + r + 1);       // pass r of this group: len = 2 * h0 << r
#pragma unroll
        for (int m = 0; m < (1 << r); ++m) {
          const int jr = j + m * h0;
          wr[u][(1 << r) - 1 + m] = tw[2 * jr * step];
          wi[u][(1 << r) - 1 + m] = tw[2 * jr * step + 1];
        }
      }
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        xr[u][k] = buf[ZA_B((b0 + k * h0))];
        xi[u][k] = buf[ZA_B((b0 + k * h0)) + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          if (k & (1 << r)) continue;
          const int c = k | (1 << r), t = (1 << r) - 1 + (k & ((1 << r) - 1));
          const double w_i = sign < 0 ? -wi[u][t] : wi[u][t];
          const double tr = xr[u][c] * wr[u][t] - xi[u][c] * w_i, ti = xr[u][c] * w_i + xi[u][c] * wr[u][t];
          const double ar = xr[u][k], ai = xi[u][k];
          xr[u][k] = ar + tr; xi[u][k] = ai + ti;
          xr[u][c] = ar - tr; xi[u][c] = ai - ti;
        }
      }
      if (base[u] >= 0) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          buf[ZA_B(base[u] + k * h0)] = xr[u][k];
          buf[ZA_B(base[u] + k * h0) + 1] = xi[u][k];
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The same R fused passes over the whole LDS buffer for the sliced transforms (za_fft_coop, nl > ZA_FFT_LDS_POINTS): the buffer
// holds a chunk of a larger transform, so which twiddle a butterfly takes is the caller's business -- twf(jl, lhb, wr, wi) with jl
// the upper point's offset inside its local group of 2 << lhb positions. Positions are swizzled by ZA_B2: low four bits XOR
// bits 4-7, a bijection of the lanes of a 16-lane phase whether they step through a row (these passes: h0 >= 16) or down a column.
#define ZA_B2(i) (2 * ((int)(i) ^ (((int)(i) >> 4) & 15)))
template <int R, class TW>
__device__ __forceinline__ void za_fft_lds_pass_v(double* buf, int h0, int sign, int rank, int nact, TW twf) {
  constexpr int Q = 1 << R;
  constexpr int U = R == 3 ? 1 : 2;
  constexpr int items = ZA_FFT_LDS_POINTS >> R;
  const int hb = 31 - __builtin_clz((unsigned)h0);
  for (int idx0 = rank; idx0 < items; idx0 += U * nact) {
    double xr[U][Q], xi[U][Q], wr[U][Q - 1], wi[U][Q - 1];
    int base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = idx0 + u * nact;
      const int ix = idx < items ? idx : 0;
      const int j = ix & (h0 - 1);
      const int b0 = ((ix >> hb) << (hb + R)) + j;        // points b0 + k * h0, k < Q
      base[u] = idx < items ? b0 : -1;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < (1 << r); ++m) twf(j + m * h0, hb + r, wr[u][(1 << r) - 1 + m], wi[u][(1 << r) - 1 + m]);
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        xr[u][k] = buf[ZA_B2(b0 + k * h0)];
        xi[u][k] = buf[ZA_B2(b0 + k * h0) + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          if (k & (1 << r)) continue;
          const int c = k | (1 << r), t = (1 << r) - 1 + (k & ((1 << r) - 1));
          const double w_i = sign < 0 ? -wi[u][t] : wi[u][t];
          const double tr = xr[u][c] * wr[u][t] - xi[u][c] * w_i, ti = xr[u][c] * w_i + xi[u][c] * wr[u][t];
          const double ar = xr[u][k], ai = xi[u][k];
          xr[u][k] = ar + tr; xi[u][k] = ai + ti;
          xr[u][c] = ar - tr; xi[u][c] = ai - ti;
        }
      }
      if (base[u] >= 0) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          buf[ZA_B2(base[u] + k * h0)] = xr[u][k];
          buf[ZA_B2(base[u] + k * h0) + 1] = xi[u][k];
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void za_fft_lds_stages(double* buf, const double* tw, int nlp, int sign, int rank, int nact) {
  int rem = 31 - __builtin_clz((unsigned)nlp), h = 1;
  while (rem > 0) {
#ifdef ZA_FFT_RADIX2_PASSES
    const int R = 1;
#else
    const int R = rem % 3 == 0 ? 3 : (rem == 1 ? 1 : 2);      // 10 passes = 2 + 2 + 3 + 3, 11 = 2 + 3 + 3 + 3
#endif
    if (R == 3) za_fft_lds_pass<3>(buf, tw, nlp, h, sign, rank, nact);
    else if (R == 2) za_fft_lds_pass<2>(buf, tw, nlp, h, sign, rank, nact);
    else za_fft_lds_pass<1>(buf, tw, nlp, h, sign, rank, nact);
    h <<= R;
    rem -= R;
  }
}

// Every lane that reached the builtin calls this (converged at the call site); `ok` says whether this lane has a valid
// request (base, n). Returns true for the lanes whose request was served here.
// Real transforms (n reals = n / 2 complex points, so n up to 2 * ZA_FFT_COOP_MAX) and convolve_c (n = complex pairs, base2 = its
// second operand) take the same route; the arithmetic per element is the serial form's, hence the same bits.
template <class S>
ZA_NOINLINE bool za_fft_coop(S& s, bool ok, int64_t base, int n, int op, int64_t base2 = 0) {
  __shared__ double buf[ZA_B(ZA_FFT_LDS_POINTS)];
#if ZA_FFT_TW_LDS
  double* const tw = za_fft_tw;
#else
  const double* const tw = za_fft_twc;
#endif
  const bool is_real = op == ZA_COOP_FFT_REAL || op == ZA_COOP_IFFT_REAL || op == ZA_COOP_FFT_REAL_NAT || op == ZA_COOP_IFFT_REAL_NAT;
  // what runs here: anything that fits the LDS buffer; complex transforms and permutations beyond it run sliced (below);
  // real transforms beyond it take the serial path
  const bool coop = ok && (op == ZA_COOP_CONVOLVE || (is_real ? n <= 2 * ZA_FFT_LDS_POINTS : n <= ZA_FFT_COOP_MAX));
  const bool mine = coop && !s.replica;            // replica lanes help with their primary's request, they add none
  const uint64_t active = __ballot(1);
  uint64_t todo = __ballot(mine);
  const int lane = (int)(threadIdx.x & 63);
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
  const int rank = __popcll(active & below), nact = __popcll(active);
  const bool is_nat = op == ZA_COOP_FFT_NAT || op == ZA_COOP_IFFT_NAT;
#if ZA_FFT_TW_LDS
  if (todo && (op == ZA_COOP_FFT || op == ZA_COOP_IFFT || is_real || is_nat) && za_fft_tw_ready != 1) {
    // twiddles of the largest cooperative size, staged once per workgroup launch (the HBM table is 1 us away per read)
    for (int j = rank; j < ZA_FFT_LDS_POINTS / 2; j += nact) {
      tw[2 * j] = za_fft_cos[j * (ZA_FFT_MAX / ZA_FFT_LDS_POINTS)];
      tw[2 * j + 1] = za_fft_sin[j * (ZA_FFT_MAX / ZA_FFT_LDS_POINTS)];
    }
    __builtin_amdgcn_wave_barrier();
    if (rank == 0) za_fft_tw_ready = 1;
    __builtin_amdgcn_wave_barrier();
  }
#endif
  while (todo) {
    const int l = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    double* const mp = (double*)za_readlane64((int64_t)(uintptr_t)s.mem, l);
    const int64_t ms = za_readlane64(s.mem_stride, l), bl = za_readlane64(base, l);
    const int nreq = __builtin_amdgcn_readlane(n, l);
    const int nl = is_real ? nreq >> 1 : nreq;                 // complex points of the transform
    const int bits = za_log2((uint32_t)nl);
#define ZA_G(a) mp[(bl + (a)) * ms]
    if (op == ZA_COOP_CONVOLVE) {
      // dest[i] *= src[i]; on overlap the reference multiplies by a copy of src taken first (src/JSFXJuceProcessor.cpp:1363-1369)
      const int64_t rl = za_readlane64(base2, l);
      double* const fp = (double*)za_readlane64((int64_t)(uintptr_t)s.fft, l);
      const int64_t fs = za_readlane64(s.fft_stride, l);
      const bool overlap = (bl < rl + 2 * (int64_t)nl) && (rl < bl + 2 * (int64_t)nl) && bl != rl;
      if (overlap) {
        for (int i = rank; i < 2 * nl; i += nact) fp[(int64_t)i * fs] = mp[(rl + i) * ms];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      // four pairs per lane and trip, all their loads before the first store (the compiler must assume the stores alias the
      // next loads: one memory latency per pair otherwise)
      for (int i0 = rank; i0 < nl; i0 += 4 * nact) {
        double ar[4], ai[4], br[4], bi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * nact < nl ? i0 + u * nact : i0;
          ar[u] = ZA_G(2 * i); ai[u] = ZA_G(2 * i + 1);
          br[u] = overlap ? fp[(int64_t)(2 * i) * fs] : mp[(rl + 2 * i) * ms];
          bi[u] = overlap ? fp[(int64_t)(2 * i + 1) * fs] : mp[(rl + 2 * i + 1) * ms];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * nact;
          if (i < nl) {
            ZA_G(2 * i) = ar[u] * br[u] - ai[u] * bi[u];
            ZA_G(2 * i + 1) = ar[u] * bi[u] + ai[u] * br[u];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (nl > ZA_FFT_LDS_POINTS) {
      // ---- sliced: the transform does not fit the LDS buffer (complex fft / ifft / permutations of 2048 or 4096 points) ----
      // Same butterflies, same twiddles, same order of operations per element as the in-LDS form, hence the same bits.
      double* const fp = (double*)za_readlane64((int64_t)(uintptr_t)s.fft, l);
      const int64_t fs = za_readlane64(s.fft_stride, l);
#define ZA_S(a) fp[(int64_t)(a) * fs]
#define ZA_SLICE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
      constexpr int P = ZA_FFT_LDS_POINTS;
      if (op == ZA_COOP_PERMUTE || op == ZA_COOP_IPERMUTE) {
        // through the scratch. Position i of a WDL-ordered buffer holds natural bin perm[i]:
        //   fft_permute  (WDL order -> natural): out[k] = in[iperm[k]];   fft_ipermute (natural -> WDL order): out[i] = in[perm[i]]
        for (int i = rank; i < 2 * nl; i += nact) ZA_S(i) = ZA_G(i);
        ZA_SLICE_SYNC();
        for (int i = rank; i < nl; i += nact) {
          const int src = op == ZA_COOP_PERMUTE ? (int)za_fft_iperm[nl + i] : (int)za_fft_perm[nl + i];
          ZA_G(2 * i) = ZA_S(2 * src);
          ZA_G(2 * i + 1) = ZA_S(2 * src + 1);
        }
        ZA_SLICE_SYNC();
        continue;
      }
      // Two stages of (up to) six radix-2 passes each, a 1024-point chunk at a time through LDS; the buffer makes ONE trip through
      // the scratch between them (four 64 KB trips per 4096-point transform instead of round 3's six, no separate sorting pass).
      // With p the position in the bit-reversed array (p = bitrev(i)), passes 1 .. a only combine positions inside aligned
      // sub-blocks of A = 2^a positions, passes a + 1 .. a + 6 only positions that agree in their low a bits:
      //   stage A (a = bits - 6): sub-block g = p >> a holds the elements i = q * 64 + c, q < A, of residue c = bitrev6(g) mod 64; a
      //     chunk takes G = 1024 / A consecutive residues (256- / 512-byte runs of the buffer), LDS position m * G + (c mod G)
      //     with m = bitrev_a(q) the position inside the sub-block; every sub-block runs the first a passes of its own
      //     A-point transform (twiddles of the compact table);
      //   stage B: group u = p mod A holds p = u + A v, v < 64; a chunk takes 16 consecutive u -- 16 KB the scratch holds
      //     contiguously, sorted (u >> 4, v, u & 15) by stage A's stores -- at LDS position v * 16 + (u & 15) and runs the
      //     last six passes (twiddle index u + A (v mod half): the 4096-point table za_fft_tw4); position p is natural bin p.
      // Same butterflies, twiddles and order of operations per element as the in-LDS form, hence the same bits. LDS positions
      // are swizzled by ZA_B2 (low four bits XOR bits 4-7): the chunk is read by rows in the passes and by columns when it is
      // sorted into the scratch, both one bank group per lane of a 16-lane phase.
      static_assert(ZA_FFT_LDS_POINTS != 1024 || ZA_FFT_COOP_MAX == 4096, "two-stage slicing: 1024-point chunks, 64 x 64 at most");
      const int sign = (op == ZA_COOP_FFT || op == ZA_COOP_FFT_NAT) ? -1 : +1;
      constexpr int ZA_GU = 16;                                // a full wavefront moves a chunk in one batch of loads
      const int a = bits - 6, A = 1 << a, gb = 10 - a, G = 1 << gb;
      for (int C = 0; C < (64 >> gb); ++C) {
        for (int e0 = rank; e0 < P; e0 += ZA_GU * nact) {
          double vr[ZA_GU], vi[ZA_GU];
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact < P ? e0 + u * nact : e0;
            const int i = ((e >> gb) << 6) + G * C + (e & (G - 1));
            const int src = op == ZA_COOP_IFFT ? (int)za_fft_iperm[nl + i] : i;      // ifft: the position that holds bin i
            vr[u] = ZA_G(2 * src); vi[u] = ZA_G(2 * src + 1);
          }
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact;
            if (e < P) {
              const int loc = ((int)za_bitrev((uint32_t)(e >> gb), a) << gb) + (e & (G - 1));
              buf[ZA_B2(loc)] = vr[u]; buf[ZA_B2(loc) + 1] = vi[u];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        auto twa = [&](int jl, int lhb, double& wr, double& wi) __attribute__((always_inline)) {
          const int ix = (jl >> gb) << (9 - (lhb - gb));
          wr = za_fft_twc[2 * ix]; wi = za_fft_twc[2 * ix + 1];
        };
        if (a == 6) za_fft_lds_pass_v<3>(buf, 16, sign, rank, nact, twa);
        else za_fft_lds_pass_v<2>(buf, 32, sign, rank, nact, twa);
        za_fft_lds_pass_v<3>(buf, 128, sign, rank, nact, twa);
        for (int e0 = rank; e0 < P; e0 += ZA_GU * nact) {
          double vr[ZA_GU], vi[ZA_GU];
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact < P ? e0 + u * nact : e0;
            const int m = ((e >> (4 + gb)) << 4) + (e & 15), loc = m * G + ((e >> 4) & (G - 1));
            vr[u] = buf[ZA_B2(loc)]; vi[u] = buf[ZA_B2(loc) + 1];
          }
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact;
            if (e < P) {
              const int m = ((e >> (4 + gb)) << 4) + (e & 15), c = G * C + ((e >> 4) & (G - 1));
              const int d = ((m >> 4) << 10) + ((int)za_bitrev((uint32_t)c, 6) << 4) + (m & 15);
              ZA_S(2 * d) = vr[u]; ZA_S(2 * d + 1) = vi[u];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ZA_SLICE_SYNC();
      for (int D = 0; D < (A >> 4); ++D) {
        for (int e0 = rank; e0 < P; e0 += ZA_GU * nact) {
          double vr[ZA_GU], vi[ZA_GU];
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact < P ? e0 + u * nact : e0;
            vr[u] = ZA_S(2 * (D * P + e)); vi[u] = ZA_S(2 * (D * P + e) + 1);
          }
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact;
            if (e < P) { buf[ZA_B2(e)] = vr[u]; buf[ZA_B2(e) + 1] = vi[u]; }
          }
        }
        __builtin_amdgcn_wave_barrier();
        auto twb = [&](int jl, int lhb, double& wr, double& wi) __attribute__((always_inline)) {
          const int ix = ((jl & 15) + 16 * D + ((jl >> 4) << a)) << (11 - (lhb - 4 + a));
          wr = za_fft_tw4[2 * ix]; wi = za_fft_tw4[2 * ix + 1];
        };
        za_fft_lds_pass_v<3>(buf, 16, sign, rank, nact, twb);
        za_fft_lds_pass_v<3>(buf, 128, sign, rank, nact, twb);
        for (int e0 = rank; e0 < P; e0 += ZA_GU * nact) {
          int dst[ZA_GU];
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact < P ? e0 + u * nact : e0;
            const int pbin = 16 * D + (e & 15) + ((e >> 4) << a);                    // natural index of the result
            dst[u] = op == ZA_COOP_FFT ? (int)za_fft_iperm[nl + pbin] : pbin;        // fft alone: stored in WDL_fft_permute order
          }
#pragma unroll
          for (int u = 0; u < ZA_GU; ++u) {
            const int e = e0 + u * nact;
            if (e < P) { ZA_G(2 * dst[u]) = buf[ZA_B2(e)]; ZA_G(2 * dst[u] + 1) = buf[ZA_B2(e) + 1]; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ZA_SLICE_SYNC();
#undef ZA_S
#undef ZA_SLICE_SYNC
      continue;
    }
#ifdef ZA_FFT_STAMPS
    // (-DZA_FFT_STAMPS: one wavefront of the launch prints the s_memtime ticks of its first in-LDS transform's phases -- staging
    //  loads, LDS passes, write-back -- tools/probes/fft_phase_clock.py; 100 MHz ticks)
    const uint64_t zs0 = __builtin_readcyclecounter();
#endif
    // ---- stage the request's buffer in LDS, in the order its transform wants ----------------------------------------
    // (eight HBM reads in flight per lane: one read per trip would cost a full memory latency per element)
    for (int i0 = rank; i0 < nl; i0 += 8 * nact) {
      double vr[8], vi[8];
      uint32_t pm[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // (a trip past the end reads the lane's first element again instead of branching around its loads: a load under a lane
        //  condition is a branch, and the compiler closes every such block with a full wait -- the eight reads went out in two
        //  or three waited groups)
        const int i = i0 + u * nact < nl ? i0 + u * nact : i0;
        vr[u] = ZA_G(2 * i);
        vi[u] = ZA_G(2 * i + 1);
        pm[u] = (op == ZA_COOP_IFFT || op == ZA_COOP_PERMUTE || op == ZA_COOP_IFFT_REAL) ? za_fft_perm[nl + i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nact;
        if (i < nl) {
          uint32_t dst;
          if (op == ZA_COOP_FFT || op == ZA_COOP_FFT_REAL || op == ZA_COOP_FFT_REAL_NAT || is_nat) dst = za_bitrev((uint32_t)i, bits);
          else if (op == ZA_COOP_IFFT) dst = za_bitrev(pm[u], bits);
          else if (op == ZA_COOP_PERMUTE || op == ZA_COOP_IFFT_REAL) dst = pm[u];      // (ifft_real: natural bin order first)
          else dst = (uint32_t)i;
          buf[ZA_B(dst)] = vr[u];
          buf[ZA_B(dst) + 1] = vi[u];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (op == ZA_COOP_IFFT_REAL || op == ZA_COOP_IFFT_REAL_NAT) {
      // packed spectrum (natural order in buf) -> Z of the half-size complex transform, in place: bins k and h - k only need
      // each other; then the in-place bit reversal the butterflies want
      const int h = nl, step = ZA_FFT_MAX / nreq;
      for (int k = rank; k <= (h >> 1); k += nact) {
        const int m = h - k;
        if (k == 0) {
          const double x0 = buf[0], xn = buf[1];
          buf[0] = x0 + xn; buf[1] = x0 - xn;
        } else {
          const double ar = buf[ZA_B(k)], ai = buf[ZA_B(k) + 1], cr = buf[ZA_B(m)], ci = buf[ZA_B(m) + 1];
          {
            const double br = cr, bi = -ci;                                  // conj X[h-k]
            const double er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
            const double c = za_fft_cos[k * step], sn = za_fft_sin[k * step];
            buf[ZA_B(k)] = er + (-sn * dr - c * di);
            buf[ZA_B(k) + 1] = ei + (-sn * di + c * dr);
          }
          if (m != k) {
            const double br = ar, bi = -ai;                                  // the same formula for bin h - k
            const double er = cr + br, ei = ci + bi, dr = cr - br, di = ci - bi;
            const double c = za_fft_cos[m * step], sn = za_fft_sin[m * step];
            buf[ZA_B(m)] = er + (-sn * dr - c * di);
            buf[ZA_B(m) + 1] = ei + (-sn * di + c * dr);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int i = rank; i < h; i += nact) {
        const int r = (int)za_bitrev((uint32_t)i, bits);
        if (i < r) {
          const double t0 = buf[ZA_B(i)], t1 = buf[ZA_B(i) + 1];
          buf[ZA_B(i)] = buf[ZA_B(r)]; buf[ZA_B(i) + 1] = buf[ZA_B(r) + 1];
          buf[ZA_B(r)] = t0; buf[ZA_B(r) + 1] = t1;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef ZA_FFT_STAMPS
    const uint64_t zs1 = __builtin_readcyclecounter();
#endif
    if (op == ZA_COOP_FFT || op == ZA_COOP_IFFT || is_real || is_nat) {
      const int sign = (op == ZA_COOP_FFT || op == ZA_COOP_FFT_REAL || op == ZA_COOP_FFT_NAT || op == ZA_COOP_FFT_REAL_NAT) ? -1 : +1;
      za_fft_lds_stages(buf, tw, nl, sign, rank, nact);
    }
#ifdef ZA_FFT_STAMPS
    const uint64_t zs2 = __builtin_readcyclecounter();
#endif
    if (op == ZA_COOP_FFT_REAL || op == ZA_COOP_FFT_REAL_NAT) {
      // Z (natural order in buf) -> the packed spectrum of the real input, position i holds bin perm_h(i), scaled by 2
      // (fused with fft_permute: bin i)
      const int h = nl, step = ZA_FFT_MAX / nreq;
      for (int i = rank; i < h; i += nact) {
        const int k = op == ZA_COOP_FFT_REAL ? (int)za_fft_perm[h + i] : i;
        double re, im;
        if (k == 0) {
          re = 2.0 * (buf[0] + buf[1]);
          im = 2.0 * (buf[0] - buf[1]);
        } else {
          const int m = h - k;
          const double zr = buf[ZA_B(k)], zi = buf[ZA_B(k) + 1], yr = buf[ZA_B(m)], yi = -buf[ZA_B(m) + 1];
          const double er = zr + yr, ei = zi + yi, dr = zr - yr, di = zi - yi;
          const double c = za_fft_cos[k * step], sn = za_fft_sin[k * step];
          re = er + (-sn * dr + c * di);
          im = ei + (-sn * di - c * dr);
        }
        ZA_G(2 * i) = re;
        ZA_G(2 * i + 1) = im;
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // ---- write back in the order the builtin defines -------------------------------------------------------------------
    for (int i0 = rank; i0 < nl; i0 += 8 * nact) {
      uint32_t src[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nact < nl ? i0 + u * nact : i0;
        src[u] = (uint32_t)i;
        if (op == ZA_COOP_FFT || op == ZA_COOP_IPERMUTE) src[u] = za_fft_perm[nl + i];
      }
      double wr_[8], wi_[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { wr_[u] = buf[ZA_B(src[u])]; wi_[u] = buf[ZA_B(src[u]) + 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nact;
        if (i < nl) {
          ZA_G(2 * i) = wr_[u];
          ZA_G(2 * i + 1) = wi_[u];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#ifdef ZA_FFT_STAMPS
    {
      __builtin_amdgcn_s_waitcnt(0);
      const uint64_t zs3 = __builtin_readcyclecounter();
      if (rank == 0 && blockIdx.x == 7 && op == ZA_COOP_FFT_NAT && atomicAdd(&za_fft_stamp_once, 1) < 4)
        printf("za_fft_coop n=%d nact=%d: stage %llu, passes %llu, write-back %llu cycles\n", nl, nact,
               (unsigned long long)(zs1 - zs0), (unsigned long long)(zs2 - zs1), (unsigned long long)(zs3 - zs2));
    }
#endif
#undef ZA_G
  }
  return coop;
}
#define ZA_FFT_TRY_COOP(op) if (za_fft_coop(s, ok, base, (int)n, op)) return 0.0
#else
#define ZA_FFT_TRY_COOP(op) (void)0
#endif

// natural-order in-scratch DIT butterflies on n complex values (already stored bit-reversed); sign = -1 forward, +1 inverse
template <class S>
ZA_FN void za_fft_butterflies(S& s, int n, int sign) {
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = ZA_FFT_MAX / len;
    for (int j = 0; j < half; ++j) {
      const double wr = za_fft_cos[j * step], wi = (sign < 0 ? -za_fft_sin[j * step] : za_fft_sin[j * step]);
      for (int i = j; i < n; i += len) {
        const int a = 2 * i, b = 2 * (i + half);
        const double br = ZA_F(b), bi = ZA_F(b + 1);
        const double tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        const double ar = ZA_F(a), ai = ZA_F(a + 1);
        ZA_F(a) = ar + tr; ZA_F(a + 1) = ai + ti;
        ZA_F(b) = ar - tr; ZA_F(b + 1) = ai - ti;
      }
    }
  }
}

// forward complex transform of mem[base .. base+2n): natural in, WDL order out
template <class S>
ZA_FN void za_fft_fwd_core(S& s, int64_t base, int n) {
  const int bits = za_log2((uint32_t)n);
  for (int i = 0; i < n; ++i) {
    const uint32_t r = za_bitrev((uint32_t)i, bits);
    ZA_F(2 * r) = ZA_M(base + 2 * i);
    ZA_F(2 * r + 1) = ZA_M(base + 2 * i + 1);
  }
  za_fft_butterflies(s, n, -1);
  for (int i = 0; i < n; ++i) {
    const uint32_t k = za_fft_bin_of_pos((uint32_t)i, (uint32_t)n);
    ZA_M(base + 2 * i) = ZA_F(2 * k);
    ZA_M(base + 2 * i + 1) = ZA_F(2 * k + 1);
  }
}
// inverse complex transform: WDL order in, natural out, unscaled
template <class S>
ZA_FN void za_fft_inv_core(S& s, int64_t base, int n) {
  const int bits = za_log2((uint32_t)n);
  for (int i = 0; i < n; ++i) {
    const uint32_t k = za_fft_bin_of_pos((uint32_t)i, (uint32_t)n);
    const uint32_t r = za_bitrev(k, bits);
    ZA_F(2 * r) = ZA_M(base + 2 * i);
    ZA_F(2 * r + 1) = ZA_M(base + 2 * i + 1);
  }
  za_fft_butterflies(s, n, +1);
  for (int i = 0; i < 2 * n; ++i) ZA_M(base + i) = ZA_F(i);
}

template <class S> ZA_NOINLINE double za_fft_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_FFT);
  if (!ok) return 0.0;
  za_fft_fwd_core(s, base, (int)n);
  return 0.0;
}
template <class S> ZA_NOINLINE double za_ifft_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_IFFT);
  if (!ok) return 0.0;
  za_fft_inv_core(s, base, (int)n);
  return 0.0;
}
template <class S> ZA_NOINLINE double za_fft_permute_o(S& s, double baseD, double sizeD);
template <class S> ZA_NOINLINE double za_fft_ipermute_o(S& s, double baseD, double sizeD);
// fft(b, n); fft_permute(b, n) and fft_ipermute(b, n); ifft(b, n) with identical arguments, fused by the translator (zajit/emit.py
// e_Seq): the permutations are exact moves, so a natural-order transform gives the same bits with one pass over the buffer
// fewer. The serial form (CPU port; device beyond the cooperative sizes) simply runs the two builtins.
template <class S> ZA_NOINLINE double za_fft_nat_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_FFT_NAT);
  if (!ok) return 0.0;
  za_fft_fwd_core(s, base, (int)n);
  return za_fft_permute_o(s, baseD, sizeD);
}
template <class S> ZA_NOINLINE double za_ifft_nat_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_IFFT_NAT);
  if (!ok) return 0.0;
  za_fft_ipermute_o(s, baseD, sizeD);
  za_fft_inv_core(s, base, (int)n);
  return 0.0;
}
// fft_real(b, n); fft_permute(b, m) and fft_ipermute(b, m); ifft_real(b, n) as adjacent calls with the same plain arguments
// (zajit/emit.py _fuse_fft_pairs; how PsychoConvolver calls them, m = n / 2): where m is half of n and the pair fits the
// wave-cooperative real transform, ONE operation that leaves / takes the packed spectrum in natural bin order -- the permutation
// is an exact move, so the bits are those of the two calls; anything else (other sizes, the CPU port) runs the two builtins.
template <class S> ZA_NOINLINE double za_fft_real_o(S& s, double baseD, double sizeD);
template <class S> ZA_NOINLINE double za_ifft_real_o(S& s, double baseD, double sizeD);
template <class S> ZA_NOINLINE double za_fft_real_nat_o(S& s, double baseD, double sizeD, double halfD) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  const int64_t n = za_round_idx(sizeD), m = za_round_idx(halfD);
  if (2 * m == n && m >= ZA_FFT_MIN && n <= 2 * ZA_FFT_LDS_POINTS) {       // both builtins would accept their size
    int64_t base = 0;
    const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, n, base, 2 * n);
    if (za_fft_coop(s, ok, base, (int)n, ZA_COOP_FFT_REAL_NAT)) return 0.0;
    return 0.0;                                                              // (!ok: both calls would have done nothing)
  }
#endif
  za_fft_real_o(s, baseD, sizeD);
  return za_fft_permute_o(s, baseD, halfD);
}
template <class S> ZA_NOINLINE double za_ifft_real_nat_o(S& s, double baseD, double sizeD, double halfD) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  const int64_t n = za_round_idx(sizeD), m = za_round_idx(halfD);
  if (2 * m == n && m >= ZA_FFT_MIN && n <= 2 * ZA_FFT_LDS_POINTS) {
    int64_t base = 0;
    const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, n, base, 2 * n);
    if (za_fft_coop(s, ok, base, (int)n, ZA_COOP_IFFT_REAL_NAT)) return 0.0;
    return 0.0;
  }
#endif
  za_fft_ipermute_o(s, baseD, halfD);
  return za_ifft_real_o(s, baseD, sizeD);
}
template <class S> ZA_NOINLINE double za_fft_permute_o(S& s, double baseD, double sizeD) {   // WDL order -> natural
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_PERMUTE);
  if (!ok) return 0.0;
  for (int i = 0; i < (int)n; ++i) {
    const uint32_t k = za_fft_bin_of_pos((uint32_t)i, (uint32_t)n);
    ZA_F(2 * k) = ZA_M(base + 2 * i);
    ZA_F(2 * k + 1) = ZA_M(base + 2 * i + 1);
  }
  for (int i = 0; i < 2 * (int)n; ++i) ZA_M(base + i) = ZA_F(i);
  return 0.0;
}
template <class S> ZA_NOINLINE double za_fft_ipermute_o(S& s, double baseD, double sizeD) {  // natural -> WDL order
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, 2 * n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_IPERMUTE);
  if (!ok) return 0.0;
  for (int i = 0; i < 2 * (int)n; ++i) ZA_F(i) = ZA_M(base + i);
  for (int i = 0; i < (int)n; ++i) {
    const uint32_t k = za_fft_bin_of_pos((uint32_t)i, (uint32_t)n);
    ZA_M(base + 2 * i) = ZA_F(2 * k);
    ZA_M(base + 2 * i + 1) = ZA_F(2 * k + 1);
  }
  return 0.0;
}

// Real transforms ("two for one"): N reals are treated as N/2 complex z[t] = x[2t] + i x[2t+1].
//   forward:  Z = FFT_{N/2}(z) (natural bins), X[k] = 0.5*(Z[k] + conj Z[h-k]) - 0.5i e^{-2 pi i k/N} (Z[k] - conj Z[h-k]),
//             stored 2*X[k] at position perm_h(k); position 0 holds (2*X[0], 2*X[N/2]).
//   inverse:  the exact reverse, then an unscaled inverse FFT_{N/2}.
template <class S> ZA_NOINLINE double za_fft_real_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_FFT_REAL);
  if (!ok) return 0.0;
  const int h = (int)(n >> 1), bits = za_log2((uint32_t)h);
  for (int i = 0; i < h; ++i) {
    const uint32_t r = za_bitrev((uint32_t)i, bits);
    ZA_F(2 * r) = ZA_M(base + 2 * i);
    ZA_F(2 * r + 1) = ZA_M(base + 2 * i + 1);
  }
  za_fft_butterflies(s, h, -1);                                   // scratch = Z[0..h) natural
  const int step = ZA_FFT_MAX / (int)n;
  for (int i = 0; i < h; ++i) {
    const int k = (int)za_fft_bin_of_pos((uint32_t)i, (uint32_t)h);
    double re, im;
    if (k == 0) {
      re = 2.0 * (ZA_F(0) + ZA_F(1));                              // 2*X[0]
      im = 2.0 * (ZA_F(0) - ZA_F(1));                              // 2*X[N/2]
    } else {
      const int m = h - k;
      const double zr = ZA_F(2 * k), zi = ZA_F(2 * k + 1), yr = ZA_F(2 * m), yi = -ZA_F(2 * m + 1);   // y = conj Z[h-k]
      const double er = zr + yr, ei = zi + yi, dr = zr - yr, di = zi - yi;
      const double c = za_fft_cos[k * step], sn = za_fft_sin[k * step];   // e^{-i t}: (c, -sn); -i*e^{-it} = (-sn, -c)
      // 2X = (er,ei) + (-sn,-c)*(dr,di)
      re = er + (-sn * dr + c * di);
      im = ei + (-sn * di - c * dr);
    }
    ZA_M(base + 2 * i) = re;
    ZA_M(base + 2 * i + 1) = im;
  }
  return 0.0;
}
template <class S> ZA_NOINLINE double za_ifft_real_o(S& s, double baseD, double sizeD) {
  const int64_t n = za_round_idx(sizeD);
  int64_t base = 0;
  const bool ok = za_fft_pow2(n) && za_fft_region(s, baseD, n, base, 2 * n);
  ZA_FFT_TRY_COOP(ZA_COOP_IFFT_REAL);
  if (!ok) return 0.0;
  const int h = (int)(n >> 1), bits = za_log2((uint32_t)h);
  // gather the packed spectrum into natural order at the start of scratch's upper half, then build Z bit-reversed
  // in the lower half. scratch capacity >= 2*(2h) doubles is guaranteed by fft_cap >= n (complex count) * 2.
  const int64_t up = 2 * (int64_t)h;
  for (int i = 0; i < h; ++i) {
    const uint32_t k = za_fft_bin_of_pos((uint32_t)i, (uint32_t)h);
    ZA_F(up + 2 * k) = ZA_M(base + 2 * i);
    ZA_F(up + 2 * k + 1) = ZA_M(base + 2 * i + 1);
  }
  const int step = ZA_FFT_MAX / (int)n;
  for (int k = 0; k < h; ++k) {
    double zr, zi;
    if (k == 0) {
      const double x0 = ZA_F(up), xn = ZA_F(up + 1);               // X[0], X[N/2] (both real)
      zr = x0 + xn; zi = x0 - xn;                                  // Z[0] = (X0 + Xn) + i (X0 - Xn)
    } else {
      const int m = h - k;
      const double ar = ZA_F(up + 2 * k), ai = ZA_F(up + 2 * k + 1);          // X[k]
      const double br = ZA_F(up + 2 * m), bi = -ZA_F(up + 2 * m + 1);         // conj X[h-k]
      const double er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
      const double c = za_fft_cos[k * step], sn = za_fft_sin[k * step];       // i*e^{+it} = (-sn, c)
      zr = er + (-sn * dr - c * di);
      zi = ei + (-sn * di + c * dr);
    }
    const uint32_t r = za_bitrev((uint32_t)k, bits);
    ZA_F(2 * r) = zr;
    ZA_F(2 * r + 1) = zi;
  }
  za_fft_butterflies(s, h, +1);
  for (int i = 0; i < (int)n; ++i) ZA_M(base + i) = ZA_F(i);
  return 0.0;
}

// convolve_c(dest, src, size): dest[i] *= src[i] for `size` complex pairs, src read before dest is written on overlap.
template <class S> ZA_NOINLINE double za_convolve_c_o(S& s, double destD, double srcD, double sizeD) {
  const int64_t cnt = za_round_idx(sizeD);
  int64_t d = 0, r = 0;
  const bool okc = cnt > 0 && cnt <= ZA_FFT_PAGE / 2 && za_fft_region(s, destD, 2 * cnt, d, 2 * cnt) && za_fft_region(s, srcD, 2 * cnt, r, 2 * cnt);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  if (za_fft_coop(s, okc, d, (int)cnt, ZA_COOP_CONVOLVE, r)) return 0.0;
#endif
  if (!okc) return 0.0;
  const bool overlap = (d < r + 2 * cnt) && (r < d + 2 * cnt) && d != r;
  if (overlap) for (int64_t i = 0; i < 2 * cnt; ++i) ZA_F(i) = ZA_M(r + i);
  for (int64_t i = 0; i < cnt; ++i) {
    const double ar = ZA_M(d + 2 * i), ai = ZA_M(d + 2 * i + 1);
    const double br = overlap ? ZA_F(2 * i) : ZA_M(r + 2 * i), bi = overlap ? ZA_F(2 * i + 1) : ZA_M(r + 2 * i + 1);
    ZA_M(d + 2 * i) = ar * br - ai * bi;
    ZA_M(d + 2 * i + 1) = ar * bi + ai * br;
  }
  return 0.0;
}

// Inline stubs (zart.h ZA_OUTCALL): the out-of-line bodies above never see the state object -- and not the whole environment
// either, only the ten words the transforms use (a call costs their stores, the three that can change come back).
struct ZaFftEnv {
  using Env = ZaFftEnv;
  double* mem; int64_t mem_stride, mem_cap, mem_high, mem_need;
  double* fft; int64_t fft_stride, fft_cap;
  uint32_t err, replica;
};
#define ZA_FFT_OUT(call) \
  ZaFftEnv e; e.mem = s.mem; e.mem_stride = s.mem_stride; e.mem_cap = s.mem_cap; e.mem_high = s.mem_high; e.mem_need = s.mem_need; \
  e.fft = s.fft; e.fft_stride = s.fft_stride; e.fft_cap = s.fft_cap; e.err = s.err; e.replica = s.replica; \
  const double r_ = (call); s.mem_high = e.mem_high; s.mem_need = e.mem_need; s.err = e.err; return r_
#define ZA_X(name) template <class S> ZA_FN double name(S& s, double baseD, double sizeD) { ZA_FFT_OUT(name##_o(e, baseD, sizeD)); }
ZA_X(za_fft) ZA_X(za_ifft) ZA_X(za_fft_permute) ZA_X(za_fft_ipermute) ZA_X(za_fft_nat) ZA_X(za_ifft_nat) ZA_X(za_fft_real) ZA_X(za_ifft_real)
#undef ZA_X
template <class S> ZA_FN double za_convolve_c(S& s, double destD, double srcD, double sizeD) { ZA_FFT_OUT(za_convolve_c_o(e, destD, srcD, sizeD)); }
template <class S> ZA_FN double za_fft_real_nat(S& s, double baseD, double sizeD, double halfD) { ZA_FFT_OUT(za_fft_real_nat_o(e, baseD, sizeD, halfD)); }
template <class S> ZA_FN double za_ifft_real_nat(S& s, double baseD, double sizeD, double halfD) { ZA_FFT_OUT(za_ifft_real_nat_o(e, baseD, sizeD, halfD)); }
#undef ZA_FFT_OUT

#undef ZA_M
#undef ZA_F
