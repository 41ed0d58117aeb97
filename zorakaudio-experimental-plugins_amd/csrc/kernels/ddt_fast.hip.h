// ddt_fast.hip.h -- zab_ddt_wide[_nw2|_nw4|_nw8]: hand-written gfx950 kernel for the @sample loop of Spatialization/DDT with ONE raw
// history ring in LDS (taps first, then the six one-poles). Round 1's headline kernel; since round 2 the fallback of
// ddt_ring2.hip.h (zab_ddt_fast: one-poles first, two filtered rings) for tap delays too long for two rings -- up to the
// script's 16 382 frames fit here -- and the home of the helpers both share.
//
// What it computes is exactly DDT's @sample section (reference: plugins/Spatialization/DDT/src/DDT.jsfx:440-536,
// run per frame by jsfx_process_block, dsp_jsfx_aot.py:5713-5905); HOW it computes it is MI355X-first:
//
//   * DDT's per-frame work is feed-forward in the INPUT history (sparse taps into the bL/bR delay rings) followed
//     by LINEAR constant-coefficient one-poles (dirZ*, eZ*, lZ*, and the seven UI meters). So time is data-parallel.
//     One WORKGROUP PER INSTANCE made of NW wavefronts (NW = 1, 2, 4 or 8, chosen so that the batch fills the chip);
//     each wavefront takes one 256-frame chunk per iteration, each lane 4 consecutive frames of it.
//   * The mono delay history M[n] = 0.5*(L[n]+R[n]) (the only thing the taps read: :467-468) lives in one LDS ring
//     shared by the workgroup. Two addressing modes (ddt_geometry): a DOUBLED ring of Dmax + NW*256 + 1 frames (rounded to
//     256) stored twice back to back, so `frame - delay` never wraps in the tap phase; or, when twice the history does
//     not fit beside 8 resident waves per CU, a power-of-two ring with masked byte offsets. Either has a copy of its
//     first 256 slots behind its end for the strided reads.
//   * Tap phase: lane l takes frames {l, 64+l, 128+l, 192+l} of its chunk, so the gather `frame - delay` of a wave reads
//     64 consecutive ring slots per k (conflict-free for every delay) and the four k differ by a constant 512 bytes:
//     one address per (tap, channel), the rest are ds_read immediates. Tap parameters live in registers -- lane j keeps
//     staged tap j (early taps first, then late taps, each in source order) -- and are broadcast with v_readlane; the
//     eight ring reads of a tap are issued one tap ahead of their FMAs. The four sums then move to the "4 consecutive
//     frames per lane" layout through a per-wave LDS transpose.
//   * The HBM read of a wave's next chunk is issued a whole iteration ahead of its use.
//   * The six filter recurrences y[n] = (1-a) x[n] + a y[n-1]: 4 serial steps inside the lane, a weighted 64-lane scan
//     of the lane aggregates with coefficient a^4 done with DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31 -- no LDS
//     crossbar traffic, no lane masks), a chunk-to-chunk carry chain across the NW waves through 6 doubles of LDS per
//     wave, then a 4-step fix-up. Wave 0 injects the running state at its first frame, so lane 0 of its chunk
//     reproduces the serial rounding exactly; everything else differs by O(1e-16) relative (FMA, re-association).
//   * The seven meter one-poles are only observable as state after the launch, so they are carried as per-lane
//     weighted partial sums and reduced once, at the end.
//   * HBM traffic per frame: 8 B in + 8 B out (float4 per lane per channel, 1 KiB per wave instruction); the f64
//     rings in mem[] are written only for the last 16384 frames of a launch (older slots would be overwritten).
//     vars[] / tap tables are touched once per launch. Measured with FETCH_SIZE/WRITE_SIZE: 1.0004x these bytes.
//
// State contract: on exit vars[] and mem[] hold what the serial path would hold (all @sample temporaries of the last
// frame included), within the scan's rounding for the filter states -- tests/test_ddt_gpu.py compares both paths.
#pragma once

#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>

static char ddt_kernel_name[24] = "zab_ddt_fast";   /* zab_ddt_{fast,wide}[_nw2|_nw4|_nw8] once a launch has picked kernel and waves per instance */
#define ZA_FAST_KERNEL_NAME ddt_kernel_name
#define DDT_KF 4                       /* frames per lane */
#define DDT_CHUNK (64 * DDT_KF)        /* frames per wave iteration */
#define DDT_MAXTAPS 64
#define DDT_RING 16384                 /* BUF_LEN of the script */
#define DDT_MAXNW 8
// The meter one-poles (a = 0.9985, 0.9990) are only observable as state after the launch, and a frame k frames before the
// end of the launch enters them with weight a^k: beyond 57 344 frames that is below 0.9990^57344 = 2^-82 of the frame's own
// magnitude -- under the rounding of the sums it would be added to -- so only the launch's last 224 chunks feed the meters.
#define DDT_METER_FRAMES 57344

typedef float ddt_f4 __attribute__((ext_vector_type(4)));   // a float4 the register allocator keeps as ONE 128-bit tuple
struct DdtTap { int32_t dL8, dR8; double gL, gR; };          // 8*delay (bytes) and gains of one tap (wave-uniform when used)
struct DdtTapRegs { int dpack; double gL, gR; };             // lane j of every wave keeps staged tap j: dL | dR << 16, gains
struct DdtPole {
  double a, c1;        // pole and (1 - pole)
  double ap[4];        // a^1..a^4
  double sp[4];        // (a^4)^(2^j), j = 0..3 : in-row scan step coefficients
  double a256;         // a^256: decay of a state across one chunk
};

// Optional in-kernel phase clock (-DDDT_STAMPS, tools/ddt_stamps.py): per-wave cycles between the stamps of an iteration,
// summed over the launch into ddt_stamp_acc. A stamp waits for its own s_memtime only; it still perturbs the LDS pipelining a
// little, so the build is for attribution, not for timing.
#ifdef DDT_STAMPS
__device__ unsigned long long ddt_stamp_acc[16];
#define DDT_STAMP_DECL unsigned long long ddt_st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ddt_tl = __builtin_amdgcn_s_memtime();
#define DDT_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ddt_st[i] += now_ - ddt_tl; ddt_tl = now_; }
#define DDT_STAMP_ARGS , unsigned long long (&ddt_st)[10], unsigned long long& ddt_tl
#define DDT_STAMP_PASS , ddt_st, ddt_tl
#else
#define DDT_STAMP_DECL
#define DDT_STAMP(i)
#define DDT_STAMP_ARGS
#define DDT_STAMP_PASS
#endif

// ---- wave-level helpers ------------------------------------------------------------------------------------------
__device__ __forceinline__ double ddt_readlane(double v, int l) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_readlane(t.x, l);
  t.y = __builtin_amdgcn_readlane(t.y, l);
  return __builtin_bit_cast(double, t);
}
// DPP move of a double; lanes without a source (or masked out by ROWS) receive 0. With every row enabled the old value of
// the destination is never kept (bound_ctrl zero-fills), so the plain mov form is used and no zero has to be materialised.
template <int CTRL, int ROWS>
__device__ __forceinline__ double ddt_dpp(double v) {
  int2 t = __builtin_bit_cast(int2, v);
  if (ROWS == 0xF) {
    t.x = __builtin_amdgcn_mov_dpp(t.x, CTRL, 0xF, 0xF, true);
    t.y = __builtin_amdgcn_mov_dpp(t.y, CTRL, 0xF, 0xF, true);
  } else {
    t.x = __builtin_amdgcn_update_dpp(0, t.x, CTRL, ROWS, 0xF, true);
    t.y = __builtin_amdgcn_update_dpp(0, t.y, CTRL, ROWS, 0xF, true);
  }
  return __builtin_bit_cast(double, t);
}
#define DDT_ROW_SHR(n) (0x110 | (n))
#define DDT_WAVE_SHR1 0x138
#define DDT_ROW_BCAST15 0x142
#define DDT_ROW_BCAST31 0x143

// Workgroup barrier of the chunk loop: LDS traffic must have landed (lgkmcnt), but the HBM prefetch of the next chunk and
// the output stores must stay in flight -- __syncthreads() would drain them too (s_waitcnt vmcnt(0)) and put a full memory
// latency back into every iteration.
__device__ __forceinline__ void ddt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double ddt_clamp(double x, double a, double b) { return x < a ? a : (x > b ? b : x); }
__device__ __forceinline__ double ddt_ipow(double base, int64_t e) {   // base^e, e >= 0, square-and-multiply
  double r = 1.0;
  while (e) { if (e & 1) r *= base; base *= base; e >>= 1; }
  return r;
}

// Inclusive weighted scan across the wave: g[l] <- sum_{i<=l} q^(l-i) g[i], q = a^4.
// cb1 = q^((l&15)+1), cb2 = q^(l-31) are per-lane constants of the pole.
__device__ __forceinline__ double ddt_scan(double g, const DdtPole& p, double cb1, double cb2) {
  g = __builtin_fma(p.sp[0], ddt_dpp<DDT_ROW_SHR(1), 0xF>(g), g);
  g = __builtin_fma(p.sp[1], ddt_dpp<DDT_ROW_SHR(2), 0xF>(g), g);
  g = __builtin_fma(p.sp[2], ddt_dpp<DDT_ROW_SHR(4), 0xF>(g), g);
  g = __builtin_fma(p.sp[3], ddt_dpp<DDT_ROW_SHR(8), 0xF>(g), g);
  g = __builtin_fma(cb1, ddt_dpp<DDT_ROW_BCAST15, 0xA>(g), g);
  g = __builtin_fma(cb2, ddt_dpp<DDT_ROW_BCAST31, 0xC>(g), g);
  return g;
}

// In-lane part of one recurrence: x[k] <- response of the lane's 4 frames to their own inputs (plus `inject`, the state
// before the chunk, at the chunk's first valid frame when this wave owns the head of the chain). Returns the inclusive
// scan G of the lane aggregates (y at the end of each lane, given zero state before the chunk apart from `inject`).
template <bool PARTIAL>
__device__ __forceinline__ double ddt_pole_local(const DdtPole& p, double cb1, double cb2, double (&x)[DDT_KF], double inject,
                                                 bool head, int lane, int first_lane, int first_k) {
  double z = 0.0;
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    double prev = z;
    if (PARTIAL) {
      if (head && lane == first_lane && k == first_k) prev = inject;
    } else {
      if (head && k == 0 && lane == 0) prev = inject;
    }
    z = __builtin_fma(p.a, prev, p.c1 * x[k]);           // (1-a)*x + a*prev
    if (PARTIAL) {
      const bool valid = (lane > first_lane) || (lane == first_lane && k >= first_k);
      z = valid ? z : 0.0;
    }
    x[k] = z;
  }
  return ddt_scan(z, p, cb1, cb2);
}

struct DdtCtx {
  // workgroup-uniform launch constants
  const float *in0, *in1;
  float *out0, *out1;
  double* Mem;
  double* ring;
  const DdtPole* P;
  double* T;           // this wave's [4][DDT_CHUNK] transpose area
  int64_t frames, wofs0, rL, rR;
  double* V;           // vars[] of the instance: the last frame's @sample temporaries are stored as they are produced
  double* SPL;
  int bufmask, W, m8, nE, nT, mon;
  double col, one_m_col, directGain, wetp, dryp, out_gain;
  double dgc, dgm, mixd, mixw;   // folded products: directGain*(1-col), directGain*col*0.5, dryp*out_gain, wetp*out_gain
  bool vec_ok;
};

struct DdtChunk {  // per-lane registers that live across the two workgroup barriers of an iteration
  float x0[DDT_KF], x1[DDT_KF];   // the chunk's input frames as read (f32 -> f64 is exact, so convert at each use)
  double y[6][DDT_KF];
  double G[6];
};

// Phase A: read the chunk's audio, publish M to the shared ring (and the f64 L/R rings of mem[] when they survive).
template <bool PARTIAL, bool DBL>
__device__ __forceinline__ void ddt_phase_a(const DdtCtx& C, DdtChunk& K, int lane, int64_t f0, int nb, const ddt_f4& p0, const ddt_f4& p1) {
  const int64_t t0 = f0 + DDT_KF * lane;
  if (!PARTIAL && C.vec_ok) {                              // prefetched one iteration ahead (ddt_prefetch)
    K.x0[0] = p0.x; K.x0[1] = p0.y; K.x0[2] = p0.z; K.x0[3] = p0.w;
    K.x1[0] = p1.x; K.x1[1] = p1.y; K.x1[2] = p1.z; K.x1[3] = p1.w;
  } else {
    // (these loads are waited for inside their branch: a wait at the merge with the prefetched path would be a vmcnt(0)
    //  for that path too, i.e. for the previous chunk's stores)
    if (C.vec_ok && t0 >= 0) {
      const float4 v0 = *reinterpret_cast<const float4*>(C.in0 + t0);
      const float4 v1 = *reinterpret_cast<const float4*>(C.in1 + t0);
      K.x0[0] = v0.x; K.x0[1] = v0.y; K.x0[2] = v0.z; K.x0[3] = v0.w;
      K.x1[0] = v1.x; K.x1[1] = v1.y; K.x1[2] = v1.z; K.x1[3] = v1.w;
    } else {
#pragma unroll
      for (int k = 0; k < DDT_KF; ++k) {
        const bool ok = t0 + k >= 0;
        K.x0[k] = ok ? C.in0[t0 + k] : 0.0f;
        K.x1[k] = ok ? C.in1[t0 + k] : 0.0f;
      }
    }
    asm volatile("" : "+v"(K.x0[0]), "+v"(K.x0[1]), "+v"(K.x0[2]), "+v"(K.x0[3]), "+v"(K.x1[0]), "+v"(K.x1[1]), "+v"(K.x1[2]), "+v"(K.x1[3]));
  }
  // nb = ring position of the chunk's first frame (uniform)
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    const double M = 0.5 * ((double)K.x0[k] + (double)K.x1[k]);   // mono (:445) == ring value 0.5*(L+R) (:467)
    if (!PARTIAL || t0 + k >= 0) {
      if (DBL) {                                           // doubled ring: [0,W) | [W,2W) | copy of the first 256 slots
        int slot = nb + DDT_KF * lane + k;
        slot = slot >= C.W ? slot - C.W : slot;
        C.ring[slot] = M;
        C.ring[C.W + slot] = M;
        if (slot < DDT_CHUNK) C.ring[2 * C.W + slot] = M;
      } else {                                             // power-of-two ring + copy of the first 256 slots
        const int slot = (nb + DDT_KF * lane + k) & (C.W - 1);
        C.ring[slot] = M;
        if (slot < DDT_CHUNK) C.ring[C.W + slot] = M;
      }
    }
  }
  if (f0 + DDT_CHUNK > C.frames - DDT_RING) {             // :441-442, only slots that survive the launch
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) {
      const int64_t t = t0 + k;
      if (t >= 0 && t >= C.frames - DDT_RING) {
        const int64_t ri = (C.wofs0 + t) & C.bufmask;
        C.Mem[C.rL + ri] = (double)K.x0[k];
        C.Mem[C.rR + ri] = (double)K.x1[k];
      }
    }
  }
}

// Tap sums over taps [i0, i1) of the staged list. Tap j's parameters live in lane j's registers and are broadcast with
// v_readlane (no memory latency); the eight ring reads of a tap are issued one tap ahead of their FMAs, so the LDS latency is
// paid once per run instead of once per tap.
__device__ __forceinline__ DdtTap ddt_tap_get(const DdtTapRegs& R, int i) {
  DdtTap t;
  const int pk = __builtin_amdgcn_readlane(R.dpack, i);
  t.dL8 = (pk & 0xffff) << 3;
  t.dR8 = (int)((unsigned)pk >> 16) << 3;
  t.gL = ddt_readlane(R.gL, i);
  t.gR = ddt_readlane(R.gR, i);
  return t;
}
// DBL: `ringb` already points at frame (nb + lane) of the SECOND copy of a doubled ring, so `- delay` never leaves the
// allocation and needs no wrap; otherwise the byte offset is wrapped with the power-of-two mask.
template <bool DBL>
__device__ __forceinline__ void ddt_tap_fetch(const char* ringb, int lane8nb, int m8, const DdtTap& tp, double (&l)[DDT_KF],
                                              double (&r)[DDT_KF]) {
  const char* pl = DBL ? ringb - tp.dL8 : ringb + ((lane8nb - tp.dL8) & m8);
  const char* pr = DBL ? ringb - tp.dR8 : ringb + ((lane8nb - tp.dR8) & m8);
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    l[k] = *reinterpret_cast<const double*>(pl + 512 * k);
    r[k] = *reinterpret_cast<const double*>(pr + 512 * k);
  }
}
__device__ __forceinline__ void ddt_tap_acc(const DdtTap& tp, const double (&l)[DDT_KF], const double (&r)[DDT_KF],
                                            double (&acc)[2][DDT_KF]) {
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    acc[0][k] = __builtin_fma(tp.gL, l[k], acc[0][k]);
    acc[1][k] = __builtin_fma(tp.gR, r[k], acc[1][k]);
  }
}
template <bool DBL>
__device__ __forceinline__ void ddt_tap_run(const DdtCtx& C, const DdtTapRegs& R, const char* ringb, int lane8nb, int i0, int i1,
                                            double (&acc)[2][DDT_KF]) {
  if (i0 >= i1) return;
  const int last = i1 - 1;
  double al[DDT_KF], ar[DDT_KF], bl[DDT_KF], br[DDT_KF];
  DdtTap ta = ddt_tap_get(R, i0), tb;
  ddt_tap_fetch<DBL>(ringb, lane8nb, C.m8, ta, al, ar);
  int i = i0;
  for (; i + 1 < i1; i += 2) {                             // pairs; the fetch past the end re-reads the last tap
    tb = ddt_tap_get(R, i + 1);
    ddt_tap_fetch<DBL>(ringb, lane8nb, C.m8, tb, bl, br);
    ddt_tap_acc(ta, al, ar, acc);
    ta = ddt_tap_get(R, i + 2 < last ? i + 2 : last);
    ddt_tap_fetch<DBL>(ringb, lane8nb, C.m8, ta, al, ar);
    ddt_tap_acc(tb, bl, br, acc);
  }
  if (i < i1) ddt_tap_acc(ta, al, ar, acc);                // odd count: the last tap is already in (ta, al, ar)
}

// Phase B: taps (:459-484), transpose, in-lane recurrences + scans. Leaves y (zero-state responses) and G in K.
template <bool PARTIAL, bool DBL>
__device__ __forceinline__ void ddt_phase_b(const DdtCtx& C, const DdtTapRegs& R, DdtChunk& K, int lane, int64_t f0, int nb, const double (&carry)[6], bool head,
                                            const double (&cb1)[3], const double (&cb2)[3], bool want_last DDT_STAMP_ARGS) {
  int first_lane = 0, first_k = 0;
  if (PARTIAL) {
    const int firstv = (int)(-f0);
    first_lane = firstv / DDT_KF;
    first_k = firstv % DDT_KF;
  }
  {
    // DBL: pointer to frame (nb + lane) in the second copy; else ring base + wrapped byte offset of frame (nb + lane)
    const char* ringb = reinterpret_cast<const char*>(DBL ? C.ring + C.W + nb + lane : C.ring);
    const int lane8nb = DBL ? 0 : ((8 * lane + 8 * nb) & C.m8);
    double sE[2][DDT_KF], sL[2][DDT_KF];
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) sE[0][k] = sE[1][k] = sL[0][k] = sL[1][k] = 0.0;
    // strided -> blocked through the wave's own [2][256] LDS area: lane l wrote frames 64k+l, reads back frames 4l..4l+3
    // (a wave's LDS accesses execute in order, so only the compiler needs the fence)
#define DDT_TRANSPOSE(acc, dst)                                                                        \
    {                                                                                                  \
      _Pragma("unroll") for (int k = 0; k < DDT_KF; ++k) {                                             \
        C.T[64 * k + lane] = acc[0][k];                                                                \
        C.T[DDT_CHUNK + 64 * k + lane] = acc[1][k];                                                    \
      }                                                                                                \
      __builtin_amdgcn_wave_barrier();                                                                 \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                  \
        const double2 lo = *reinterpret_cast<const double2*>(C.T + c * DDT_CHUNK + DDT_KF * lane);     \
        const double2 hi = *reinterpret_cast<const double2*>(C.T + c * DDT_CHUNK + DDT_KF * lane + 2); \
        K.y[dst + c][0] = lo.x; K.y[dst + c][1] = lo.y; K.y[dst + c][2] = hi.x; K.y[dst + c][3] = hi.y; \
      }                                                                                                \
      __builtin_amdgcn_wave_barrier();                                                                 \
    }
    ddt_tap_run<DBL>(C, R, ringb, lane8nb, 0, C.nE, sE);
    DDT_STAMP(2)
    DDT_TRANSPOSE(sE, 2)
    DDT_STAMP(3)
    ddt_tap_run<DBL>(C, R, ringb, lane8nb, C.nE, C.nT, sL);
    DDT_STAMP(2)
    DDT_TRANSPOSE(sL, 4)
    DDT_STAMP(3)
#undef DDT_TRANSPOSE
  }
  // direct path (:444-451)
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    // dIn = directGain * (x*(1-col) + mono*col) (:447-451) with the constant products folded (re-association, ~1e-16)
    const double x0 = (double)K.x0[k], x1 = (double)K.x1[k];
    const double mc = (x0 + x1) * C.dgm;
    K.y[0][k] = __builtin_fma(x0, C.dgc, mc);
    K.y[1][k] = __builtin_fma(x1, C.dgc, mc);
  }
  if (want_last && lane == 63) {                           // the launch's final frame: leave its temporaries in vars[]
    const int q = DDT_KF - 1;
    const double x0 = (double)K.x0[q], x1 = (double)K.x1[q], M = 0.5 * (x0 + x1);
    double* V = C.V;
    V[ZA_VAR_mono] = M;
    V[ZA_VAR_srcL] = __builtin_fma(x0, C.one_m_col, M * C.col); V[ZA_VAR_srcR] = __builtin_fma(x1, C.one_m_col, M * C.col);
    V[ZA_VAR_dInL] = C.directGain * __builtin_fma(x0, C.one_m_col, M * C.col);
    V[ZA_VAR_dInR] = C.directGain * __builtin_fma(x1, C.one_m_col, M * C.col);
    V[ZA_VAR_sumEL] = K.y[2][q]; V[ZA_VAR_sumER] = K.y[3][q]; V[ZA_VAR_sumLL] = K.y[4][q]; V[ZA_VAR_sumLR] = K.y[5][q];
  }
  // one-poles (:453-454, 486-490): local responses + wave scans
#pragma unroll
  for (int s = 0; s < 6; ++s)
    K.G[s] = ddt_pole_local<PARTIAL>(C.P[s >> 1], cb1[s >> 1], cb2[s >> 1], K.y[s], carry[s], head, lane, first_lane, first_k);
  DDT_STAMP(4)
}

// Phase C: apply the state carried into the chunk, mix (:492-505), meters (:510-536), store audio.
template <bool PARTIAL, bool METERS>
__device__ __forceinline__ void ddt_phase_c(const DdtCtx& C, DdtChunk& K, int lane, int64_t f0, const double (&cw)[6], bool chained,
                                            const double (&ql)[3], double (&accM)[6], double& accC, double dMi, double dCi,
                                            double wM, double wC, const double (&cwM)[DDT_KF], const double (&cwC)[DDT_KF],
                                            bool want_last) {
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    double cin = ddt_dpp<DDT_WAVE_SHR1, 0xF>(K.G[s]);      // y at the end of the previous lane (0 for lane 0)
    if (chained) cin = __builtin_fma(ql[s >> 1], cw[s], cin);   // + a^(4*lane) * state at the chunk's start
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) K.y[s][k] = __builtin_fma(C.P[s >> 1].ap[k], cin, K.y[s][k]);
  }
  float o0[DDT_KF], o1[DDT_KF];
  double zM[6] = {0, 0, 0, 0, 0, 0}, zC = 0.0;
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    const double dirZL = K.y[0][k], dirZR = K.y[1][k], eZL = K.y[2][k], eZR = K.y[3][k], lZL = K.y[4][k], lZR = K.y[5][k];
    const double yL = dirZL + eZL + lZL, yR = dirZR + eZR + lZR;
    const double dL = eZL + lZL, dR = eZR + lZR;
    double oL, oR;
    const double x0 = (double)K.x0[k], x1 = (double)K.x1[k];
    if (C.mon == 3) { oL = x0; oR = x1; }
    else if (C.mon == 1) { oL = dirZL; oR = dirZR; }
    else if (C.mon == 2) { oL = dL; oR = dR; }
    else { oL = yL; oR = yR; }
    double s0 = __builtin_fma(C.mixd, x0, C.mixw * oL);      // (dryp*x + wetp*o) * out_gain (:500-503), gains folded
    double s1 = __builtin_fma(C.mixd, x1, C.mixw * oR);
    s0 = s0 > 8.0 ? 8.0 : (s0 < -8.0 ? -8.0 : s0);
    s1 = s1 > 8.0 ? 8.0 : (s1 < -8.0 ? -8.0 : s1);
    o0[k] = (float)s0; o1[k] = (float)s1;
    if (!METERS) continue;
    // meters (:510-531): the 0.5 of s_dir/s_ear/s_lat is applied once to the reduced sums (exact: a power of two), and
    // the s_tot one-pole, being linear, is the sum of the three (final reduction)
    const double s_dir2 = fabs(dirZL) + fabs(dirZR);
    const double s_ear2 = fabs(eZL) + fabs(eZR);
    const double s_lat2 = fabs(lZL) + fabs(lZR);
    const double adL = fabs(dL), adR = fabs(dR);
    // c = dL*dR / max(1e-7, |dL||dR| + 1e-7) (:534): the divisor is within [1e-7, ~1e2], so a hardware reciprocal
    // refined by two Newton steps (~1e-16 relative) replaces the full IEEE division sequence; c only feeds a meter.
    const double den = __builtin_fmax(0.0000001, __builtin_fma(adL, adR, 0.0000001));
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    const double cc = (dL * dR) * rc;
    // lane-local one-pole with zero start == sum_k val_k * (1-a) a^(3-k): linear, so accumulate with weights
    zM[0] = __builtin_fma(cwM[k], s_dir2, zM[0]);
    zM[1] = __builtin_fma(cwM[k], s_ear2, zM[1]);
    zM[2] = __builtin_fma(cwM[k], s_lat2, zM[2]);
    zM[4] = __builtin_fma(cwM[k], adL, zM[4]);
    zM[5] = __builtin_fma(cwM[k], adR, zM[5]);
    zC = __builtin_fma(cwC[k], __builtin_fmin(__builtin_fmax(cc, -1.0), 1.0), zC);
    if (want_last && k == DDT_KF - 1 && lane == 63) {
      double* V = C.V;
      V[ZA_VAR_yL] = yL; V[ZA_VAR_yR] = yR; V[ZA_VAR_oL] = oL; V[ZA_VAR_oR] = oR;
      const double s_dir = 0.5 * s_dir2, s_ear = 0.5 * s_ear2, s_lat = 0.5 * s_lat2;
      V[ZA_VAR_s_dir] = s_dir; V[ZA_VAR_s_ear] = s_ear; V[ZA_VAR_s_lat] = s_lat; V[ZA_VAR_s_tot] = s_dir + s_ear + s_lat;
      V[ZA_VAR_dL] = dL; V[ZA_VAR_dR] = dR; V[ZA_VAR_c] = cc;
      C.SPL[0] = s0; C.SPL[1] = s1;
    }
  }
  if (METERS) {
#pragma unroll
    for (int q = 0; q < 6; ++q) accM[q] = __builtin_fma(accM[q], dMi, wM * zM[q]);
    accC = __builtin_fma(accC, dCi, wC * zC);
  }

  const int64_t t0 = f0 + DDT_KF * lane;
  if (C.vec_ok && (!PARTIAL || t0 >= 0)) {
    *reinterpret_cast<float4*>(C.out0 + t0) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(C.out1 + t0) = make_float4(o1[0], o1[1], o1[2], o1[3]);
  } else {
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k)
      if (t0 + k >= 0) { C.out0[t0 + k] = o0[k]; C.out1[t0 + k] = o1[k]; }
  }
}

__device__ __forceinline__ int ddt_pos(int64_t n, int W) { int r = (int)(n % W); return r < 0 ? r + W : r; }

template <int NW, bool DBL>
__device__ __forceinline__ void ddt_fast_body(const ZabBatch& b, const ZabAudio& a, int W) {
  extern __shared__ double ddt_lds[];
  double* ring = ddt_lds;                                  // [W + 256] (power-of-two W) or [2 W + 256] (doubled), see ddt_geometry
  double* Tall = ddt_lds + (DBL ? 2 * W : W) + DDT_CHUNK;  // [NW][2][256] per-wave transpose areas
  double* gend = Tall + NW * 2 * DDT_CHUNK;                // [NW][6] chunk-end responses (zero incoming state)
  double* mred = Tall;                                     // [NW][7] meter partials, after the last chunk (aliases Tall)
  DdtTap* taps = (DdtTap*)Tall;                            // [DDT_MAXTAPS] staging only (aliases Tall): early taps, then late
  DdtPole* P = (DdtPole*)(gend + NW * 6);                  // [3]
  int* scratch = (int*)(P + 3);                            // [4]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int inst = blockIdx.x;

  double* V = b.vars + (int64_t)inst * b.var_si;           // instance-major (checked by za_fast_applies)
  double* Mem = b.mem + (int64_t)inst * b.mem_si;
  const double* SL = b.sliders + (int64_t)inst * b.sl_si;
  const int64_t frames = a.frames;
  if (frames <= 0) return;

  DdtCtx C;
  C.in0 = a.in + (int64_t)inst * 2 * a.frame_stride;
  C.in1 = C.in0 + a.frame_stride;
  C.out0 = a.out + (int64_t)inst * 2 * a.frame_stride;
  C.out1 = C.out0 + a.frame_stride;
  C.V = V; C.SPL = b.spl + (int64_t)inst * b.sl_si;
  C.Mem = Mem; C.ring = ring; C.T = Tall + wave * 2 * DDT_CHUNK; C.P = P; C.frames = frames;
  C.W = W; C.m8 = 8 * (W - 1);

  // ---- per-launch scalars (workgroup-uniform) --------------------------------------------------------------------------
  const double mbase = V[ZA_VAR_m];
  C.rL = za_addr(mbase, V[ZA_VAR_bL]); C.rR = za_addr(mbase, V[ZA_VAR_bR]);
  const int64_t tDL = za_addr(mbase, V[ZA_VAR_bDL]), tDR = za_addr(mbase, V[ZA_VAR_bDR]);
  const int64_t tGL = za_addr(mbase, V[ZA_VAR_bGL]), tGR = za_addr(mbase, V[ZA_VAR_bGR]), tD0 = za_addr(mbase, V[ZA_VAR_bD0]);
  C.bufmask = za_i32(V[ZA_VAR_BUF_MASK]);
  int tapN = (int)za_loopcount(V[ZA_VAR_tapN]);
  if (tapN > DDT_MAXTAPS) tapN = DDT_MAXTAPS;
  const double splitSamp = V[ZA_VAR_splitSamp];
  C.directGain = V[ZA_VAR_directGain];
  C.wetp = V[ZA_VAR_wetp]; C.dryp = V[ZA_VAR_dryp]; C.out_gain = V[ZA_VAR_out_gain];
  const double slider1 = SL[0], slider8 = SL[7];
  C.wofs0 = za_f2i64(V[ZA_VAR_wofs]);

  // distN = smooth01(slider1/100); col = distN^0.8   (:444-446; clamp/smooth01 :62-64)
  double tt = ddt_clamp(slider1 / 100.0, 0.0, 1.0);
  const double distN = (tt * tt) * (3.0 - 2.0 * tt);
  C.col = pow(distN, 0.8);
  C.one_m_col = 1.0 - C.col;
  C.dgc = C.directGain * C.one_m_col; C.dgm = C.directGain * C.col * 0.5;
  C.mixd = C.dryp * C.out_gain; C.mixw = C.wetp * C.out_gain;
  C.mon = za_i32(slider8);

  const double poles[3] = {V[ZA_VAR_a_dir], V[ZA_VAR_a_early], V[ZA_VAR_a_late]};
  if (tid < 3) {
    DdtPole pl;
    const double pole = poles[tid];
    pl.a = pole; pl.c1 = 1.0 - pole;
    pl.ap[0] = pole; pl.ap[1] = pole * pole; pl.ap[2] = pl.ap[1] * pole; pl.ap[3] = pl.ap[1] * pl.ap[1];
    pl.sp[0] = pl.ap[3];
    for (int j = 1; j < 4; ++j) pl.sp[j] = pl.sp[j - 1] * pl.sp[j - 1];
    pl.a256 = ddt_ipow(pole, DDT_CHUNK);
    P[tid] = pl;
  }
  double carry[6] = {V[ZA_VAR_dirZL], V[ZA_VAR_dirZR], V[ZA_VAR_eZL], V[ZA_VAR_eZR], V[ZA_VAR_lZL], V[ZA_VAR_lZR]};
  // per-lane constants of the three poles: q^((l&15)+1), q^(l-31) for the scan, q^l for the cross-wave carry (q = a^4)
  double cb1[3], cb2[3], ql[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const double q = (poles[p] * poles[p]) * (poles[p] * poles[p]);
    cb1[p] = ddt_ipow(q, (lane & 15) + 1);
    cb2[p] = ddt_ipow(q, lane >= 32 ? lane - 31 : 0);
    ql[p] = ddt_ipow(q, lane);
  }

  // meters: m = (1-aM)*val + aM*m  (:128-131,518-536); six with aM, the correlation one with 0.9990
  const double aM = 0.9985, aC = 0.9990;
  const double cM = 1.0 - aM, cC = 1.0 - aC;
  const double wM = ddt_ipow((aM * aM) * (aM * aM), 63 - lane), wC = ddt_ipow((aC * aC) * (aC * aC), 63 - lane);
  const double dMi = ddt_ipow(aM, DDT_CHUNK * NW), dCi = ddt_ipow(aC, DDT_CHUNK * NW);   // a wave's chunks are NW apart
  double cwM[DDT_KF], cwC[DDT_KF];
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) { cwM[k] = cM * ddt_ipow(aM, DDT_KF - 1 - k); cwC[k] = cC * ddt_ipow(aC, DDT_KF - 1 - k); }
  double accM[6] = {0, 0, 0, 0, 0, 0}, accC = 0.0;

  // ---- stage tap lists (wave 0): early taps (baseD < splitSamp) then late taps, each in source order -------------------
  int nE = 0;
  if (wave == 0) {
    bool early = false;
    DdtTap mine = {0, 0, 0.0, 0.0};
    if (lane < tapN) {
      mine.dL8 = 8 * za_i32(Mem[tDL + lane]);
      mine.dR8 = 8 * za_i32(Mem[tDR + lane]);
      mine.gL = Mem[tGL + lane];
      mine.gR = Mem[tGR + lane];
      early = (double)za_i32(Mem[tD0 + lane]) < splitSamp;
    }
    const unsigned long long emask = __ballot(lane < tapN && early);
    const unsigned long long lmask = __ballot(lane < tapN && !early);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    nE = __popcll(emask);
    if (lane < tapN) taps[early ? __popcll(emask & below) : nE + __popcll(lmask & below)] = mine;
    if (lane == tapN - 1) { scratch[0] = mine.dL8 >> 3; scratch[1] = mine.dR8 >> 3; }   // source-order last tap
    if (lane == 0) scratch[2] = nE;
  }
  const int H = W - NW * DDT_CHUNK;                        // history frames kept (> max tap delay)
  for (int j = tid; j < H; j += NW * 64) {
    const int64_t n = C.wofs0 - H + j;
    const int64_t ri = n & C.bufmask;
    const double mv = 0.5 * (Mem[C.rL + ri] + Mem[C.rR + ri]);
    if (DBL) {
      const int slot = ddt_pos(n, W);
      ring[slot] = mv;
      ring[W + slot] = mv;
      if (slot < DDT_CHUNK) ring[2 * W + slot] = mv;
    } else {
      const int slot = (int)(n & (W - 1));
      ring[slot] = mv;
      if (slot < DDT_CHUNK) ring[W + slot] = mv;
    }
  }
  __syncthreads();
  C.nE = scratch[2];
  C.nT = tapN;
  DdtTapRegs R;                                            // lane j keeps staged tap j for the whole launch
  {
    const DdtTap t = taps[lane];                           // entries >= tapN are never broadcast
    R.dpack = (t.dL8 >> 3) | ((t.dR8 >> 3) << 16);         // delays < 16384 (checked by the plan kernel)
    R.gL = t.gL; R.gR = t.gR;
  }

  C.vec_ok = ((frames & 3) == 0) && ((a.frame_stride & 3) == 0) && ((((uintptr_t)C.in0) | ((uintptr_t)C.out0)) & 15) == 0;

  const int64_t nchunks = (frames + DDT_CHUNK - 1) / DDT_CHUNK;
  const int64_t f_first = frames - DDT_CHUNK * nchunks;    // <= 0; chunks are end-aligned
  const int64_t niter = (nchunks + NW - 1) / NW;
  DdtChunk K;
  DDT_STAMP_DECL
  int64_t my_last_chunk = -1;
  int pos = DBL ? ddt_pos(C.wofs0 + f_first + (int64_t)DDT_CHUNK * wave, W) : 0;
  // The HBM read of a wave's next chunk is issued a whole iteration ahead of its use.
  // (unconditional per lane, addresses clamped into the buffer: see ddt_ring2.hip.h on what a conditional load costs)
  ddt_f4 pf0 = {0.f, 0.f, 0.f, 0.f}, pf1 = pf0;
  auto prefetch = [&](const int64_t c) __attribute__((always_inline)) {
    if (C.vec_ok) {
      int64_t t0 = f_first + DDT_CHUNK * c + DDT_KF * lane;
      t0 = t0 < 0 ? 0 : (t0 > frames - DDT_KF ? frames - DDT_KF : t0);
      pf0 = *reinterpret_cast<const ddt_f4*>(C.in0 + t0);
      pf1 = *reinterpret_cast<const ddt_f4*>(C.in1 + t0);
    }
  };
  if (f_first >= 0) prefetch(wave);
  asm volatile("" : "+v"(pf0), "+v"(pf1));
  // One iteration = NW consecutive chunks, one per wave. PART: the launch's first chunk starts before frame 0.
  auto iteration = [&](auto part_c, const int64_t it) __attribute__((always_inline)) {
    constexpr bool PART = decltype(part_c)::value;
    const int64_t c = it * NW + wave;
    const bool active = c < nchunks;
    const int64_t f0 = f_first + DDT_CHUNK * c;
    const bool want_last = active && c == nchunks - 1;
    const bool head = (wave == 0);                         // wave 0 owns the head of this iteration's carry chain
    // ring position of the chunk's first frame: tracked incrementally for the doubled ring (W is not a power of two)
    const int nb = DBL ? pos : (int)((C.wofs0 + f0) & (W - 1));
    if (DBL) { pos += NW * DDT_CHUNK; pos = pos >= W ? pos - W : pos; }
    if (active) ddt_phase_a<PART, DBL>(C, K, lane, f0, nb, pf0, pf1);
    prefetch(c + NW);                                      // this wave's next chunk: in flight across phases B and C
    DDT_STAMP(0)
    ddt_barrier();                                         // ring holds every frame of this iteration
    DDT_STAMP(1)
    if (active) {
      ddt_phase_b<PART, DBL>(C, R, K, lane, f0, nb, carry, head, cb1, cb2, want_last DDT_STAMP_PASS);
      if (lane == 63) {
#pragma unroll
        for (int s = 0; s < 6; ++s) gend[wave * 6 + s] = K.G[s];
      }
      my_last_chunk = c;
    }
    DDT_STAMP(4)
    ddt_barrier();                                         // chunk-end responses published; taps of this iteration done
    DDT_STAMP(5)
    // carry chain: state entering wave w's chunk = a^256 * (state entering w-1) + response of w-1; wave 0 injected `carry`
    double cw[6] = {0, 0, 0, 0, 0, 0};
    {
      double run[6] = {0, 0, 0, 0, 0, 0};
      const int nact = (int)((nchunks - it * NW) < NW ? (nchunks - it * NW) : NW);
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        if (u < nact) {
          if (u == wave) {
#pragma unroll
            for (int s = 0; s < 6; ++s) cw[s] = run[s];
          }
#pragma unroll
          for (int s = 0; s < 6; ++s) run[s] = __builtin_fma(P[s >> 1].a256, run[s], gend[u * 6 + s]);
        }
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) carry[s] = run[s];       // state after this iteration's last chunk (same in every wave)
    }
    // the next chunk's audio is waited for here, before this chunk's stores join the (in-order) queue -- not at the top of the
    // loop, where the wait would be a full vmcnt(0) that includes the stores' trip to HBM
    asm volatile("" : "+v"(pf0), "+v"(pf1));
    if (active) {
      if (f0 + DDT_CHUNK > frames - DDT_METER_FRAMES)     // wave-uniform
        ddt_phase_c<PART, true>(C, K, lane, f0, cw, wave != 0, ql, accM, accC, dMi, dCi, wM, wC, cwM, cwC, want_last);
      else
        ddt_phase_c<PART, false>(C, K, lane, f0, cw, wave != 0, ql, accM, accC, dMi, dCi, wM, wC, cwM, cwC, want_last);
    }
    DDT_STAMP(6)
  };
  int64_t it = 0;
  if (f_first < 0) iteration(std::true_type{}, it++);     // workgroup-uniform
  for (; it < niter; ++it) iteration(std::false_type{}, it);

#ifdef DDT_STAMPS
  if (lane == 0) {
    for (int q = 0; q < 7; ++q) atomicAdd(&ddt_stamp_acc[q], ddt_st[q]);
    atomicAdd(&ddt_stamp_acc[15], 1ull);
  }
#endif
  // ---- meters: m_final = a^frames * m_start + sum over waves/lanes of the weighted partials ----------------------------
  double red[7];
  {
    // a wave's partials are relative to the end of ITS last chunk; bring them to the end of the launch
    const int64_t behind = my_last_chunk >= 0 ? (nchunks - 1 - my_last_chunk) * DDT_CHUNK : 0;
    const double fM = ddt_ipow(aM, behind), fC = ddt_ipow(aC, behind);
#pragma unroll
    for (int q = 0; q < 6; ++q) red[q] = accM[q] * fM;
    red[6] = accC * fC;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int q = 0; q < 7; ++q) red[q] += __shfl_xor(red[q], d, 64);
  if (lane == 0) {                                         // (every wave is past the loop's last barrier: Tall is free)
#pragma unroll
    for (int q = 0; q < 7; ++q) mred[wave * 7 + q] = red[q];
  }
  __syncthreads();

  // ---- state write-back: lane 63 of the wave that processed the launch's last chunk ------------------------------------
  const int last_wave = (int)((nchunks - 1) % NW);
  if (wave == last_wave && lane == 63) {
    double tot[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < NW; ++u)
      for (int q = 0; q < 7; ++q) tot[q] += mred[u * 7 + q];
    tot[0] *= 0.5; tot[1] *= 0.5; tot[2] *= 0.5;
    tot[3] = tot[0] + tot[1] + tot[2];
    const double pM = pow(aM, (double)frames), pC = pow(aC, (double)frames);
    V[ZA_VAR_m_dirE] = pM * V[ZA_VAR_m_dirE] + tot[0];
    V[ZA_VAR_m_earlyE] = pM * V[ZA_VAR_m_earlyE] + tot[1];
    V[ZA_VAR_m_lateE] = pM * V[ZA_VAR_m_lateE] + tot[2];
    V[ZA_VAR_m_totalE] = pM * V[ZA_VAR_m_totalE] + tot[3];
    V[ZA_VAR_m_diffL] = pM * V[ZA_VAR_m_diffL] + tot[4];
    V[ZA_VAR_m_diffR] = pM * V[ZA_VAR_m_diffR] + tot[5];
    V[ZA_VAR_m_diffCorr] = pC * V[ZA_VAR_m_diffCorr] + tot[6];
    V[ZA_VAR_dirZL] = carry[0]; V[ZA_VAR_dirZR] = carry[1];
    V[ZA_VAR_eZL] = carry[2]; V[ZA_VAR_eZR] = carry[3];
    V[ZA_VAR_lZL] = carry[4]; V[ZA_VAR_lZR] = carry[5];
    V[ZA_VAR_wofs] = V[ZA_VAR_wofs] + (double)frames;
    // @sample temporaries of the last frame, exactly as the script leaves them
    const int64_t nlast = C.wofs0 + frames - 1;
    V[ZA_VAR_distN] = distN; V[ZA_VAR_col] = C.col;
    V[ZA_VAR_i] = (double)tapN;
    if (tapN > 0) {
      const int dLl = scratch[0], dRl = scratch[1];
      const int cl = DBL ? ddt_pos(nlast - dLl, W) : (int)((nlast - dLl) & (W - 1));
      const int cr = DBL ? ddt_pos(nlast - dRl, W) : (int)((nlast - dRl) & (W - 1));
      V[ZA_VAR_idxL] = (double)(int32_t)((nlast - dLl) & C.bufmask);
      V[ZA_VAR_idxR] = (double)(int32_t)((nlast - dRl) & C.bufmask);
      V[ZA_VAR_xL] = ring[cl];                                        // the LDS ring still holds frame - delay
      V[ZA_VAR_xR] = ring[cr];
      V[ZA_VAR_gL] = Mem[tGL + tapN - 1]; V[ZA_VAR_gR] = Mem[tGR + tapN - 1];
      V[ZA_VAR_baseD] = (double)za_i32(Mem[tD0 + tapN - 1]);
    }
    V[ZA_VAR_mon] = (double)C.mon;
    V[ZA_VAR_aM] = aM;
    const int64_t hi = (C.rL > C.rR ? C.rL : C.rR) + DDT_RING;
    if (b.mem_high[inst] < hi) b.mem_high[inst] = hi;
  }
}

// _nwK: K wavefronts per instance. W < 0 selects the doubled ring of length -W (any multiple of 256), W > 0 the
// power-of-two ring of length W (long delays that do not fit twice in LDS).
#define DDT_KERNEL(name, NW)                                                                             \
  extern "C" __global__ void __launch_bounds__(64 * NW, 2) name(ZabBatch b, ZabAudio a, int W) {         \
    if (W < 0) ddt_fast_body<NW, true>(b, a, -W); else ddt_fast_body<NW, false>(b, a, W);                \
  }
DDT_KERNEL(zab_ddt_wide, 1)
DDT_KERNEL(zab_ddt_wide_nw2, 2)
DDT_KERNEL(zab_ddt_wide_nw4, 4)
DDT_KERNEL(zab_ddt_wide_nw8, 8)
#undef DDT_KERNEL

// ---- plan: max tap delay over the batch (decides the LDS ring length) -----------------------------------------------
__device__ int ddt_plan_word[2];
extern "C" __global__ void zab_ddt_plan(ZabBatch b) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= b.n_inst) return;
  const double* V = b.vars + (int64_t)inst * b.var_si;
  const double* Mem = b.mem + (int64_t)inst * b.mem_si;
  const double mbase = V[ZA_VAR_m];
  const int64_t tDL = za_addr(mbase, V[ZA_VAR_bDL]), tDR = za_addr(mbase, V[ZA_VAR_bDR]);
  int tapN = (int)za_loopcount(V[ZA_VAR_tapN]);
  int bad = 0;
  if (tapN > DDT_MAXTAPS) bad = 1;
  if (za_i32(V[ZA_VAR_BUF_MASK]) != DDT_RING - 1) bad = 1;
  if (tDL + DDT_MAXTAPS > b.mem_cap || tDR + DDT_MAXTAPS > b.mem_cap) bad = 1;
  int dmax = 0;
  if (!bad)
    for (int i = 0; i < tapN; ++i) {
      const int dl = za_i32(Mem[tDL + i]), dr = za_i32(Mem[tDR + i]);
      if (dl < 0 || dr < 0) bad = 1;
      dmax = dl > dmax ? dl : dmax;
      dmax = dr > dmax ? dr : dmax;
    }
  atomicMax(&ddt_plan_word[0], dmax);
  if (bad) atomicMax(&ddt_plan_word[1], 1);
}

#ifdef DDT_STAMPS
extern "C" int zab_ddt_stamps(unsigned long long* out, int reset) {      // [16]: cycles per phase summed over waves, [15] = waves
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ddt_stamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ddt_stamp_acc), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
struct DdtPlan { uint64_t epoch; int dmax; bool ok; };
static std::mutex ddt_mu;
// one plan per engine: keyed by device AND state pointer (device pointers of different GPUs can coincide: zab_group runs one
// engine per GPU in one process) and valid for the engine's current epoch only (sliders / state uploads bump it)
static std::map<std::pair<int, const void*>, DdtPlan> ddt_plans;

static DdtPlan ddt_plan(const ZabBatch* b) {
  std::lock_guard<std::mutex> lk(ddt_mu);
  int dev = -1;
  (void)hipGetDevice(&dev);
  const auto key = std::make_pair(dev, (const void*)b->vars);
  auto it = ddt_plans.find(key);
  if (it != ddt_plans.end() && it->second.epoch == b->epoch) return it->second;
  int zero[2] = {0, 0}, res[2] = {0, 1};
  DdtPlan p{b->epoch, 0, false};
  if (hipMemcpyToSymbol(HIP_SYMBOL(ddt_plan_word), zero, sizeof zero) == hipSuccess) {
    hipLaunchKernelGGL(zab_ddt_plan, dim3((b->n_inst + 255) / 256), dim3(256), 0, 0, *b);
    if (hipMemcpyFromSymbol(res, HIP_SYMBOL(ddt_plan_word), sizeof res) == hipSuccess) { p.dmax = res[0]; p.ok = !res[1]; }
  }
  ddt_plans[key] = p;
  return p;
}

// waves per instance: the register budget allows two waves per SIMD, so aim at 2 x 1024 SIMDs resident waves and no more
// (measured on MI355X, 480 000 frames: N=1024 -> NW 1/2/4 = 7.9/5.95/6.2 ms; N=4096 -> NW 1/2 = 20.8/22.6 ms)
static int ddt_pick_nw(int n_inst) {
  int nw = 1;
  while (nw < DDT_MAXNW && (int64_t)n_inst * nw < 2048) nw <<= 1;
  if (const char* e = getenv("ZAB_DDT_NW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) nw = v; }
  return nw;
}
static size_t ddt_lds_bytes(int W, int nw, bool dbl) {
  return (size_t)((dbl ? 2 * W : W) + DDT_CHUNK + nw * 2 * DDT_CHUNK + nw * 6) * sizeof(double) + 3 * sizeof(DdtPole) + 16;
}
// Ring geometry and waves per instance. The ring must hold Dmax + NW*256 + 1 frames. Preferred: a DOUBLED ring of exactly
// that length rounded up to 256 (returned as -W): `frame - delay` then needs no wrap at all in the tap phase. When twice
// that does not fit in LDS: a power-of-two ring with masked offsets (returned as +W). 0: not applicable (generic kernel).
static void ddt_geometry(const ZabBatch* b, int& W, int& nw) {
  const DdtPlan p = ddt_plan(b);
  W = 0; nw = 1;
  if (!p.ok) return;
  const size_t cap = 160 * 1024 - 512, cu = 160 * 1024;
  const char* force = getenv("ZAB_DDT_RING");              // "dbl" / "pow2": tests pin the addressing mode
  for (nw = ddt_pick_nw(b->n_inst); nw >= 1; nw >>= 1) {
    const int need = p.dmax + nw * DDT_CHUNK + 1;
    const int wd = (need + DDT_CHUNK - 1) / DDT_CHUNK * DDT_CHUNK;
    int w = 1024;
    while (w < need) w <<= 1;
    const size_t ld = ddt_lds_bytes(wd, nw, true), lp = ddt_lds_bytes(w, nw, false);
    const bool fd = ld <= cap && !(force && !strcmp(force, "pow2")), fp = w <= 16384 && lp <= cap && !(force && !strcmp(force, "dbl"));
    // the register budget allows 8 wavefronts per CU; an LDS footprint that admits fewer costs more than the wrap does
    const bool full_d = fd && (cu / ld) * nw >= 8, full_p = fp && (cu / lp) * nw >= 8;
    if (full_d || (fd && !full_p)) { W = -wd; return; }
    if (fp) { W = w; return; }
  }
  nw = 1;
}

static hipError_t ddt_wide_launch(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  int W, nw;
  ddt_geometry(b, W, nw);
  if (W == 0) return hipErrorInvalidValue;
  const size_t lds = ddt_lds_bytes(W < 0 ? -W : W, nw, W < 0);
  static ZaPerDevice once;               // (function attributes are per device: a group runs one engine per GPU)
  once.once([] {
    const int cap = 160 * 1024 - 512;
    (void)hipFuncSetAttribute((const void*)zab_ddt_wide, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)zab_ddt_wide_nw2, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)zab_ddt_wide_nw4, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)zab_ddt_wide_nw8, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  });
  const dim3 grid(b->n_inst), block(64 * nw);
  snprintf(ddt_kernel_name, sizeof ddt_kernel_name, nw == 1 ? "zab_ddt_wide" : "zab_ddt_wide_nw%d", nw);
  switch (nw) {
    case 1: hipLaunchKernelGGL(zab_ddt_wide, grid, block, lds, st, *b, *a, W); break;
    case 2: hipLaunchKernelGGL(zab_ddt_wide_nw2, grid, block, lds, st, *b, *a, W); break;
    case 4: hipLaunchKernelGGL(zab_ddt_wide_nw4, grid, block, lds, st, *b, *a, W); break;
    default: hipLaunchKernelGGL(zab_ddt_wide_nw8, grid, block, lds, st, *b, *a, W); break;
  }
  return hipGetLastError();
}
