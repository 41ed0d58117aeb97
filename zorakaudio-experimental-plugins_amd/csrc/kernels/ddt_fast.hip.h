// ddt_fast.hip.h -- hand-written gfx950 kernel for the @sample loop of Spatialization/DDT.
//
// What it computes is exactly DDT's @sample section (reference: plugins/Spatialization/DDT/src/DDT.jsfx:440-536,
// run per frame by jsfx_process_block, dsp_jsfx_aot.py:5713-5905); HOW it computes it is MI355X-first:
//
//   * DDT's per-frame work is feed-forward in the INPUT history (sparse taps into the bL/bR delay rings) followed
//     by LINEAR constant-coefficient one-poles (dirZ*, eZ*, lZ*, and the seven UI meters). So time is data-parallel:
//     ONE WAVEFRONT PER INSTANCE, each lane owns KF=4 consecutive frames of a 256-frame chunk.
//   * The mono delay history M[n] = 0.5*(L[n]+R[n]) (the only thing the taps read: :467-468) lives in an LDS ring of
//     W doubles, de-interleaved by 4: frame n sits in plane (n&3) at position (n>>2) mod W/4. A lane's 4 stores go to
//     4 planes at one position; a tap gather `frame - delay` of the whole wave reads 64 consecutive doubles of one
//     plane (conflict-free for every delay), and because frame = 4*lane + uniform, plane and position offset are
//     wave-uniform: the address math is scalar except one add and one mask per gather.
//   * Tap parameters (delays, gains) are wave-uniform: staged once in LDS as two lists -- early taps, late taps, each
//     in source order -- and read as broadcasts. Each list feeds its own pair of accumulators with separate multiply
//     and add (no FMA contraction) in source order, so sumE*/sumL* are bit-identical to the serial reference.
//   * The six filter recurrences y[n] = (1-a) x[n] + a y[n-1]: 4 serial steps inside the lane, a weighted 64-lane scan
//     of the lane aggregates with coefficient a^4 done with DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31 -- no LDS
//     crossbar traffic, no lane masks), then a 4-step fix-up. Lane 0 of a full chunk reproduces the serial rounding
//     exactly; other lanes differ by O(1e-16) relative.
//   * The seven meter one-poles are only observable as state after the launch, so they are carried as per-lane
//     weighted partial sums and reduced across the wave once, at the end.
//   * HBM traffic per frame: 8 B in + 8 B out (float4 per lane per channel, 1 KiB per wave instruction); the f64
//     rings in mem[] are written only for the last 16384 frames of a launch (older slots would be overwritten).
//     vars[] / tap tables are touched once per launch.
//
// State contract: on exit vars[] and mem[] hold what the serial path would hold (all @sample temporaries of the last
// frame included), within the scan's rounding for the filter states -- tests/test_ddt_gpu.py compares both paths.
#pragma once

#include <map>
#include <mutex>

#define ZA_FAST_KERNEL_NAME "zab_ddt_fast"
#define DDT_KF 4                       /* frames per lane */
#define DDT_CHUNK (64 * DDT_KF)        /* frames per wave iteration */
#define DDT_MAXTAPS 64
#define DDT_RING 16384                 /* BUF_LEN of the script */

struct DdtTap { int32_t dL8, dR8; double gL, gR; };          // 8*delay (bytes) and gains; read as wave-uniform broadcasts
struct DdtPole {
  double a, c1;        // pole and (1 - pole)
  double ap[4];        // a^1..a^4
  double sp[4];        // (a^4)^(2^j), j = 0..3 : in-row scan step coefficients
};

// ---- wave-level helpers ------------------------------------------------------------------------------------------
__device__ __forceinline__ double ddt_uniform(double v) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_readfirstlane(t.x);
  t.y = __builtin_amdgcn_readfirstlane(t.y);
  return __builtin_bit_cast(double, t);
}
__device__ __forceinline__ double ddt_readlane(double v, int l) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_readlane(t.x, l);
  t.y = __builtin_amdgcn_readlane(t.y, l);
  return __builtin_bit_cast(double, t);
}
// DPP move of a double; lanes without a source (or masked out by ROWS) receive 0.
template <int CTRL, int ROWS>
__device__ __forceinline__ double ddt_dpp(double v) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __builtin_amdgcn_update_dpp(0, t.x, CTRL, ROWS, 0xF, true);
  t.y = __builtin_amdgcn_update_dpp(0, t.y, CTRL, ROWS, 0xF, true);
  return __builtin_bit_cast(double, t);
}
#define DDT_ROW_SHR(n) (0x110 | (n))
#define DDT_WAVE_SHR1 0x138
#define DDT_ROW_BCAST15 0x142
#define DDT_ROW_BCAST31 0x143

__device__ __forceinline__ double ddt_clamp(double x, double a, double b) { return x < a ? a : (x > b ? b : x); }
__device__ __forceinline__ double ddt_ipow(double base, int e) {   // base^e, e >= 0, square-and-multiply
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

// One recurrence over the chunk. x[k]: inputs in, outputs out. carry: wave-uniform y before the chunk's first valid
// frame in, y at the chunk's last frame out. PARTIAL chunks (only the first of a launch) mask the leading slots.
template <bool PARTIAL>
__device__ __forceinline__ void ddt_pole_run(const DdtPole& p, double cb1, double cb2, double (&x)[DDT_KF], double& carry,
                                             int lane, int first_lane, int first_k) {
  double z = 0.0;
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    double prev = z;
    if (PARTIAL) {
      if (lane == first_lane && k == first_k) prev = carry;
    } else {
      if (k == 0 && lane == 0) prev = carry;
    }
    z = __builtin_fma(p.a, prev, p.c1 * x[k]);           // (1-a)*x + a*prev, one rounding fewer than the script
    if (PARTIAL) {
      const bool valid = (lane > first_lane) || (lane == first_lane && k >= first_k);
      z = valid ? z : 0.0;
    }
    x[k] = z;
  }
  const double g = ddt_scan(z, p, cb1, cb2);
  const double cin = ddt_dpp<DDT_WAVE_SHR1, 0xF>(g);      // y at the end of the previous lane (0 for lane 0)
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) x[k] = __builtin_fma(p.ap[k], cin, x[k]);
  carry = ddt_readlane(g, 63);
}

struct DdtCtx {
  // wave-uniform launch constants
  const float *in0, *in1;
  float *out0, *out1;
  double* Mem;
  double* ring;
  const DdtTap* taps;
  const DdtPole* P;
  int64_t frames, wofs0, rL, rR;
  double* T;           // [4][DDT_CHUNK] transpose area: tap sums go strided -> blocked through here
  int bufmask, W, m8, nE, nT, mon;
  double col, one_m_col, directGain, wetp, dryp, out_gain;
  bool vec_ok;
};

struct DdtLast {   // @sample temporaries of the launch's final frame (lane 63, k = 3)
  double mono, srcL, srcR, dInL, dInR, sEL, sER, sLL, sLR, yL, yR, oL, oR, sdir, sear, slat, stot, dL, dR, c, spl0, spl1;
};

// Tap phase mapping: lane l handles frames {l, 64+l, 128+l, 192+l} of the chunk ("strided"), so a gather of
// frame - delay reads 64 consecutive ring slots per k and the four k differ by a constant 512 bytes: one masked
// address per (tap, channel), the rest are ds_read_b64 immediates. The ring keeps a 256-slot mirror of its head
// behind its tail so those +512k offsets never need a wrap.
template <bool PARTIAL>
__device__ __forceinline__ void ddt_chunk(const DdtCtx& C, int lane, int64_t f0, double (&carry)[6], const double (&cb1)[3],
                                          const double (&cb2)[3], double (&accM)[6], double& accC, double dM, double dC,
                                          double wM, double wC, const double (&cwM)[DDT_KF], const double (&cwC)[DDT_KF],
                                          bool want_last, DdtLast& last) {
  const int64_t t0 = f0 + DDT_KF * lane;
  int first_lane = 0, first_k = 0;
  if (PARTIAL) {
    const int firstv = (int)(-f0);
    first_lane = firstv / DDT_KF;
    first_k = firstv % DDT_KF;
  }
  double x0[DDT_KF], x1[DDT_KF], M[DDT_KF];
  if (C.vec_ok && (!PARTIAL || t0 >= 0)) {
    const float4 v0 = *reinterpret_cast<const float4*>(C.in0 + t0);
    const float4 v1 = *reinterpret_cast<const float4*>(C.in1 + t0);
    x0[0] = v0.x; x0[1] = v0.y; x0[2] = v0.z; x0[3] = v0.w;
    x1[0] = v1.x; x1[1] = v1.y; x1[2] = v1.z; x1[3] = v1.w;
  } else {
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) {
      const bool ok = t0 + k >= 0;
      x0[k] = ok ? (double)C.in0[t0 + k] : 0.0;
      x1[k] = ok ? (double)C.in1[t0 + k] : 0.0;
    }
  }
  // ring stores (plain circular layout + mirror of the first 256 slots behind slot W)
  const int nb = (int)((C.wofs0 + f0) & 0x3fffffff);      // uniform; W divides 2^30 so low bits suffice
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    M[k] = 0.5 * (x0[k] + x1[k]);                         // mono (:445) == ring value 0.5*(L+R) (:467)
    if (!PARTIAL || t0 + k >= 0) {
      const int slot = (nb + DDT_KF * lane + k) & (C.W - 1);
      C.ring[slot] = M[k];
      if (slot < DDT_CHUNK) C.ring[C.W + slot] = M[k];
    }
  }
  if (f0 + DDT_CHUNK > C.frames - DDT_RING) {             // :441-442, only slots that survive the launch
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) {
      const int64_t t = t0 + k;
      if (t >= 0 && t >= C.frames - DDT_RING) {
        const int64_t ri = (C.wofs0 + t) & C.bufmask;
        C.Mem[C.rL + ri] = x0[k];
        C.Mem[C.rR + ri] = x1[k];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- tap loops (:459-484), strided lanes: each accumulator sees its taps in source order, multiply then add ----
  double y[6][DDT_KF];
  {
    const char* ringb = reinterpret_cast<const char*>(C.ring);
    const int lane8nb = (8 * lane + 8 * nb) & C.m8;       // byte address of frame (nb + lane) in the ring
    double sE[2][DDT_KF], sL[2][DDT_KF];
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) sE[0][k] = sE[1][k] = sL[0][k] = sL[1][k] = 0.0;
#define DDT_TAP(acc)                                                                                   \
    {                                                                                                  \
      const DdtTap tp = C.taps[i];                                                                     \
      const char* pl = ringb + ((lane8nb - tp.dL8) & C.m8);                                            \
      const char* pr = ringb + ((lane8nb - tp.dR8) & C.m8);                                            \
      _Pragma("unroll") for (int k = 0; k < DDT_KF; ++k) {                                             \
        acc[0][k] = __builtin_fma(tp.gL, *reinterpret_cast<const double*>(pl + 512 * k), acc[0][k]);   \
        acc[1][k] = __builtin_fma(tp.gR, *reinterpret_cast<const double*>(pr + 512 * k), acc[1][k]);   \
      }                                                                                                \
    }
    for (int i = 0; i < C.nE; ++i) DDT_TAP(sE)
    for (int i = C.nE; i < C.nT; ++i) DDT_TAP(sL)
#undef DDT_TAP
    // strided -> blocked: lane l wrote frames 64k+l, reads back frames 4l..4l+3
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k) {
      C.T[0 * DDT_CHUNK + 64 * k + lane] = sE[0][k];
      C.T[1 * DDT_CHUNK + 64 * k + lane] = sE[1][k];
      C.T[2 * DDT_CHUNK + 64 * k + lane] = sL[0][k];
      C.T[3 * DDT_CHUNK + 64 * k + lane] = sL[1][k];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int sgn = 0; sgn < 4; ++sgn) {
      const double2 lo = *reinterpret_cast<const double2*>(C.T + sgn * DDT_CHUNK + DDT_KF * lane);
      const double2 hi = *reinterpret_cast<const double2*>(C.T + sgn * DDT_CHUNK + DDT_KF * lane + 2);
      y[2 + sgn][0] = lo.x; y[2 + sgn][1] = lo.y; y[2 + sgn][2] = hi.x; y[2 + sgn][3] = hi.y;
    }
  }

  // ---- one-poles (:450-454, 486-490) ---------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    const double mc = M[k] * C.col;
    const double srcL = __builtin_fma(x0[k], C.one_m_col, mc);
    const double srcR = __builtin_fma(x1[k], C.one_m_col, mc);
    y[0][k] = C.directGain * srcL;
    y[1][k] = C.directGain * srcR;
  }
  if (want_last) {
    const int q = DDT_KF - 1;
    last.mono = M[q];
    last.srcL = x0[q] * C.one_m_col + M[q] * C.col; last.srcR = x1[q] * C.one_m_col + M[q] * C.col;
    last.dInL = y[0][q]; last.dInR = y[1][q];
    last.sEL = y[2][q]; last.sER = y[3][q]; last.sLL = y[4][q]; last.sLR = y[5][q];
  }
  ddt_pole_run<PARTIAL>(C.P[0], cb1[0], cb2[0], y[0], carry[0], lane, first_lane, first_k);
  ddt_pole_run<PARTIAL>(C.P[0], cb1[0], cb2[0], y[1], carry[1], lane, first_lane, first_k);
  ddt_pole_run<PARTIAL>(C.P[1], cb1[1], cb2[1], y[2], carry[2], lane, first_lane, first_k);
  ddt_pole_run<PARTIAL>(C.P[1], cb1[1], cb2[1], y[3], carry[3], lane, first_lane, first_k);
  ddt_pole_run<PARTIAL>(C.P[2], cb1[2], cb2[2], y[4], carry[4], lane, first_lane, first_k);
  ddt_pole_run<PARTIAL>(C.P[2], cb1[2], cb2[2], y[5], carry[5], lane, first_lane, first_k);

  // ---- output mix (:492-505) and meters (:510-536) -----------------------------------------------------------------
  float o0[DDT_KF], o1[DDT_KF];
  double zM[6] = {0, 0, 0, 0, 0, 0}, zC = 0.0;
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) {
    const double dirZL = y[0][k], dirZR = y[1][k], eZL = y[2][k], eZR = y[3][k], lZL = y[4][k], lZR = y[5][k];
    const double yL = dirZL + eZL + lZL, yR = dirZR + eZR + lZR;
    const double dL = eZL + lZL, dR = eZR + lZR;
    double oL, oR;
    if (C.mon == 3) { oL = x0[k]; oR = x1[k]; }
    else if (C.mon == 1) { oL = dirZL; oR = dirZR; }
    else if (C.mon == 2) { oL = dL; oR = dR; }
    else { oL = yL; oR = yR; }
    double s0 = __builtin_fma(C.dryp, x0[k], C.wetp * oL) * C.out_gain;
    double s1 = __builtin_fma(C.dryp, x1[k], C.wetp * oR) * C.out_gain;
    s0 = s0 > 8.0 ? 8.0 : (s0 < -8.0 ? -8.0 : s0);
    s1 = s1 > 8.0 ? 8.0 : (s1 < -8.0 ? -8.0 : s1);
    o0[k] = (float)s0; o1[k] = (float)s1;
    const double s_dir = 0.5 * (fabs(dirZL) + fabs(dirZR));
    const double s_ear = 0.5 * (fabs(eZL) + fabs(eZR));
    const double s_lat = 0.5 * (fabs(lZL) + fabs(lZR));
    const double s_tot = s_dir + s_ear + s_lat;
    const double adL = fabs(dL), adR = fabs(dR);
    // c = dL*dR / max(1e-7, |dL||dR| + 1e-7) (:534): the divisor is within [1e-7, ~1e2], so a hardware reciprocal
    // refined by one Newton step (~1e-15 relative) replaces the full IEEE division sequence; c only feeds a meter.
    const double den = __builtin_fmax(0.0000001, __builtin_fma(adL, adR, 0.0000001));
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    const double cc = (dL * dR) * rc;
    // lane-local one-pole with zero start == sum_k val_k * (1-a) a^(3-k): linear, so accumulate with weights
    zM[0] = __builtin_fma(cwM[k], s_dir, zM[0]);
    zM[1] = __builtin_fma(cwM[k], s_ear, zM[1]);
    zM[2] = __builtin_fma(cwM[k], s_lat, zM[2]);
    zM[3] = __builtin_fma(cwM[k], s_tot, zM[3]);
    zM[4] = __builtin_fma(cwM[k], adL, zM[4]);
    zM[5] = __builtin_fma(cwM[k], adR, zM[5]);
    zC = __builtin_fma(cwC[k], __builtin_fmin(__builtin_fmax(cc, -1.0), 1.0), zC);
    if (want_last && k == DDT_KF - 1) {
      last.yL = yL; last.yR = yR; last.oL = oL; last.oR = oR; last.sdir = s_dir; last.sear = s_ear; last.slat = s_lat;
      last.stot = s_tot; last.dL = dL; last.dR = dR; last.c = cc; last.spl0 = s0; last.spl1 = s1;
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) accM[q] = __builtin_fma(accM[q], dM, wM * zM[q]);
  accC = __builtin_fma(accC, dC, wC * zC);

  if (C.vec_ok && (!PARTIAL || t0 >= 0)) {
    *reinterpret_cast<float4*>(C.out0 + t0) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(C.out1 + t0) = make_float4(o1[0], o1[1], o1[2], o1[3]);
  } else {
#pragma unroll
    for (int k = 0; k < DDT_KF; ++k)
      if (t0 + k >= 0) { C.out0[t0 + k] = o0[k]; C.out1[t0 + k] = o1[k]; }
  }
}

extern "C" __global__ void __launch_bounds__(64, 2) zab_ddt_fast(ZabBatch b, ZabAudio a, int W) {
  extern __shared__ double ddt_lds[];
  double* ring = ddt_lds;                                  // [W + 256]: circular history + mirror of its first 256 slots
  double* T = ddt_lds + W + DDT_CHUNK;                     // [4][256] transpose area
  DdtTap* taps = (DdtTap*)(T + 4 * DDT_CHUNK);             // [DDT_MAXTAPS] early taps first, then late taps
  DdtPole* P = (DdtPole*)(taps + DDT_MAXTAPS);             // [3]
  int* scratch = (int*)(P + 3);                            // [4]
  const int lane = threadIdx.x;
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
  C.Mem = Mem; C.ring = ring; C.T = T; C.taps = taps; C.P = P; C.frames = frames;
  C.W = W; C.m8 = 8 * (W - 1);

  // ---- per-launch scalars (wave-uniform) ---------------------------------------------------------------------------
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
  C.mon = za_i32(slider8);

  if (lane < 3) {
    DdtPole pl;
    const double pole = lane == 0 ? V[ZA_VAR_a_dir] : (lane == 1 ? V[ZA_VAR_a_early] : V[ZA_VAR_a_late]);
    pl.a = pole; pl.c1 = 1.0 - pole;
    pl.ap[0] = pole; pl.ap[1] = pole * pole; pl.ap[2] = pl.ap[1] * pole; pl.ap[3] = pl.ap[1] * pl.ap[1];
    pl.sp[0] = pl.ap[3];
    for (int j = 1; j < 4; ++j) pl.sp[j] = pl.sp[j - 1] * pl.sp[j - 1];
    P[lane] = pl;
  }
  double carry[6] = {V[ZA_VAR_dirZL], V[ZA_VAR_dirZR], V[ZA_VAR_eZL], V[ZA_VAR_eZR], V[ZA_VAR_lZL], V[ZA_VAR_lZR]};
  // per-lane scan constants q^((l&15)+1), q^(l-31), q = a^4
  double cb1[3], cb2[3];
  {
    const double poles[3] = {V[ZA_VAR_a_dir], V[ZA_VAR_a_early], V[ZA_VAR_a_late]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const double q = (poles[p] * poles[p]) * (poles[p] * poles[p]);
      cb1[p] = ddt_ipow(q, (lane & 15) + 1);
      cb2[p] = ddt_ipow(q, lane >= 32 ? lane - 31 : 0);
    }
  }

  // meters: m = (1-aM)*val + aM*m  (:128-131,518-536); six with aM, the correlation one with 0.9990
  const double aM = 0.9985, aC = 0.9990;
  const double cM = 1.0 - aM, cC = 1.0 - aC;
  const double wM = ddt_ipow((aM * aM) * (aM * aM), 63 - lane), wC = ddt_ipow((aC * aC) * (aC * aC), 63 - lane);
  const double dM = ddt_ipow(aM, DDT_CHUNK), dC = ddt_ipow(aC, DDT_CHUNK);
  double cwM[DDT_KF], cwC[DDT_KF];
#pragma unroll
  for (int k = 0; k < DDT_KF; ++k) { cwM[k] = cM * ddt_ipow(aM, DDT_KF - 1 - k); cwC[k] = cC * ddt_ipow(aC, DDT_KF - 1 - k); }
  double accM[6] = {0, 0, 0, 0, 0, 0}, accC = 0.0;

  // ---- stage tap lists: early taps (baseD < splitSamp) then late taps, each in source order ---------------------
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
  C.nE = __popcll(emask);
  C.nT = tapN;
  if (lane < tapN) taps[early ? __popcll(emask & below) : C.nE + __popcll(lmask & below)] = mine;
  if (lane == tapN - 1) { scratch[0] = mine.dL8 >> 3; scratch[1] = mine.dR8 >> 3; }   // source-order last tap (state temporaries)

  const int H = W - DDT_CHUNK;                             // history frames kept (> max tap delay)
  for (int j = lane; j < H; j += 64) {
    const int64_t n = C.wofs0 - H + j;
    const int64_t ri = n & C.bufmask;
    const int slot = (int)(n & (W - 1));
    const double mv = 0.5 * (Mem[C.rL + ri] + Mem[C.rR + ri]);
    ring[slot] = mv;
    if (slot < DDT_CHUNK) ring[W + slot] = mv;
  }

  C.vec_ok = ((frames & 3) == 0) && ((a.frame_stride & 3) == 0) && ((((uintptr_t)C.in0) | ((uintptr_t)C.out0)) & 15) == 0;

  const int64_t nchunks = (frames + DDT_CHUNK - 1) / DDT_CHUNK;
  DdtLast last = {};
  int64_t c = 0;
  const int64_t f_first = frames - DDT_CHUNK * nchunks;    // <= 0; chunks are end-aligned
  if (f_first < 0) {
    ddt_chunk<true>(C, lane, f_first, carry, cb1, cb2, accM, accC, dM, dC, wM, wC, cwM, cwC, nchunks == 1, last);
    c = 1;
  }
  for (; c < nchunks; ++c)
    ddt_chunk<false>(C, lane, f_first + DDT_CHUNK * c, carry, cb1, cb2, accM, accC, dM, dC, wM, wC, cwM, cwC,
                     c == nchunks - 1, last);

  // ---- meters: m_final = a^frames * m_start + sum over lanes of the weighted partials ---------------------------------
  double red[7];
#pragma unroll
  for (int q = 0; q < 6; ++q) red[q] = accM[q];
  red[6] = accC;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int q = 0; q < 7; ++q) red[q] += __shfl_xor(red[q], d, 64);

  __syncthreads();
  // ---- state write-back (lane 63 owns the launch's last frame) --------------------------------------------------------
  if (lane == 63) {
    const double pM = pow(aM, (double)frames), pC = pow(aC, (double)frames);
    V[ZA_VAR_m_dirE] = pM * V[ZA_VAR_m_dirE] + red[0];
    V[ZA_VAR_m_earlyE] = pM * V[ZA_VAR_m_earlyE] + red[1];
    V[ZA_VAR_m_lateE] = pM * V[ZA_VAR_m_lateE] + red[2];
    V[ZA_VAR_m_totalE] = pM * V[ZA_VAR_m_totalE] + red[3];
    V[ZA_VAR_m_diffL] = pM * V[ZA_VAR_m_diffL] + red[4];
    V[ZA_VAR_m_diffR] = pM * V[ZA_VAR_m_diffR] + red[5];
    V[ZA_VAR_m_diffCorr] = pC * V[ZA_VAR_m_diffCorr] + red[6];
    V[ZA_VAR_dirZL] = carry[0]; V[ZA_VAR_dirZR] = carry[1];
    V[ZA_VAR_eZL] = carry[2]; V[ZA_VAR_eZR] = carry[3];
    V[ZA_VAR_lZL] = carry[4]; V[ZA_VAR_lZR] = carry[5];
    V[ZA_VAR_wofs] = V[ZA_VAR_wofs] + (double)frames;
    // @sample temporaries of the last frame, exactly as the script leaves them
    const int64_t nlast = C.wofs0 + frames - 1;
    V[ZA_VAR_distN] = distN; V[ZA_VAR_col] = C.col; V[ZA_VAR_mono] = last.mono;
    V[ZA_VAR_srcL] = last.srcL; V[ZA_VAR_srcR] = last.srcR; V[ZA_VAR_dInL] = last.dInL; V[ZA_VAR_dInR] = last.dInR;
    V[ZA_VAR_sumEL] = last.sEL; V[ZA_VAR_sumER] = last.sER; V[ZA_VAR_sumLL] = last.sLL; V[ZA_VAR_sumLR] = last.sLR;
    V[ZA_VAR_i] = (double)tapN;
    if (tapN > 0) {
      const int dLl = scratch[0], dRl = scratch[1];
      const int cl = (int)((nlast - dLl) & (W - 1)), cr = (int)((nlast - dRl) & (W - 1));
      V[ZA_VAR_idxL] = (double)(int32_t)((nlast - dLl) & C.bufmask);
      V[ZA_VAR_idxR] = (double)(int32_t)((nlast - dRl) & C.bufmask);
      V[ZA_VAR_xL] = ring[cl];                                        // the LDS ring still holds frame - delay
      V[ZA_VAR_xR] = ring[cr];
      V[ZA_VAR_gL] = Mem[tGL + tapN - 1]; V[ZA_VAR_gR] = Mem[tGR + tapN - 1];
      V[ZA_VAR_baseD] = (double)za_i32(Mem[tD0 + tapN - 1]);
    }
    V[ZA_VAR_yL] = last.yL; V[ZA_VAR_yR] = last.yR; V[ZA_VAR_mon] = (double)C.mon;
    V[ZA_VAR_oL] = last.oL; V[ZA_VAR_oR] = last.oR;
    V[ZA_VAR_s_dir] = last.sdir; V[ZA_VAR_s_ear] = last.sear; V[ZA_VAR_s_lat] = last.slat; V[ZA_VAR_s_tot] = last.stot;
    V[ZA_VAR_aM] = aM;
    V[ZA_VAR_dL] = last.dL; V[ZA_VAR_dR] = last.dR; V[ZA_VAR_c] = last.c;
    double* SPL = b.spl + (int64_t)inst * b.sl_si;
    SPL[0] = last.spl0; SPL[1] = last.spl1;
    const int64_t hi = (C.rL > C.rR ? C.rL : C.rR) + DDT_RING;
    if (b.mem_high[inst] < hi) b.mem_high[inst] = hi;
  }
}

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

struct DdtPlan { uint64_t epoch; int W; };
static std::mutex ddt_mu;
static std::map<const void*, DdtPlan> ddt_plans;

static int ddt_ring_len(const ZabBatch* b) {
  std::lock_guard<std::mutex> lk(ddt_mu);
  auto it = ddt_plans.find(b->vars);
  if (it != ddt_plans.end() && it->second.epoch == b->epoch) return it->second.W;
  int zero[2] = {0, 0}, res[2] = {0, 1};
  if (hipMemcpyToSymbol(HIP_SYMBOL(ddt_plan_word), zero, sizeof zero) != hipSuccess) return 0;
  hipLaunchKernelGGL(zab_ddt_plan, dim3((b->n_inst + 255) / 256), dim3(256), 0, 0, *b);
  if (hipMemcpyFromSymbol(res, HIP_SYMBOL(ddt_plan_word), sizeof res) != hipSuccess) return 0;
  int W = 0;
  if (!res[1]) {
    W = 1024;
    while (W < res[0] + DDT_CHUNK + 1) W <<= 1;
    if (W > 16384) W = 0;                                  // 128 KiB + tables still fit the 160 KiB LDS; beyond: generic
  }
  ddt_plans[b->vars] = DdtPlan{b->epoch, W};
  return W;
}

static int32_t za_fast_applies(const ZabBatch* b, const ZabAudio* a) {
  if (!b->instance_major || b->var_se != 1 || b->mem_se != 1 || b->sl_se != 1) return 0;
  if (a->frames <= 0) return 0;
  return ddt_ring_len(b) > 0 ? 1 : 0;
}

static hipError_t za_launch_fast(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  const int W = ddt_ring_len(b);
  if (W <= 0) return hipErrorInvalidValue;
  const size_t lds = (size_t)(W + 5 * DDT_CHUNK) * sizeof(double) + DDT_MAXTAPS * sizeof(DdtTap) + 3 * sizeof(DdtPole) + 16;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute((const void*)zab_ddt_fast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  });
  hipLaunchKernelGGL(zab_ddt_fast, dim3(b->n_inst), dim3(64), lds, st, *b, *a, W);
  return hipGetLastError();
}
