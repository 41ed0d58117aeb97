// ddt_ring2.hip.h -- zab_ddt_fast[_nw2|_nw4|_nw8]: the @sample loop of Spatialization/DDT with the one-poles moved in FRONT
// of the taps (reference: plugins/Spatialization/DDT/src/DDT.jsfx:440-536, run per frame by jsfx_process_block,
// dsp_jsfx_aot.py:5713-5905).
//
// The script computes, per frame n and channel c,   eZ_c[n] = F_e( sum_{early taps i} g_ci * M[n - d_ci] )[n]   (and the same
// with the late taps and the late pole), F_a being the one-pole y[n] = (1-a) x[n] + a y[n-1] and M = 0.5 (L + R). Within a
// launch gains, delays and poles are constants, so filter and taps commute:
//
//     eZ_c[n] = sum_i g_ci * Me[n - d_ci]  +  a^(n+1) * K_c ,      Me = F_e(M) started at rest somewhere before the oldest
//     K_c     = eZ_c(state at launch start) - sum_i g_ci * Me[-1 - d_ci]                      frame a tap can reach,
//
// (n counted from the launch's first frame). K_c carries everything the incoming state knows that the new taps do not: it
// is 0 to rounding while nothing moves and decays with the pole after a slider change. So instead of one raw history ring,
// four tap sums, two LDS transposes and six scans (ddt_fast.hip.h, kept as zab_ddt_wide for delays too long for this
// layout), a chunk costs FOUR scans (Me, Ml, and the direct path's two) on data that is still in HBM order, two filtered
// rings in LDS, and one tap phase whose early and late sums share an accumulator; the mix needs one LDS round trip of the
// dry + direct part. Frames before the launch (the f64 L/R rings of mem[]) are filtered first, so the rings hold Me / Ml
// for every frame a tap can reach. The meter one-poles are fed only over the launch's last DDT_METER_FRAMES frames.
//
// Layouts: filter phase lane = KF consecutive frames (float4 per channel from HBM); tap phase lane = frames
// {l, 64 + l, ...} (conflict-free ds_read_b64 for every delay; stores are 256 B per wave instruction). Rings are plain
// (not doubled) with W a multiple of 8 and a copy of their first chunk behind the end; a tap's wrap is scalar arithmetic.
// Registers: the chunk loop is split by what a chunk needs (history / first / plain / metered), launch constants that only
// one of them uses live in LDS (D2Uni), and the last frame's @sample temporaries are rebuilt after the loop from six
// stashed values: 194 VGPRs, no scratch (two waves per SIMD; three were measured and did not pay).
// Differences from the serial order are re-association of linear terms (~1e-16 relative): tests/test_ddt_gpu.py.
#pragma once

#include "ddt_fast.hip.h"

#define D2_KF 4
#define D2_CH (64 * D2_KF)

typedef float d2_f4 __attribute__((ext_vector_type(4)));   // a float4 the register allocator keeps as ONE 128-bit tuple

struct D2Uni {               // launch constants kept in LDS (read where used)
  double dgc, dgm;           // dIn = dgc * x + dgm * (L + R)
  double pdx, pdz, tapc;     // out = clamp(tapc * (eZ + lZ) + pdx * x + pdz * dirZ)   (monitor modes folded in)
  double kE[2], kL[2];       // K of the header, per filter and channel
  double cwMd[D2_KF], cwM[D2_KF], cwC[D2_KF], dMi, dCi;   // meter weights
  double fin[8];             // last frame: eZL eZR lZL lZR spl0 spl1 dirZL dirZR
  double qpow[3][64];        // (a^KF)^j per pole: scan / carry weights by lane
};

struct D2Ctx {               // what the chunk loop keeps in scalar registers
  const float *in0, *in1;
  float *out0, *out1;
  double* Mem;
  double *ringE, *ringL;     // LDS [W + D2_CH] each
  const DdtPole* P;          // LDS [3]: direct, early, late
  D2Uni* U;                  // LDS
  double* PT;                // LDS, this wave's [2][D2_CH]: dry + direct part of the mix, filter layout -> tap layout
  int64_t frames, wofs0, rL, rR;
  int bufmask, W, nE, nT;
  bool vec_ok;
};

__device__ __forceinline__ double d2_ld_l2(const double* p) {     // reads what other waves of the launch stored (skips the CU's L1)
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(double, v);
}

// ---- filter phase, part 1: the chunk's inputs, local responses, wave scans -----------------------------------------------
// HIST: every frame of the chunk lies before the launch (inputs from the mem[] rings, no direct path).
// PARTIAL: the chunk straddles the launch's first frame.
template <bool PARTIAL, bool HIST>
__device__ __forceinline__ void d2_filter_local(const D2Ctx& C, int lane, int64_t f0, const d2_f4& p0, const d2_f4& p1, const double (&carry)[4],
                                                bool head, double (&x0)[D2_KF], double (&x1)[D2_KF], double (&y)[4][D2_KF], double (&G)[4]) {
  const int64_t t0 = f0 + D2_KF * lane;
  if (HIST || PARTIAL) {
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) {
      const int64_t t = t0 + k;
      if (HIST || t < 0) {
        const int64_t ri = (C.wofs0 + t) & C.bufmask;
        x0[k] = C.Mem[C.rL + ri]; x1[k] = C.Mem[C.rR + ri];
      } else {
        x0[k] = (double)C.in0[t]; x1[k] = (double)C.in1[t];
      }
    }
  } else if (C.vec_ok) {
    x0[0] = (double)p0.x; x0[1] = (double)p0.y; x0[2] = (double)p0.z; x0[3] = (double)p0.w;
    x1[0] = (double)p1.x; x1[1] = (double)p1.y; x1[2] = (double)p1.z; x1[3] = (double)p1.w;
  } else {
    float u0[D2_KF], u1[D2_KF];
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) { u0[k] = C.in0[t0 + k]; u1[k] = C.in1[t0 + k]; }
    // (waited for inside this branch: a wait at the merge with the prefetched path would be vmcnt(0) for that path too)
    asm volatile("" : "+v"(u0[0]), "+v"(u0[1]), "+v"(u0[2]), "+v"(u0[3]), "+v"(u1[0]), "+v"(u1[1]), "+v"(u1[2]), "+v"(u1[3]));
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) { x0[k] = (double)u0[k]; x1[k] = (double)u1[k]; }
  }
  if (!HIST && f0 + D2_CH > C.frames - DDT_RING) {         // :441-442, only slots that survive the launch
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) {
      const int64_t t = t0 + k;
      if (t >= 0 && t >= C.frames - DDT_RING) {
        const int64_t ri = (C.wofs0 + t) & C.bufmask;
        C.Mem[C.rL + ri] = x0[k];
        C.Mem[C.rR + ri] = x1[k];
      }
    }
  }
  int first_lane = 0, first_k = 0;
  if (PARTIAL) {
    const int firstv = (int)(-f0);
    first_lane = firstv / D2_KF;
    first_k = firstv % D2_KF;
  }
  const double dgc = C.U->dgc, dgm = C.U->dgm;
#pragma unroll
  for (int k = 0; k < D2_KF; ++k) {
    const double S = x0[k] + x1[k];
    const double M = 0.5 * S;                              // mono (:445) == ring value 0.5*(L+R) (:467)
    y[2][k] = M; y[3][k] = M;
    if (!HIST) {                                           // dIn = directGain * (x*(1-col) + mono*col) (:447-451), products folded
      const double mc = S * dgm;
      y[0][k] = __builtin_fma(x0[k], dgc, mc);
      y[1][k] = __builtin_fma(x1[k], dgc, mc);
    }
  }
  // scan weights of the lane: q^((l&15)+1) and q^(l-31) (q = a^KF), out of the pole's power table
  const int i1 = (lane & 15) + 1, i2 = lane >= 32 ? lane - 31 : 0;
  if (!HIST) {
    const double c1 = C.U->qpow[0][i1], c2 = C.U->qpow[0][i2];
    G[0] = ddt_pole_local<PARTIAL>(C.P[0], c1, c2, y[0], carry[0], head, lane, first_lane, first_k);
    G[1] = ddt_pole_local<PARTIAL>(C.P[0], c1, c2, y[1], carry[1], head, lane, first_lane, first_k);
  } else {
    G[0] = G[1] = 0.0;
  }
  G[2] = ddt_pole_local<false>(C.P[1], C.U->qpow[1][i1], C.U->qpow[1][i2], y[2], carry[2], head, lane, 0, 0);
  G[3] = ddt_pole_local<false>(C.P[2], C.U->qpow[2][i1], C.U->qpow[2][i2], y[3], carry[3], head, lane, 0, 0);
}

// ---- filter phase, part 2: apply the state carried into the chunk, publish rings and the mix's dry + direct part ----------
template <bool HIST, bool METERS>
__device__ __forceinline__ void d2_filter_publish(const D2Ctx& C, int lane, int pos, const double (&cw)[4], bool chained,
                                                  const double (&x0)[D2_KF], const double (&x1)[D2_KF], double (&y)[4][D2_KF],
                                                  const double (&G)[4], double& accD, bool want_last) {
#pragma unroll
  for (int s = HIST ? 2 : 0; s < 4; ++s) {
    const int p = s < 2 ? 0 : s - 1;
    double cin = ddt_dpp<DDT_WAVE_SHR1, 0xF>(G[s]);        // y at the end of the previous lane (0 for lane 0)
    if (chained) cin = __builtin_fma(C.U->qpow[p][lane], cw[s], cin);   // + a^(KF*lane) * state at the chunk's start
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) y[s][k] = __builtin_fma(C.P[p].ap[k], cin, y[s][k]);
  }
  int slot = pos + D2_KF * lane;
  slot = slot >= C.W ? slot - C.W : slot;
  {
    double2* pe = reinterpret_cast<double2*>(C.ringE + slot);
    double2* pl = reinterpret_cast<double2*>(C.ringL + slot);
    pe[0] = make_double2(y[2][0], y[2][1]); pe[1] = make_double2(y[2][2], y[2][3]);
    pl[0] = make_double2(y[3][0], y[3][1]); pl[1] = make_double2(y[3][2], y[3][3]);
    if (slot < D2_CH) {                                    // copy of the ring's first chunk behind its end (strided reads)
      pe = reinterpret_cast<double2*>(C.ringE + C.W + slot);
      pl = reinterpret_cast<double2*>(C.ringL + C.W + slot);
      pe[0] = make_double2(y[2][0], y[2][1]); pe[1] = make_double2(y[2][2], y[2][3]);
      pl[0] = make_double2(y[3][0], y[3][1]); pl[1] = make_double2(y[3][2], y[3][3]);
    }
  }
  if (!HIST) {
    const double pdx = C.U->pdx, pdz = C.U->pdz;
    double pL[D2_KF], pR[D2_KF];
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) {
      pL[k] = __builtin_fma(pdx, x0[k], pdz * y[0][k]);
      pR[k] = __builtin_fma(pdx, x1[k], pdz * y[1][k]);
    }
    double2* tl = reinterpret_cast<double2*>(C.PT + D2_KF * lane);
    double2* tr = reinterpret_cast<double2*>(C.PT + D2_CH + D2_KF * lane);
    tl[0] = make_double2(pL[0], pL[1]); tl[1] = make_double2(pL[2], pL[3]);
    tr[0] = make_double2(pR[0], pR[1]); tr[1] = make_double2(pR[2], pR[3]);
    if (METERS) {                                          // s_dir (:510-518): frames before the launch hold y = 0
      double z = 0.0;
#pragma unroll
      for (int k = 0; k < D2_KF; ++k) z = __builtin_fma(C.U->cwMd[k], fabs(y[0][k]) + fabs(y[1][k]), z);
      accD = __builtin_fma(accD, C.U->dMi, z);            // (the lane's own weight is applied once, after the loop)
      if (want_last && lane == 63) { C.U->fin[6] = y[0][D2_KF - 1]; C.U->fin[7] = y[1][D2_KF - 1]; }
    }
  }
}

// ---- tap phase ---------------------------------------------------------------------------------------------------------
// Lane j keeps staged tap j (early taps first, then late, each in source order): its gains and, per channel, the LDS byte
// address of (chunk's first frame - delay) in ITS ring, advanced once per iteration for all taps at once (one vector add and
// wrap). A tap then costs two v_readlane for the addresses, four for the gains, one vector add per channel for the lane
// offset, eight ds_read_b64 and eight FMAs -- and no scalar arithmetic: at two waves per SIMD every instruction of a wave,
// scalar ones included, takes one of its issue slots.
struct D2TapRegs { int aL, aR, lim, dpack; double gL, gR; };
struct D2TapS { int aL, aR; double gL, gR; };              // one tap, wave-uniform
__device__ __forceinline__ D2TapS d2_tap_get(const D2TapRegs& R, int i) {
  D2TapS t;
  t.aL = __builtin_amdgcn_readlane(R.aL, i);
  t.aR = __builtin_amdgcn_readlane(R.aR, i);
  t.gL = ddt_readlane(R.gL, i);
  t.gR = ddt_readlane(R.gR, i);
  return t;
}
typedef const volatile double __attribute__((address_space(3))) * d2_lds_cvd;
// eight ds_read_b64 (volatile: the compiler would pair them into ds_read2st64_b64, half the LDS rate per byte)
__device__ __forceinline__ void d2_tap_fetch(int lane8, const D2TapS& tp, double (&l)[D2_KF], double (&r)[D2_KF]) {
  const unsigned pl = (unsigned)(tp.aL + lane8), pr = (unsigned)(tp.aR + lane8);
#pragma unroll
  for (int k = 0; k < D2_KF; ++k) {
    l[k] = *(d2_lds_cvd)(uintptr_t)(pl + 512u * k);
    r[k] = *(d2_lds_cvd)(uintptr_t)(pr + 512u * k);
  }
}
// The empty asm takes all eight values at once, so the compiler waits for them with ONE s_waitcnt (they were issued back to
// back and return in order) instead of seven interleaved with the FMAs -- each would be an issue slot of the wave.
__device__ __forceinline__ void d2_tap_acc(const D2TapS& tp, double (&l)[D2_KF], double (&r)[D2_KF], double (&acc)[2][D2_KF]) {
  asm volatile("" : "+v"(l[0]), "+v"(r[0]), "+v"(l[1]), "+v"(r[1]), "+v"(l[2]), "+v"(r[2]), "+v"(l[3]), "+v"(r[3]));
#pragma unroll
  for (int k = 0; k < D2_KF; ++k) {
    acc[0][k] = __builtin_fma(tp.gL, l[k], acc[0][k]);
    acc[1][k] = __builtin_fma(tp.gR, r[k], acc[1][k]);
  }
}
// taps [i0, i1) of the staged list
// MW = 2: the reads of a tap are issued one tap ahead of their FMAs (two sets of tap values: 64 registers).
// MW = 3 (three waves per SIMD, 168 registers): ONE set of tap values; what a tap waits for its reads is filled with the next
// tap's address arithmetic (two v_readlane, two adds) and its own gains' four v_readlane, the rest is the other two waves'.
template <int MW>
__device__ __forceinline__ void d2_tap_run(const D2TapRegs& R, int lane8, int i0, int i1, double (&acc)[2][D2_KF]) {
  if (i0 >= i1) return;
  if constexpr (MW >= 3) {
    const int last = i1 - 1;
    unsigned pl = (unsigned)(__builtin_amdgcn_readlane(R.aL, i0) + lane8), pr = (unsigned)(__builtin_amdgcn_readlane(R.aR, i0) + lane8);
    for (int i = i0; i < i1; ++i) {
      double l[D2_KF], r[D2_KF];
#pragma unroll
      for (int k = 0; k < D2_KF; ++k) {
        l[k] = *(d2_lds_cvd)(uintptr_t)(pl + 512u * k);
        r[k] = *(d2_lds_cvd)(uintptr_t)(pr + 512u * k);
      }
      D2TapS tp;
      tp.gL = ddt_readlane(R.gL, i);
      tp.gR = ddt_readlane(R.gR, i);
      const int nx = i < last ? i + 1 : last;
      pl = (unsigned)(__builtin_amdgcn_readlane(R.aL, nx) + lane8);
      pr = (unsigned)(__builtin_amdgcn_readlane(R.aR, nx) + lane8);
      __builtin_amdgcn_sched_barrier(0);
      d2_tap_acc(tp, l, r, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  const int last = i1 - 1;
  double al[D2_KF], ar[D2_KF], bl[D2_KF], br[D2_KF];
  D2TapS ta = d2_tap_get(R, i0), tb;
  d2_tap_fetch(lane8, ta, al, ar);
  int i = i0;
  for (; i + 1 < i1; i += 2) {                             // pairs; the fetch past the end re-reads the last tap
    // (sched_barrier: keep "issue tap i+1, then the FMAs of tap i" -- the scheduler would group both fetches and expose
    //  a whole LDS latency per tap)
    tb = d2_tap_get(R, i + 1);
    d2_tap_fetch(lane8, tb, bl, br);
    __builtin_amdgcn_sched_barrier(0);
    d2_tap_acc(ta, al, ar, acc);
    __builtin_amdgcn_sched_barrier(0);
    ta = d2_tap_get(R, i + 2 < last ? i + 2 : last);
    d2_tap_fetch(lane8, ta, al, ar);
    __builtin_amdgcn_sched_barrier(0);
    d2_tap_acc(tb, bl, br, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (i < i1) d2_tap_acc(ta, al, ar, acc);
}

// lane = frames {f0 + l, f0 + 64 + l, ...}. METERS: early and late sums kept apart, meters fed (:510-536).
// corr: the a^(n+1) K terms still matter for this chunk (wave-uniform).
template <bool PARTIAL, bool METERS, int MW>
__device__ __forceinline__ void d2_tap_phase(const D2Ctx& C, const D2TapRegs& R, int lane, int64_t f0, bool corr,
                                             double (&accM)[6], double& accC, bool want_last,
                                             float (&o0)[D2_KF], float (&o1)[D2_KF] DDT_STAMP_ARGS) {
  const int lane8 = 8 * lane;
  double sA[2][D2_KF], sB[2][D2_KF];                       // METERS: sA early, sB late; else sA holds both
#pragma unroll
  for (int k = 0; k < D2_KF; ++k) sA[0][k] = sA[1][k] = sB[0][k] = sB[1][k] = 0.0;
  if (corr) {                                              // the first chunk(s) after a change of state or taps
    const double kE0 = C.U->kE[0], kE1 = C.U->kE[1], kL0 = C.U->kL[0], kL1 = C.U->kL[1];
#pragma unroll
    for (int k = 0; k < D2_KF; ++k) {
      const int64_t e = f0 + 64 * k + lane + 1;
      const double pe = e > 0 ? ddt_ipow(C.P[1].a, e) : 0.0, pl = e > 0 ? ddt_ipow(C.P[2].a, e) : 0.0;
      if (METERS) {
        sA[0][k] = kE0 * pe; sA[1][k] = kE1 * pe;
        sB[0][k] = kL0 * pl; sB[1][k] = kL1 * pl;
      } else {
        sA[0][k] = __builtin_fma(kE0, pe, kL0 * pl);
        sA[1][k] = __builtin_fma(kE1, pe, kL1 * pl);
      }
    }
  }
  if (METERS) {
    d2_tap_run<MW>(R, lane8, 0, C.nE, sA);
    d2_tap_run<MW>(R, lane8, C.nE, C.nT, sB);
  } else {
    d2_tap_run<MW>(R, lane8, 0, C.nT, sA);                     // early and late taps into one sum
  }
  DDT_STAMP(5)

  const double tapc = C.U->tapc;
  double zM[6] = {0, 0, 0, 0, 0, 0}, zC = 0.0;
#pragma unroll
  for (int k = 0; k < D2_KF; ++k) {
    const double dL = METERS ? sA[0][k] + sB[0][k] : sA[0][k];
    const double dR = METERS ? sA[1][k] + sB[1][k] : sA[1][k];
    const double s0 = __builtin_fma(tapc, dL, C.PT[64 * k + lane]);
    const double s1 = __builtin_fma(tapc, dR, C.PT[D2_CH + 64 * k + lane]);
    // clamp to +-8 (:504-505) after the rounding to float: the same samples (8 is a float, rounding is monotonic, NaN stays)
    float q0 = (float)s0, q1 = (float)s1;
    q0 = q0 > 8.0f ? 8.0f : (q0 < -8.0f ? -8.0f : q0);
    q1 = q1 > 8.0f ? 8.0f : (q1 < -8.0f ? -8.0f : q1);
    o0[k] = q0; o1[k] = q1;                                // stored by the caller
    if (!METERS) continue;
    const bool valid = !PARTIAL || f0 + 64 * k + lane >= 0;
    const double eZL = sA[0][k], eZR = sA[1][k], lZL = sB[0][k], lZR = sB[1][k];
    const double s_ear2 = valid ? fabs(eZL) + fabs(eZR) : 0.0;
    const double s_lat2 = valid ? fabs(lZL) + fabs(lZR) : 0.0;
    const double adL = valid ? fabs(dL) : 0.0, adR = valid ? fabs(dR) : 0.0;
    // c = dL*dR / max(1e-7, |dL||dR| + 1e-7) (:534): the divisor is within [1e-7, ~1e2], so a hardware reciprocal
    // refined by two Newton steps (~1e-16 relative) replaces the full IEEE division sequence; c only feeds a meter.
    const double den = __builtin_fmax(0.0000001, __builtin_fma(adL, adR, 0.0000001));
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    const double cc = valid ? (dL * dR) * rc : 0.0;
    const double cM = C.U->cwM[k], cC = C.U->cwC[k];
    zM[1] = __builtin_fma(cM, s_ear2, zM[1]);
    zM[2] = __builtin_fma(cM, s_lat2, zM[2]);
    zM[4] = __builtin_fma(cM, adL, zM[4]);
    zM[5] = __builtin_fma(cM, adR, zM[5]);
    zC = __builtin_fma(cC, __builtin_fmin(__builtin_fmax(cc, -1.0), 1.0), zC);
    if (want_last && k == D2_KF - 1 && lane == 63) {       // the launch's last frame: what the epilogue rebuilds vars[] from
      double* fin = C.U->fin;
      fin[0] = eZL; fin[1] = eZR; fin[2] = lZL; fin[3] = lZR;
      fin[4] = s0 > 8.0 ? 8.0 : (s0 < -8.0 ? -8.0 : s0);
      fin[5] = s1 > 8.0 ? 8.0 : (s1 < -8.0 ? -8.0 : s1);
    }
  }
  if (METERS) {
    const double dMi = C.U->dMi, dCi = C.U->dCi;
#pragma unroll
    for (int q = 1; q < 6; ++q) accM[q] = __builtin_fma(accM[q], dMi, zM[q]);      // (lane weights: after the loop)
    accC = __builtin_fma(accC, dCi, zC);
  }
}

template <int NW, int MW>
__device__ __forceinline__ void d2_body(const ZabBatch& b, const ZabAudio& a, int W, int nh) {
  extern __shared__ __attribute__((aligned(16))) double ddt_lds[];
  double* ringE = ddt_lds;                                 // [W + CH]
  double* ringL = ringE + W + D2_CH;                       // [W + CH]
  double* PTall = ringL + W + D2_CH;                       // [NW][2][CH]
  double* gend = PTall + NW * 2 * D2_CH;                   // [NW][4] chunk-end responses
  D2Uni* U = (D2Uni*)(gend + NW * 4);
  DdtPole* P = (DdtPole*)(U + 1);                          // [3]
  int* scratch = (int*)(P + 3);                            // [4]
  double* mred = PTall;                                    // [NW][7] meter partials, after the last chunk (aliases PTall)
  DdtTap* taps = (DdtTap*)PTall;                           // [DDT_MAXTAPS] staging only (aliases PTall)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: ring positions stay scalar)
  const int inst = blockIdx.x;

  double* V = b.vars + (int64_t)inst * b.var_si;           // instance-major (checked by za_fast_applies)
  double* Mem = b.mem + (int64_t)inst * b.mem_si;
  const double* SL = b.sliders + (int64_t)inst * b.sl_si;
  const int64_t frames = a.frames;
  if (frames <= 0) return;

  D2Ctx C;
  C.in0 = a.in + (int64_t)inst * 2 * a.frame_stride;
  C.in1 = C.in0 + a.frame_stride;
  C.out0 = a.out + (int64_t)inst * 2 * a.frame_stride;
  C.out1 = C.out0 + a.frame_stride;
  C.Mem = Mem; C.ringE = ringE; C.ringL = ringL; C.PT = PTall + wave * 2 * D2_CH; C.U = U; C.P = P; C.frames = frames;
  C.W = W;

  // ---- per-launch scalars (workgroup-uniform) --------------------------------------------------------------------------
  const double mbase = V[ZA_VAR_m];
  C.rL = za_addr(mbase, V[ZA_VAR_bL]); C.rR = za_addr(mbase, V[ZA_VAR_bR]);
  const int64_t tDL = za_addr(mbase, V[ZA_VAR_bDL]), tDR = za_addr(mbase, V[ZA_VAR_bDR]);
  const int64_t tGL = za_addr(mbase, V[ZA_VAR_bGL]), tGR = za_addr(mbase, V[ZA_VAR_bGR]), tD0 = za_addr(mbase, V[ZA_VAR_bD0]);
  C.bufmask = za_i32(V[ZA_VAR_BUF_MASK]);
  int tapN = (int)za_loopcount(V[ZA_VAR_tapN]);
  if (tapN > DDT_MAXTAPS) tapN = DDT_MAXTAPS;
  C.wofs0 = za_f2i64(V[ZA_VAR_wofs]);
  const double aM = 0.9985, aC = 0.9990;                   // meter poles (:128-131)

  if (tid == 0) {
    // distN = smooth01(slider1/100); col = distN^0.8   (:444-446; clamp/smooth01 :62-64)
    const double tt = ddt_clamp(SL[0] / 100.0, 0.0, 1.0);
    const double col = pow((tt * tt) * (3.0 - 2.0 * tt), 0.8), directGain = V[ZA_VAR_directGain];
    U->dgc = directGain * (1.0 - col); U->dgm = directGain * col * 0.5;
    // (dryp*x + wetp*o) * out_gain (:500-503), o by monitor mode (:492-498): y = dirZ + d | dirZ | d | x
    const int mon = za_i32(SL[7]);
    const double mixd = V[ZA_VAR_dryp] * V[ZA_VAR_out_gain], mixw = V[ZA_VAR_wetp] * V[ZA_VAR_out_gain];
    U->pdx = mon == 3 ? mixd + mixw : mixd;
    U->pdz = (mon == 2 || mon == 3) ? 0.0 : mixw;
    U->tapc = (mon == 1 || mon == 3) ? 0.0 : mixw;
    U->kE[0] = U->kE[1] = U->kL[0] = U->kL[1] = 0.0;
    // meters: m = (1-aM)*val + aM*m  (:128-131,518-536). A chunk's frames enter with weight (1-a) a^(CH-1-j), j their index in
    // the chunk: j = KF*lane + k in the filter phase, 64*k + lane in the tap phase.
    for (int k = 0; k < D2_KF; ++k) {
      U->cwMd[k] = (1.0 - aM) * ddt_ipow(aM, D2_KF - 1 - k);
      U->cwM[k] = (1.0 - aM) * ddt_ipow(aM, 64 * (D2_KF - 1 - k));
      U->cwC[k] = (1.0 - aC) * ddt_ipow(aC, 64 * (D2_KF - 1 - k));
    }
    U->dMi = ddt_ipow(aM, D2_CH * NW); U->dCi = ddt_ipow(aC, D2_CH * NW);   // a wave's chunks are NW apart
  }
  const double poles[3] = {V[ZA_VAR_a_dir], V[ZA_VAR_a_early], V[ZA_VAR_a_late]};
  if (tid < 3) {
    DdtPole pl;
    const double pole = poles[tid];
    pl.a = pole; pl.c1 = 1.0 - pole;
    pl.ap[0] = pole; pl.ap[1] = pole * pole; pl.ap[2] = pl.ap[1] * pole; pl.ap[3] = pl.ap[1] * pl.ap[1];
    pl.sp[0] = pl.ap[3];
    for (int j = 1; j < 4; ++j) pl.sp[j] = pl.sp[j - 1] * pl.sp[j - 1];
    pl.a256 = ddt_ipow(pole, D2_CH);
    P[tid] = pl;
  }
  if (wave == 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p) U->qpow[p][lane] = ddt_ipow((poles[p] * poles[p]) * (poles[p] * poles[p]), lane);
  }
  // running states of the four filters: the direct path's are the script's; Me / Ml start at rest before the history
  double carry[4] = {V[ZA_VAR_dirZL], V[ZA_VAR_dirZR], 0.0, 0.0};

  // ---- stage tap lists (wave 0): early taps (baseD < splitSamp) then late taps, each in source order -------------------
  if (wave == 0) {
    const double splitSamp = V[ZA_VAR_splitSamp];
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
    const int nE = __popcll(emask);
    if (lane < tapN) taps[early ? __popcll(emask & below) : nE + __popcll(lmask & below)] = mine;
    if (lane == tapN - 1) { scratch[0] = mine.dL8 >> 3; scratch[1] = mine.dR8 >> 3; }   // source-order last tap
    if (lane == 0) scratch[2] = nE;
  }
  __syncthreads();
  C.nE = scratch[2];
  C.nT = tapN;
  const int W8 = 8 * W;
  const int pos0 = (int)(((int64_t)D2_CH * wave) % W);     // ring slot of this wave's first chunk (chunk c starts at (c*CH) mod W)
  const int pos_step = (NW * D2_CH) % W;
  D2TapRegs R;                                             // lane j keeps staged tap j for the whole launch
  {
    const DdtTap t = taps[lane];                           // entries >= tapN are never broadcast
    R.dpack = (t.dL8 >> 3) | ((t.dR8 >> 3) << 16);         // delays < 16384 (checked by the plan kernel)
    R.gL = t.gL; R.gR = t.gR;
    const int base = (int)(unsigned)(uintptr_t)(lane < C.nE ? ringE : ringL);   // LDS byte address of the tap's ring
    int sl = 8 * pos0 - t.dL8, sr = 8 * pos0 - t.dR8;      // (delays <= Dmax < W)
    sl += sl < 0 ? W8 : 0; sr += sr < 0 ? W8 : 0;
    R.aL = base + sl; R.aR = base + sr; R.lim = base + W8;
  }
  __syncthreads();                                         // (taps aliases the PT areas)

  C.vec_ok = ((frames & 3) == 0) && ((a.frame_stride & 3) == 0) && ((((uintptr_t)C.in0) | ((uintptr_t)C.out0)) & 15) == 0;

  // Chunks are end-aligned; nh chunks of history (a multiple of NW, so an iteration is all history or all launch) precede them.
  const int64_t nchunks = (frames + D2_CH - 1) / D2_CH;
  const int64_t f_first = frames - D2_CH * nchunks;        // <= 0
  const int64_t ntot = nh + nchunks;
  const int64_t niter = (ntot + NW - 1) / NW;
  const int64_t it0 = nh / NW;                             // the iteration that holds the launch's first frame
  // metered iterations: from the one that holds frame (frames - DDT_METER_FRAMES) on
  int64_t it_m = it0;
  if (frames > DDT_METER_FRAMES) {
    const int64_t c_t = nh + (frames - DDT_METER_FRAMES - f_first) / D2_CH;
    it_m = c_t / NW;
    if (it_m < it0) it_m = it0;
  }
  int pos = pos0;                                          // ring slot of this wave's current chunk
  int64_t corr_until = 0;                                  // first frame from which every a^(n+1) K term is below 1e-60
  int64_t my_last_chunk = -1;
  DDT_STAMP_DECL

  // The HBM read of a wave's next chunk is issued a whole iteration ahead of its use.
  d2_f4 pf0 = {0.f, 0.f, 0.f, 0.f}, pf1 = pf0;
  // The loads are inline asm writing straight into the two 128-bit tuples that live across the loop (unconditional per
  // lane: addresses clamped into the buffer, what a lane does not need it never looks at). Left to the compiler a load
  // under a lane condition -- or one whose result it wants in other registers -- goes through a temporary and is copied,
  // behind a full s_waitcnt right after the load: a whole HBM round trip per chunk (measured: 14 % of the kernel). The
  // matching wait is D2_PF_WAIT (the compiler does not count an asm's memory operations; its own waits only get stricter).
  auto prefetch = [&](const int64_t c) __attribute__((always_inline)) {
    if (C.vec_ok) {                                        // wave-uniform
      int64_t t0 = f_first + D2_CH * (c - nh) + D2_KF * lane;
      t0 = t0 < 0 ? 0 : (t0 > frames - D2_KF ? frames - D2_KF : t0);
      const float* a0 = C.in0 + t0;
      const float* a1 = C.in1 + t0;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pf0) : "v"(a0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pf1) : "v"(a1) : "memory");
    }
  };
#define D2_PF_WAIT asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf0), "+v"(pf1) : : "memory")
  if (nh == 0) prefetch(wave);
  D2_PF_WAIT;

  // per-lane meter state and weights: only the metered iterations touch them (defined just before those run)
  double accM[6] = {0, 0, 0, 0, 0, 0}, accC = 0.0;

  auto iteration = [&](auto part_c, auto hist_c, auto met_c, const int64_t it) __attribute__((always_inline)) {
    constexpr bool PART = decltype(part_c)::value, HIST = decltype(hist_c)::value, MET = decltype(met_c)::value;
    const int64_t c = it * NW + wave;
    const bool active = c < ntot;
    const int64_t f0 = f_first + D2_CH * (c - nh);
    const bool want_last = MET && active && c == ntot - 1;
    const bool head = (wave == 0);                         // wave 0 owns the head of this iteration's carry chain
    const bool part = PART && wave == 0;                   // only the launch's first chunk can straddle frame 0
    double x0[D2_KF], x1[D2_KF], y[4][D2_KF], G[4] = {0, 0, 0, 0};
    if (active) {
      if (part) d2_filter_local<true, false>(C, lane, f0, pf0, pf1, carry, head, x0, x1, y, G);
      else d2_filter_local<false, HIST>(C, lane, f0, pf0, pf1, carry, head, x0, x1, y, G);
    }
    DDT_STAMP(0)
    prefetch(c + NW);                                      // this wave's next chunk: in flight across the rest of the iteration
    DDT_STAMP(4)
    // carry chain: state entering wave w's chunk = a^CH * (state entering w-1) + response of w-1; wave 0 injected `carry`
    double cw[4] = {0, 0, 0, 0};
    constexpr int S0 = HIST ? 2 : 0;                       // the direct path's state is not touched before the launch's first frame
    if (NW == 1) {
#pragma unroll
      for (int s = S0; s < 4; ++s) carry[s] = ddt_readlane(G[s], 63);
    } else {
      if (active && lane == 63) {
#pragma unroll
        for (int s = 0; s < 4; ++s) gend[wave * 4 + s] = G[s];
      }
      ddt_barrier();                                       // chunk-end responses published; last iteration's tap phases done
      double run[4] = {0, 0, 0, 0};
      const int nact = (int)((ntot - it * NW) < NW ? (ntot - it * NW) : NW);
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        if (u < nact) {
          if (u == wave) {
#pragma unroll
            for (int s = 0; s < 4; ++s) cw[s] = run[s];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) run[s] = __builtin_fma(P[s < 2 ? 0 : s - 1].a256, run[s], gend[u * 4 + s]);
        }
      }
#pragma unroll
      for (int s = S0; s < 4; ++s) carry[s] = run[s];      // state after this iteration's last chunk (same in every wave)
    }
    DDT_STAMP(1)
    if (active) d2_filter_publish<HIST, MET>(C, lane, pos, cw, wave != 0, x0, x1, y, G, accM[0], want_last);
    DDT_STAMP(2)
    if (NW > 1) ddt_barrier();                             // rings hold every frame of this iteration
    else {                                                 // (one wave: its LDS accesses execute in order; the compiler must keep them so)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    DDT_STAMP(3)
    float o0[D2_KF], o1[D2_KF];
    if (!HIST) {
      if (it == it0) {
        // K of the header, once per launch: lane j holds staged tap j; the rings hold every frame from -1 - Dmax on
        double vL = 0.0, vR = 0.0;
        if (lane < C.nT) {
          const double* ring = lane < C.nE ? ringE : ringL;
          const int pos_first = (int)(((int64_t)D2_CH * nh) % W);          // slot of frame f_first
          const int dl = R.dpack & 0xffff, dr = (int)((unsigned)R.dpack >> 16);
          int sl = (int)((pos_first + (-1 - dl - f_first)) % W), sr = (int)((pos_first + (-1 - dr - f_first)) % W);
          sl += sl < 0 ? W : 0; sr += sr < 0 ? W : 0;
          vL = R.gL * ring[sl]; vR = R.gR * ring[sr];
        }
        double sum[4] = {lane < C.nE ? vL : 0.0, lane < C.nE ? vR : 0.0, lane < C.nE ? 0.0 : vL, lane < C.nE ? 0.0 : vR};
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
          for (int q = 0; q < 4; ++q) sum[q] += __shfl_xor(sum[q], d, 64);
        const double k0 = V[ZA_VAR_eZL] - sum[0], k1 = V[ZA_VAR_eZR] - sum[1], k2 = V[ZA_VAR_lZL] - sum[2], k3 = V[ZA_VAR_lZR] - sum[3];
        if (tid == 0) { U->kE[0] = k0; U->kE[1] = k1; U->kL[0] = k2; U->kL[1] = k3; }
        // |K| a^n < 1e-60 from n = ln(1e-60 / |K|) / ln(a) on
        const double mE = fmax(fabs(k0), fabs(k1)), mL = fmax(fabs(k2), fabs(k3));
        double nE_ = 0.0, nL_ = 0.0;
        if (mE > 1e-60 && poles[1] > 0.0) nE_ = poles[1] < 1.0 ? ceil(log(1e-60 / mE) / log(poles[1])) : 1e18;
        if (mL > 1e-60 && poles[2] > 0.0) nL_ = poles[2] < 1.0 ? ceil(log(1e-60 / mL) / log(poles[2])) : 1e18;
        const double nn = fmax(nE_, nL_);
        corr_until = nn >= 9e17 ? (int64_t)1 << 62 : (int64_t)nn;
        if (NW > 1) ddt_barrier();                         // (K is read from LDS by every wave and lane)
        else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
      if (active) {
        const bool corr = f0 < corr_until;
        if (part) d2_tap_phase<true, MET, MW>(C, R, lane, f0, corr, accM, accC, want_last, o0, o1 DDT_STAMP_PASS);
        else d2_tap_phase<false, MET, MW>(C, R, lane, f0, corr, accM, accC, want_last, o0, o1 DDT_STAMP_PASS);
        my_last_chunk = c;
      }
    }
    // The next chunk's audio (issued at the top of this iteration) is waited for HERE, on every path, before this chunk's
    // stores join the queue: vmcnt retires in order, and where paths merge at the top of the loop the compiler can only
    // wait for everything (vmcnt(0)) -- which would include these stores' trip to HBM, every chunk.
    D2_PF_WAIT;
    if (!HIST && active) {
      float* q0 = C.out0 + f0;
      float* q1 = C.out1 + f0;
#pragma unroll
      for (int k = 0; k < D2_KF; ++k)
        if (!PART || f0 + 64 * k + lane >= 0) { q0[64 * k + lane] = o0[k]; q1[64 * k + lane] = o1[k]; }
    }
    DDT_STAMP(6)
    pos += pos_step; pos = pos >= W ? pos - W : pos;
    R.aL += 8 * pos_step; R.aL -= R.aL >= R.lim ? W8 : 0;  // every tap's read address moves with the chunk
    R.aR += 8 * pos_step; R.aR -= R.aR >= R.lim ? W8 : 0;
  };
  {
    const std::false_type no{};
    const std::true_type yes{};
    int64_t it = 0;
    for (; it < it0; ++it) iteration(no, yes, no, it);
    if (f_first < 0) {                                     // workgroup-uniform
      if (it >= it_m) iteration(yes, no, yes, it); else iteration(yes, no, no, it);
      ++it;
    }
    for (; it < it_m; ++it) iteration(no, no, no, it);
    for (; it < niter; ++it) iteration(no, no, yes, it);
  }
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
    const int64_t behind = my_last_chunk >= 0 ? (ntot - 1 - my_last_chunk) * D2_CH : 0;
    const double fM = ddt_ipow(aM, behind), fC = ddt_ipow(aC, behind);
    // ... and each lane's frames enter with the lane's own weight, a^(distance to the chunk's end): the accumulators above are
    // kept unweighted (accM = accM * a^(CH NW) + z per chunk), the weight factors out of the sum and is applied here -- three
    // per-lane doubles fewer across the loop (under the 168-register cap they were what pushed the accumulators into scratch,
    // whose stores then stood in every chunk's vmcnt(0): 2392 -> 1912 ticks per metered wave-chunk, tools/ddt_stamps.py)
    const double wMd = ddt_ipow((aM * aM) * (aM * aM), 63 - lane);    // filter phase: lane = KF consecutive frames
    const double wM = ddt_ipow(aM, 63 - lane), wC = ddt_ipow(aC, 63 - lane);   // tap phase: lane = frames 64 apart
    red[0] = accM[0] * (wMd * fM);
#pragma unroll
    for (int q = 1; q < 6; ++q) red[q] = accM[q] * (wM * fM);
    red[6] = accC * (wC * fC);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int q = 0; q < 7; ++q) red[q] += __shfl_xor(red[q], d, 64);
  __threadfence();                                         // the mem[] rings written above are read back below
  __syncthreads();                                         // (every wave is past its last tap phase: PTall is free)
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 7; ++q) mred[wave * 7 + q] = red[q];
  }
  __syncthreads();

  // ---- state write-back: the wave that processed the launch's last chunk ------------------------------------------------
  const int last_wave = (int)((ntot - 1) % NW);
  if (wave == last_wave) {
    // the last frame's raw tap reads and sums (:459-484), in the script's order within each sum; M from the mem[] rings
    const int64_t nlast = C.wofs0 + frames - 1;
    double mL = 0.0, mR = 0.0;
    if (lane < C.nT) {
      const int dl = R.dpack & 0xffff, dr = (int)((unsigned)R.dpack >> 16);
      const int64_t il = (nlast - dl) & C.bufmask, ir = (nlast - dr) & C.bufmask;
      mL = 0.5 * (d2_ld_l2(Mem + C.rL + il) + d2_ld_l2(Mem + C.rR + il));
      mR = 0.5 * (d2_ld_l2(Mem + C.rL + ir) + d2_ld_l2(Mem + C.rR + ir));
    }
    double sEL = 0.0, sER = 0.0, sLL = 0.0, sLR = 0.0;
    for (int j = 0; j < C.nE; ++j) {
      sEL = __builtin_fma(ddt_readlane(R.gL, j), ddt_readlane(mL, j), sEL);
      sER = __builtin_fma(ddt_readlane(R.gR, j), ddt_readlane(mR, j), sER);
    }
    for (int j = C.nE; j < C.nT; ++j) {
      sLL = __builtin_fma(ddt_readlane(R.gL, j), ddt_readlane(mL, j), sLL);
      sLR = __builtin_fma(ddt_readlane(R.gR, j), ddt_readlane(mR, j), sLR);
    }
    if (lane == 63) {
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
      V[ZA_VAR_wofs] = V[ZA_VAR_wofs] + (double)frames;
      // @sample temporaries of the last frame, exactly as the script leaves them (:444-536), from what the loop stashed
      const double* fin = U->fin;
      const double eZL = fin[0], eZR = fin[1], lZL = fin[2], lZR = fin[3], dirZL = fin[6], dirZR = fin[7];
      const double xl = (double)C.in0[frames - 1], xr = (double)C.in1[frames - 1], M = 0.5 * (xl + xr);
      const double tt = ddt_clamp(SL[0] / 100.0, 0.0, 1.0), distN = (tt * tt) * (3.0 - 2.0 * tt);
      const double col = pow(distN, 0.8), one_m_col = 1.0 - col, directGain = V[ZA_VAR_directGain];
      const int mon = za_i32(SL[7]);
      V[ZA_VAR_distN] = distN; V[ZA_VAR_col] = col;
      V[ZA_VAR_mono] = M;
      V[ZA_VAR_srcL] = __builtin_fma(xl, one_m_col, M * col); V[ZA_VAR_srcR] = __builtin_fma(xr, one_m_col, M * col);
      V[ZA_VAR_dInL] = directGain * __builtin_fma(xl, one_m_col, M * col);
      V[ZA_VAR_dInR] = directGain * __builtin_fma(xr, one_m_col, M * col);
      V[ZA_VAR_dirZL] = dirZL; V[ZA_VAR_dirZR] = dirZR;
      V[ZA_VAR_eZL] = eZL; V[ZA_VAR_eZR] = eZR; V[ZA_VAR_lZL] = lZL; V[ZA_VAR_lZR] = lZR;
      V[ZA_VAR_sumEL] = sEL; V[ZA_VAR_sumER] = sER; V[ZA_VAR_sumLL] = sLL; V[ZA_VAR_sumLR] = sLR;
      const double yL = dirZL + eZL + lZL, yR = dirZR + eZR + lZR, dL = eZL + lZL, dR = eZR + lZR;
      double oL, oR;
      if (mon == 3) { oL = xl; oR = xr; }
      else if (mon == 1) { oL = dirZL; oR = dirZR; }
      else if (mon == 2) { oL = dL; oR = dR; }
      else { oL = yL; oR = yR; }
      V[ZA_VAR_yL] = yL; V[ZA_VAR_yR] = yR; V[ZA_VAR_oL] = oL; V[ZA_VAR_oR] = oR;
      const double s_dir = 0.5 * (fabs(dirZL) + fabs(dirZR)), s_ear = 0.5 * (fabs(eZL) + fabs(eZR)), s_lat = 0.5 * (fabs(lZL) + fabs(lZR));
      V[ZA_VAR_s_dir] = s_dir; V[ZA_VAR_s_ear] = s_ear; V[ZA_VAR_s_lat] = s_lat; V[ZA_VAR_s_tot] = s_dir + s_ear + s_lat;
      V[ZA_VAR_dL] = dL; V[ZA_VAR_dR] = dR;
      V[ZA_VAR_c] = (dL * dR) / __builtin_fmax(0.0000001, fabs(dL) * fabs(dR) + 0.0000001);
      double* SPL = b.spl + (int64_t)inst * b.sl_si;
      SPL[0] = fin[4]; SPL[1] = fin[5];
      V[ZA_VAR_i] = (double)tapN;
      if (tapN > 0) {
        const int dLlast = scratch[0], dRlast = scratch[1];
        const int64_t il = (nlast - dLlast) & C.bufmask, ir = (nlast - dRlast) & C.bufmask;
        V[ZA_VAR_idxL] = (double)(int32_t)il;
        V[ZA_VAR_idxR] = (double)(int32_t)ir;
        V[ZA_VAR_xL] = 0.5 * (d2_ld_l2(Mem + C.rL + il) + d2_ld_l2(Mem + C.rR + il));
        V[ZA_VAR_xR] = 0.5 * (d2_ld_l2(Mem + C.rL + ir) + d2_ld_l2(Mem + C.rR + ir));
        V[ZA_VAR_gL] = Mem[tGL + tapN - 1]; V[ZA_VAR_gR] = Mem[tGR + tapN - 1];
        V[ZA_VAR_baseD] = (double)za_i32(Mem[tD0 + tapN - 1]);
      }
      V[ZA_VAR_mon] = (double)mon;
      V[ZA_VAR_aM] = aM;
      const int64_t hi = (C.rL > C.rR ? C.rL : C.rR) + DDT_RING;
      if (b.mem_high[inst] < hi) b.mem_high[inst] = hi;
    }
  }
}

#undef D2_PF_WAIT

// _nwK: K wavefronts per instance; W = ring length (frames), nh = history chunks filtered before the launch's first chunk
#define D2_KERNEL(name, NW, MW)                                                                             \
  extern "C" __global__ void __launch_bounds__(64 * NW, MW) name(ZabBatch b, ZabAudio a, int W, int nh) {      \
    d2_body<NW, MW>(b, a, W, nh);                                                                               \
  }
D2_KERNEL(zab_ddt_fast, 1, 2)
D2_KERNEL(zab_ddt_fast_nw2, 2, 2)
D2_KERNEL(zab_ddt_fast_nw4, 4, 2)
D2_KERNEL(zab_ddt_fast_nw8, 8, 2)
// ..w3: register allocation for three waves per SIMD (168), single-buffered taps (d2_tap_run)
D2_KERNEL(zab_ddt_fast_nw2w3, 2, 3)
D2_KERNEL(zab_ddt_fast_nw4w3, 4, 3)
D2_KERNEL(zab_ddt_fast_nw8w3, 8, 3)
#undef D2_KERNEL

static size_t d2_lds_bytes(int W, int nw) {
  return (size_t)(2 * (W + D2_CH) + nw * 2 * D2_CH + nw * 4) * sizeof(double) + sizeof(D2Uni) + 3 * sizeof(DdtPole) + 16;
}
// Waves per instance: two share the rings at any batch size (measured on MI355X, 96 000 frames, NW = 1/2/4/8: N = 256 ->
// 1.30/0.80/0.50/0.41 ms, 1024 -> 1.49/1.24/1.35/1.63, 2048 -> 3.35/2.66/2.62/3.27, 4096 -> 5.17/4.61/4.92/6.40, 8192 ->
// 9.38/8.69/9.39/12.6), more while the batch has fewer than 2048 wavefronts.
static int d2_pick_nw(int n_inst) {
  int nw = 2;
  while (nw < DDT_MAXNW && (int64_t)n_inst * nw < 2048) nw <<= 1;
  if (const char* e = getenv("ZAB_DDT_NW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) nw = v; }
  return nw;
}
// Ring length and waves per instance; W = 0: the two filtered rings do not fit (or the history they need is longer than
// the mem[] rings hold) -> the wide-history kernel.
static void d2_geometry_nw(const ZabBatch* b, int64_t frames, int nw0, int& W, int& nw, int& nh) {
  const DdtPlan p = ddt_plan(b);
  W = 0; nw = 1; nh = 0;
  if (!p.ok) return;
  const size_t cap = 160 * 1024 - 512;
  for (nw = nw0; nw >= 1; nw >>= 1) {
    const int need = p.dmax + nw * D2_CH + 1;
    const int w = (need + 7) / 8 * 8;
    // history: frames [-(dmax + 1), 0) must have been filtered; the launch's first chunk reaches back to f_first
    const int64_t nchunks = (frames + D2_CH - 1) / D2_CH, f_first = frames - D2_CH * nchunks;
    int64_t h = (p.dmax + 1 + f_first + D2_CH - 1) / D2_CH;
    if (h < 0) h = 0;
    h = (h + nw - 1) / nw * nw;
    if (d2_lds_bytes(w, nw) <= cap && h * D2_CH - f_first <= DDT_RING) { W = w; nh = (int)h; return; }
  }
  nw = 1;
}
// Waves per SIMD the launch is compiled for (mw) and waves per instance. Three waves per SIMD need twelve waves per CU inside the
// LDS: at the default delays (rings of 1176 + 256 frames) that is three workgroups of FOUR waves (49 KB each), not five of two
// (33 KB each: four fit). Measured on MI355X (profiles/r04_ddt_w3.txt), 4096 instances: the 168-register kernels' PLAIN chunks
// cost 0.0291 ms per 1000 frames against 0.0347 (nw2, two waves per SIMD, double-buffered taps), their METERED chunks -- two
// tap sums, the meters' arithmetic, what the allocation spills (100 B) -- 0.0578 against 0.0492. A launch meters its last
// 57 344 frames, so the 168-register kernel takes the launches that are mostly plain: 480 000 frames 15.6 against 17.5 ms,
// 96 000 frames 4.44 against 4.17. nw2w3 (8 waves per CU either way, the taps' LDS latency exposed) loses everywhere: 19.8 ms.
// ZAB_DDT_MINW = 2 | 3 and ZAB_DDT_NW pin the choice.
#ifndef D2_W3_MIN_FRAMES
#define D2_W3_MIN_FRAMES 160000
#endif
static int d2_cu_count() {                     // compute units of the current device (cached per device ordinal)
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cached[dev]) {
    int v = 0;
    cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cached[dev];
}
// Which of the two wins is a matter of ROUNDS: the 194-register kernel keeps 4 workgroups (of two waves) per CU resident, the
// 168-register one 3 (of four waves), and a batch runs in ceil(instances / resident workgroups) rounds -- 1024 instances are one
// round of the former (4.54 ms x 480 000 frames) and two of the latter (4.97), 1536 two and two (8.76 against 6.56), 4096 four and
// six (17.5 against 15.7). A round's time goes with the workgroups per CU times the per-instance rate of plain chunks measured above.
static void d2_geometry(const ZabBatch* b, int64_t frames, int& W, int& nw, int& nh, int& mw) {
  mw = 2;
  const char* em = getenv("ZAB_DDT_MINW");
  const bool pinned_nw = getenv("ZAB_DDT_NW") != nullptr;
  const size_t cap = 160 * 1024 - 512;
  if (em ? atoi(em) != 2 : (b->n_inst >= 1024 && frames >= D2_W3_MIN_FRAMES)) {
    int W3, nw3, nh3;
    d2_geometry_nw(b, frames, pinned_nw ? d2_pick_nw(b->n_inst) : 4, W3, nw3, nh3);
    bool take = W3 != 0 && nw3 > 1 && (em || pinned_nw);
    if (W3 != 0 && nw3 == 4 && !take && 3 * d2_lds_bytes(W3, 4) <= cap) {
      int W2, nw2, nh2;
      d2_geometry_nw(b, frames, d2_pick_nw(b->n_inst), W2, nw2, nh2);
      if (W2 == 0 || nw2 < 2) take = true;
      else {
        const int64_t cus = d2_cu_count();
        int64_t wg2 = (int64_t)(cap / d2_lds_bytes(W2, nw2));
        if (wg2 > 8 / nw2) wg2 = 8 / nw2;
        if (wg2 < 1) wg2 = 1;
        const double cost2 = (double)((b->n_inst + wg2 * cus - 1) / (wg2 * cus)) * (double)wg2 * 0.0347;
        const double cost3 = (double)((b->n_inst + 3 * cus - 1) / (3 * cus)) * 3.0 * 0.0291;
        take = cost3 < cost2;
      }
    }
    if (take) { W = W3; nw = nw3; nh = nh3; mw = 3; return; }
  }
  d2_geometry_nw(b, frames, d2_pick_nw(b->n_inst), W, nw, nh);
}

// Which kernel takes a launch. zab_ddt_fast filters the history a tap can reach (Dmax frames) before the launch's first
// chunk and its metered chunks (the last 57 344 frames) cost more than its plain ones, so short launches are zab_ddt_wide's.
// Measured on MI355X at the default sliders (fast / wide, ms): 1024 instances x 9 600 frames 0.25 / 0.16, 48 000: 0.67 / 0.65,
// 96 000: 1.13 / 1.24, 192 000: 2.12 / 2.31, 480 000: 4.50 / 5.44; 4096 instances x 48 000: 2.60 / 2.35, 96 000: 4.36 / 4.31,
// 192 000: 7.58 / 8.03, 480 000: 18.2 / 19.6. ZAB_DDT_KERNEL = fast | wide pins it (tests cover both on every fixture).
#ifndef D2_MIN_FRAMES
#define D2_MIN_FRAMES 96000
#endif
static bool d2_wanted(int64_t frames) {
  if (const char* e = getenv("ZAB_DDT_KERNEL")) {
    if (!strcmp(e, "wide")) return false;
    if (!strcmp(e, "fast")) return true;
  }
  return frames >= D2_MIN_FRAMES;
}

static int32_t za_fast_applies(const ZabBatch* b, const ZabAudio* a) {
  if (!b->instance_major || b->var_se != 1 || b->mem_se != 1 || b->sl_se != 1) return 0;
  if (a->frames <= 0) return 0;
  int W, nw, nh, mw;
  if (d2_wanted(a->frames)) {
    d2_geometry(b, a->frames, W, nw, nh, mw);
    if (W != 0) return 1;
  }
  ddt_geometry(b, W, nw);
  return W != 0 ? 1 : 0;
}

static hipError_t za_launch_fast(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  int W = 0, nw, nh, mw = 2;
  if (d2_wanted(a->frames)) d2_geometry(b, a->frames, W, nw, nh, mw);
  if (W == 0) return ddt_wide_launch(b, a, st);
  const size_t lds = d2_lds_bytes(W, nw);
  typedef void (*D2Fn)(ZabBatch, ZabAudio, int, int);
  static const D2Fn fns[2][4] = {{zab_ddt_fast, zab_ddt_fast_nw2, zab_ddt_fast_nw4, zab_ddt_fast_nw8},
                                 {zab_ddt_fast, zab_ddt_fast_nw2w3, zab_ddt_fast_nw4w3, zab_ddt_fast_nw8w3}};
  static ZaPerDevice once;               // (function attributes are per device: a group runs one engine per GPU)
  once.once([] {
    const int cap = 160 * 1024 - 512;
    for (int m = 0; m < 2; ++m)
      for (int k = 0; k < 4; ++k) (void)hipFuncSetAttribute((const void*)fns[m][k], hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  });
  if (nw == 1) mw = 2;
  const int ki = nw == 1 ? 0 : nw == 2 ? 1 : nw == 4 ? 2 : 3;
  const D2Fn fn = fns[mw - 2][ki];
  const dim3 grid(b->n_inst), block(64 * nw);
  if (nw == 1) snprintf(ddt_kernel_name, sizeof ddt_kernel_name, "zab_ddt_fast");
  else snprintf(ddt_kernel_name, sizeof ddt_kernel_name, mw == 3 ? "zab_ddt_fast_nw%dw3" : "zab_ddt_fast_nw%d", nw);
  if (getenv("ZAB_DDT_DEBUG")) {                           // geometry and residency of the launch
    int wgs = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, (const void*)fn, 64 * nw, lds);
    fprintf(stderr, "%s: nw %d W %d nh %d lds %zu B -> %d workgroups (%d waves) per CU\n", ddt_kernel_name, nw, W, nh, lds, wgs, wgs * nw);
  }
  hipLaunchKernelGGL(fn, grid, block, lds, st, *b, *a, W, nh);
  return hipGetLastError();
}
