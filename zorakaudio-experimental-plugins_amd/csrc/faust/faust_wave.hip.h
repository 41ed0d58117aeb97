// faust_wave.hip.h -- helpers for wave-per-instance kernels of Faust leaves whose recursions are short one-poles separated
// by feed-forward maps (ModTilt, RED, VAR): the maps run ONE LANE PER FRAME over a 64-frame chunk, each recursion runs on
// lane 0 over the chunk's 64 values through an LDS row (same operations in the same order as the lane-per-instance kernel,
// hence the same bits), so 1024 instances occupy 1024 wavefronts instead of 16 and the per-sample libm calls (log10, pow,
// sqrt) are spread over 64 lanes.
#pragma once

#include "faust_lane.hip.h"

// Run `step(x) -> y` over row[0 .. tn) in place, serially, on lane 0. Full chunks take the straight-line path (all LDS
// reads up front); the ragged last chunk of a launch takes the loop.
template <class F>
__device__ __forceinline__ void zf_serial64(float* row, int lane, int tn, F step) {
  if (lane == 0) {
    if (tn == 64) {
      float v[64];
#pragma unroll
      for (int n = 0; n < 64; ++n) v[n] = row[n];
#pragma unroll
      for (int n = 0; n < 64; ++n) row[n] = step(v[n]);
    } else {
      for (int n = 0; n < tn; ++n) row[n] = step(row[n]);
    }
  }
  zf_wave_sync();
}

// K independent recursions of the SAME shape at once: lane k (k < K) runs `step(k, x) -> y` over rows[k][0 .. tn) in place.
// `step` must use lane-local state only (each lane keeps its own copy; lane k's copy of recursion k is the authoritative one).
template <int K, class F>
__device__ __forceinline__ void zf_serial64_rows(float (*rows)[64], int lane, int tn, F step) {
  if (lane < K) {
    float* row = rows[lane];
    if (tn == 64) {
      float v[64];
#pragma unroll
      for (int n = 0; n < 64; ++n) v[n] = row[n];
#pragma unroll
      for (int n = 0; n < 64; ++n) row[n] = step(lane, v[n]);
    } else {
      for (int n = 0; n < tn; ++n) row[n] = step(lane, row[n]);
    }
  }
  zf_wave_sync();
}

// ---- constant-coefficient one-pole over the 64 frames of a chunk as a wave scan (f32) ---------------------------------------------
// y[n] = u[n] + q * y[n-1] with lane = frame: a Kogge-Stone scan with DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31) on the lane
// values, weights q^(2^j) and two per-lane powers, then q^(lane+1) times the state carried into the chunk. The additions are
// re-associated against the serial order (~1e-7 relative in f32), so it is only used where what follows is continuous in
// the filter's output (no hard gates downstream): the tests hold such a kernel to a tolerance, not to the bits.
struct ZfPoleScan {
  float q1, q2, q4, q8;      // q^1, q^2, q^4, q^8            (uniform)
  float cb1, cb2, pl;        // per lane: q^((l&15)+1), q^(l-31) (lanes >= 32, else unused), q^(l+1)
};
__device__ __forceinline__ ZfPoleScan zf_pole_scan_init(float q, int lane) {
  auto ipow = [](double b, int e) { double r = 1.0; while (e) { if (e & 1) r *= b; b *= b; e >>= 1; } return r; };
  ZfPoleScan w;
  const double Q = (double)q;
  w.q1 = q; w.q2 = (float)(Q * Q); w.q4 = (float)ipow(Q, 4); w.q8 = (float)ipow(Q, 8);
  w.cb1 = (float)ipow(Q, (lane & 15) + 1);
  w.cb2 = (float)ipow(Q, lane >= 32 ? lane - 31 : 0);
  w.pl = (float)ipow(Q, lane + 1);
  return w;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ float zf_dpp(float v) {       // lanes without a source (or masked out by ROWS) receive 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, true));
}
// u = the lane's input term (already scaled), carry = the state before the chunk; returns y of every lane
__device__ __forceinline__ float zf_pole_scan(const ZfPoleScan& w, float u, float carry) {
  float g = u;
  g = __builtin_fmaf(w.q1, zf_dpp<0x111, 0xF>(g), g);
  g = __builtin_fmaf(w.q2, zf_dpp<0x112, 0xF>(g), g);
  g = __builtin_fmaf(w.q4, zf_dpp<0x114, 0xF>(g), g);
  g = __builtin_fmaf(w.q8, zf_dpp<0x118, 0xF>(g), g);
  g = __builtin_fmaf(w.cb1, zf_dpp<0x142, 0xA>(g), g);
  g = __builtin_fmaf(w.cb2, zf_dpp<0x143, 0xC>(g), g);
  return __builtin_fmaf(w.pl, carry, g);
}
__device__ __forceinline__ float zf_readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
