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
