// red_wave.hip.h -- wave-per-instance kernel for Dynamics/RED ("Reverb Expanding Downwards (RED).dsp"): eight one-pole /
// attack-release recursions in five dependent groups, with the detector maths (sqrt, log10, smoothsteps, pow) between them
// running one lane per frame. Arithmetic identical, operation for operation, to ZfRed::frame (red.hip.h).
#pragma once

#include "faust_wave.hip.h"
#include "red.hip.h"

#define ZF_RED_FAST_NAME "zf_red_wave"

__global__ void __launch_bounds__(64) zf_red_wave(ZabBatch b, ZabAudio a) {
  using L = ZfRed;
  __shared__ float r2[2][64];          // two rows for the paired recursions
  __shared__ float r1[64];
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  float par[L::NPARAM], st[L::NSTATE];
#pragma unroll
  for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
#pragma unroll
  for (int k = 0; k < L::NSTATE; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];
  const L::Ctl c = L::control(par, zf_sr(b.srate));
  if (lane == 0) b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  const float* in0 = a.in + (int64_t)inst * 6 * a.frame_stride;
  float* out0 = a.out + (int64_t)inst * 6 * a.frame_stride;
  const float eps = 1e-12f;
  // paired recursions run on lanes 0 and 1 with the SAME code: each lane keeps its member's state and coefficient
  float g1 = lane == 0 ? st[L::S_WET] : st[L::S_REF];
  float g3 = lane == 0 ? st[L::S_TGT] : st[L::S_DRY];
  float g5 = lane == 0 ? st[L::S_GRN] : st[L::S_GRF];
  const float pole3 = lane == 0 ? c.pole_tgt : c.pole_10, rel5 = lane == 0 ? c.crel : c.crel_in;
  for (int64_t t0 = 0; t0 < a.frames; t0 += 64) {
    const int tn = (int)((a.frames - t0 < 64) ? (a.frames - t0) : 64);
    float x[6];
#pragma unroll
    for (int ch = 0; ch < 6; ++ch) x[ch] = lane < tn ? in0[ch * a.frame_stride + t0 + lane] : 0.0f;
    const float wetL = x[0], wetR = x[1], refL = x[4], refR = x[5];
    r2[0][lane] = 0.5f * (wetL * wetL + wetR * wetR);                                   // wet_p, ref_p (:66-67)
    r2[1][lane] = 0.5f * (refL * refL + refR * refR);
    zf_wave_sync();
    // group 1: wet_env2 (lane 0), ref_env2 (lane 1)                                       :70-71
    zf_serial64_rows<2>(r2, lane, tn, [&](int, float v) { return L::smooth(c.pole_rms, v, g1); });
    const float Ey = zf_max(sqrtf(zf_max(r2[0][lane], 0.0f)), c.floor_lin), Ex = zf_max(sqrtf(zf_max(r2[1][lane], 0.0f)), c.floor_lin);
    const float dryA = (float)(Ex > c.dry_on_lin), offA = (float)(Ex <= c.ref_off_lin);   // :76-77
    r1[lane] = offA;
    zf_wave_sync();
    zf_serial64(r1, lane, tn, [&](float v) { return L::smooth(c.pole_grace, v, st[L::S_OFF]); });    // group 2: offA_s (:81)
    const float tail_w = (1.0f - offA) + offA * L::smoothstep01(r1[lane]);                // :82
    const float rdB = 20.0f * zf_log10(zf_max((Ey + eps) / (Ex + eps), 1e-30f));          // :85
    const float over = rdB - c.thr_db;
    const float over_eff = (over <= 0.0f) ? 0.0f : over * L::smoothstep01(L::clampf(over / c.knee, 0.0f, 1.0f));
    const float tgt0 = (over_eff > 0.0f) ? zf_min(c.maxduck_dB, over_eff * c.ratio) : 0.0f;          // :95-96
    zf_wave_sync();
    r2[0][lane] = tgt0 * tail_w;                                                          // tgt1
    r2[1][lane] = dryA;
    zf_wave_sync();
    // group 3: tgt_db (lane 0, pole_tgt), dryA_s (lane 1, 10 ms pole)                      :99,105
    zf_serial64_rows<2>(r2, lane, tn, [&](int, float v) { return L::smooth(pole3, v, g3); });
    const float tgt_db = r2[0][lane], dryA_s = r2[1][lane];
    r1[lane] = tgt_db;
    zf_wave_sync();
    zf_serial64(r1, lane, tn, [&](float v) { return L::smooth(c.pole_hold, v, st[L::S_HOLD]); });    // group 4: hold (:110)
    const float tgt_hold = zf_max(tgt_db, r1[lane]);
    const float tgt_pin = (1.0f - dryA) * tgt_hold + dryA * tgt_db;
    zf_wave_sync();
    r2[0][lane] = fabsf(tgt_pin); r2[1][lane] = fabsf(tgt_pin);
    zf_wave_sync();
    // group 5: gr_norm (lane 0, release), gr_fast (lane 1, release with reference present)   :118-119
    zf_serial64_rows<2>(r2, lane, tn, [&](int, float v) { return L::ar(c.catt, rel5, v, g5); });
    const float gr_db = (1.0f - dryA_s) * r2[0][lane] + dryA_s * r2[1][lane];             // :122
    const float g = zf_pow(10.0f, (0.0f - gr_db) / 20.0f);                                // :125
    if (lane < tn) {
      out0[t0 + lane] = wetL * g;
      out0[a.frame_stride + t0 + lane] = wetR * g;
#pragma unroll
      for (int ch = 2; ch < 6; ++ch) out0[ch * a.frame_stride + t0 + lane] = x[ch];      // pass-through channels
    }
    zf_wave_sync();
  }
  // lane 0 holds the states it advanced alone and those of the paired groups' first member; lane 1 the second members
#define ZF_PUT(K, v) b.vars[(K) * b.var_se + inst * b.var_si] = (double)(v)
  if (lane == 0) { ZF_PUT(L::S_WET, g1); ZF_PUT(L::S_TGT, g3); ZF_PUT(L::S_GRN, g5); ZF_PUT(L::S_OFF, st[L::S_OFF]); ZF_PUT(L::S_HOLD, st[L::S_HOLD]); }
  if (lane == 1) { ZF_PUT(L::S_REF, g1); ZF_PUT(L::S_DRY, g3); ZF_PUT(L::S_GRF, g5); }
#undef ZF_PUT
}

static int32_t zf_red_applies(const ZabBatch*, const ZabAudio* a) { return a->frames > 0 ? 1 : 0; }
static hipError_t zf_red_launch(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  hipLaunchKernelGGL(zf_red_wave, dim3(b->n_inst), dim3(64), 0, st, *b, *a);
  return hipGetLastError();
}
