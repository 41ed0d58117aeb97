// modtilt_wave.hip.h -- wave-per-instance kernel for Dynamics/ModTilt (ModTilt.dsp): five one-pole recursions in a row
// (env2 -> base -> pivot LPF -> ratio smoother -> auto-trim), each fed by a feed-forward map of the previous one.
// Everything runs ONE LANE PER FRAME: the maps as in ZfModTilt::frame (modtilt.hip.h), the five one-poles as wave scans
// (faust_wave.hip.h zf_pole_scan) -- they are constant-coefficient and everything downstream of them is continuous (max / min
// clamps, no gates), so re-associating their sums moves the output by ~1e-7 relative and nothing else. The lane-per-instance
// kernel (ZAB_PATH_GENERIC) keeps the restatement's operation order and bits; tests hold this one to 1e-6.
// Round 1's form of this kernel ran each recursion serially on lane 0 through an LDS row: 5.1 ms at 1024 x 48 000.
#pragma once

#include "faust_wave.hip.h"
#include "modtilt.hip.h"

#define ZF_MODTILT_FAST_NAME "zf_modtilt_wave"

__global__ void __launch_bounds__(64) zf_modtilt_wave(ZabBatch b, ZabAudio a) {
  using L = ZfModTilt;
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  float par[L::NPARAM], st[L::NSTATE];
#pragma unroll
  for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
#pragma unroll
  for (int k = 0; k < L::NSTATE; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];   // (wave-uniform)
  const L::Ctl c = L::control(par, zf_sr(b.srate));
  if (lane == 0) b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  // onepole(a): y = x * a + y * (1 - a)  ->  scan weight q = 1 - a (formed in f32, as the restatement forms it)
  const float aa[5] = {c.a_env, c.a_base, c.a_piv, 0.05f, c.a_trim};
  ZfPoleScan w[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) w[k] = zf_pole_scan_init(1.0f - aa[k], lane);
  const float* in0 = a.in + (int64_t)inst * 2 * a.frame_stride;
  float* out0 = a.out + (int64_t)inst * 2 * a.frame_stride;
  float nxL = lane < a.frames ? in0[lane] : 0.0f, nxR = lane < a.frames ? in0[a.frame_stride + lane] : 0.0f;
  for (int64_t t0 = 0; t0 < a.frames; t0 += 64) {
    const int tn = (int)((a.frames - t0 < 64) ? (a.frames - t0) : 64);
    const int last = tn - 1;                                             // the lane whose outputs are the next chunk's states
    const float xL = nxL, xR = nxR;
    {
      const int64_t t = t0 + 64 + lane;                                  // next chunk's HBM read, a chunk ahead
      nxL = t < a.frames ? in0[t] : 0.0f;
      nxR = t < a.frames ? in0[a.frame_stride + t] : 0.0f;
    }
    const float x = 0.5f * (xL + xR);                                    // :52
    const float env2 = zf_pole_scan(w[0], (x * x) * c.a_env, st[L::S_ENV2]);                        // :55-56
    const float e = sqrtf(zf_max(env2, 0.0f));                           // :57
    const float bs = zf_pole_scan(w[1], e * c.a_base, st[L::S_BASE]);    // :60
    const float m = e - bs;                                              // :63
    const float l = zf_pole_scan(w[2], m * c.a_piv, st[L::S_LP]);        // pivot LPF (:66)
    const float m_lo = l, m_hi = m - l;                                  // :67-68
    const float m2_tilt = m_lo * c.g_lo + m_hi * c.g_hi;                 // :71-72
    const float m2 = m * 0.25f + m2_tilt * 0.75f;
    const float env_t = bs + m2;                                         // :75
    const float env_tp = zf_max(env_t, 0.05f * e);                       // :78-80
    const float r0 = (env_tp + 1e-9f) / (e + 1e-9f);
    const float r0c = zf_min(zf_max(r0, 0.67f), 1.5f);                   // :83
    const float rs = zf_pole_scan(w[3], (r0c - 1.0f) * 0.05f, st[L::S_RS]);                         // ratio smoother (:86)
    const float r_s = 1.0f + rs;
    const float rdb = 20.0f * zf_log10(zf_max(r_s, 1e-12f));             // :90
    const float mean = zf_pole_scan(w[4], rdb * c.a_trim, st[L::S_MEAN]);                           // auto-trim (:91)
    const float trim = zf_pow(10.0f, (0.0f - mean) / 20.0f);             // :92
    const float yL = xL * r_s, yR = xR * r_s;                            // :95-99
    if (lane < tn) {
      out0[t0 + lane] = (xL * (1.0f - c.mix) + yL * c.mix) * trim;
      out0[a.frame_stride + t0 + lane] = (xR * (1.0f - c.mix) + yR * c.mix) * trim;
    }
    st[L::S_ENV2] = zf_readlane_f(env2, last); st[L::S_BASE] = zf_readlane_f(bs, last); st[L::S_LP] = zf_readlane_f(l, last);
    st[L::S_RS] = zf_readlane_f(rs, last); st[L::S_MEAN] = zf_readlane_f(mean, last);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < L::NSTATE; ++k) b.vars[k * b.var_se + inst * b.var_si] = (double)st[k];
  }
}

static int32_t zf_modtilt_applies(const ZabBatch*, const ZabAudio* a) { return a->frames > 0 ? 1 : 0; }
static hipError_t zf_modtilt_launch(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  hipLaunchKernelGGL(zf_modtilt_wave, dim3(b->n_inst), dim3(64), 0, st, *b, *a);
  return hipGetLastError();
}
