// var_wave.hip.h -- wave-per-instance kernel for Restoration/VAR ("Vocal Air Recovery (VAR).dsp"). Two serial passes per
// 64-frame chunk instead of ~28 dependent state updates per frame:
//   S1, lanes 0/1 = channel L/R: detector band-pass -> two smoothing one-poles, and the two cascaded HF high-pass biquads
//       (two independent chains interleaved in one loop);
//   S2, lane 0: the two parameter smoothers (si.smoo), the detector-level one-pole, the attack/release envelope of the
//       normalised curvature and the "air" band-pass on the noise (five independent chains interleaved).
// Everything else -- the Laplacian of the smoothed detector, gate, soft trigger, the two per-sample pow(), the mix -- runs
// one lane per frame. The no.noise LCG (int32 wrap-around) is advanced in parallel with per-lane jump constants.
// Arithmetic identical, operation for operation, to ZfVar::frame (var.hip.h).
#pragma once

#include "faust_wave.hip.h"
#include "var.hip.h"

#define ZF_VAR_FAST_NAME "zf_var_wave"

__global__ void __launch_bounds__(64) zf_var_wave(ZabBatch b, ZabAudio a) {
  using L = ZfVar;
  __shared__ float xin[2][64], detr[2][64], sm2r[2][64], hfr[2][64];
  __shared__ float hfabs[64], curvn[64], nzr[64], amr[64], ser[64], lvr[64], envr[64], airr[64];
  __shared__ float dly[2][2];                      // sm2' and sm2'' of each channel at the chunk boundary
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  float par[L::NPARAM], st[L::NSTATE];
#pragma unroll
  for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
#pragma unroll
  for (int k = 0; k < L::NSTATE; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];
  const L::Ctl c = L::control(par, zf_sr(b.srate));
  if (lane == 0) b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  // channel-owned state of lanes 0/1 (same code on both lanes, each with its channel's registers)
  const int ch = lane & 1;
  float detz[2] = {st[L::S_DET + 2 * ch], st[L::S_DET + 2 * ch + 1]};
  float hf1z[2] = {st[L::S_HF + 4 * ch], st[L::S_HF + 4 * ch + 1]}, hf2z[2] = {st[L::S_HF + 4 * ch + 2], st[L::S_HF + 4 * ch + 3]};
  float sm1 = st[L::S_SM + 4 * ch], sm2 = st[L::S_SM + 4 * ch + 1];
  if (lane < 2) { dly[lane][0] = st[L::S_SM + 4 * lane + 2]; dly[lane][1] = st[L::S_SM + 4 * lane + 3]; }
  // lane 0's own chains
  float smA = st[L::S_SMA], smS = st[L::S_SMS], hfLvl = st[L::S_HFLVL], env = st[L::S_ENV];
  float airz[2] = {st[L::S_AIR], st[L::S_AIR + 1]};
  // no.noise: r[n+1] = r[n] * 1103515245 + 12345 (mod 2^32); lane l needs r advanced l + 1 times: r * A^(l+1) + 12345 * S_l
  uint32_t rnd = ((uint32_t)(int32_t)st[L::S_RND] << 16) | ((uint32_t)(int32_t)st[L::S_RND + 1] & 0xffffu);
  uint32_t jumpA = 1103515245u, jumpS = 1u;
  for (int k = 0; k < lane; ++k) { jumpS = jumpS * 1103515245u + 1u; jumpA *= 1103515245u; }
  const float* in0 = a.in + (int64_t)inst * 2 * a.frame_stride;
  float* out0 = a.out + (int64_t)inst * 2 * a.frame_stride;
  const float eps = 1e-12f;
  const float one_m_ds = 1.0f - c.detSmooth_a, one_m_lv = 1.0f - c.hfLvl_a;
  float nxL = lane < a.frames ? in0[lane] : 0.0f, nxR = lane < a.frames ? in0[a.frame_stride + lane] : 0.0f;
  zf_wave_sync();
  for (int64_t t0 = 0; t0 < a.frames; t0 += 64) {
    const int tn = (int)((a.frames - t0 < 64) ? (a.frames - t0) : 64);
    const float xL = nxL, xR = nxR;
    {
      const int64_t t = t0 + 64 + lane;
      nxL = t < a.frames ? in0[t] : 0.0f;
      nxR = t < a.frames ? in0[a.frame_stride + t] : 0.0f;
    }
    xin[0][lane] = xL; xin[1][lane] = xR;
    const uint32_t ri = rnd * jumpA + 12345u * jumpS;                                  // this frame's LCG state
    nzr[lane] = (float)(int32_t)ri / 2147483647.0f;
    zf_wave_sync();
    rnd = (uint32_t)__shfl((int)ri, tn - 1, 64);                                       // state after the chunk's last frame
    // ---- S1: per-channel chains on lanes 0 / 1 -----------------------------------------------------------------------------
    if (lane < 2) {
#pragma unroll 8
      for (int n = 0; n < tn; ++n) {
        const float x = xin[lane][n];
        const float det = L::tf22t(detz, c.det, x);                                    // :126-127
        sm1 = det * one_m_ds + c.detSmooth_a * sm1;                                    // :137-138
        sm2 = sm1 * one_m_ds + c.detSmooth_a * sm2;
        const float hf = L::tf22t(hf2z, c.hf, L::tf22t(hf1z, c.hf, x));                // :167-168
        detr[lane][n] = det; sm2r[lane][n] = sm2; hfr[lane][n] = hf;
      }
    }
    zf_wave_sync();
    // ---- M2: curvature of the smoothed detector, detector level, per frame -------------------------------------------------
    float curv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float s0 = sm2r[k][lane];
      const float s1p = lane >= 1 ? sm2r[k][lane - 1] : dly[k][0];
      const float s2p = lane >= 2 ? sm2r[k][lane - 2] : (lane == 1 ? dly[k][0] : dly[k][1]);
      const float lap = s0 - 2.0f * s1p + s2p;                                         // :143-151
      const float denom = fabsf(s0) + 2.0f * fabsf(s1p) + fabsf(s2p) + eps;
      curv[k] = fabsf(lap) / denom;
    }
    hfabs[lane] = 0.5f * (fabsf(detr[0][lane]) + fabsf(detr[1][lane]));                // :129
    curvn[lane] = 0.5f * (curv[0] + curv[1]);
    zf_wave_sync();
    if (lane < 2) {                                                                    // x', x'' for the next chunk
      const float d1 = sm2r[lane][tn - 1];
      const float d2 = tn >= 2 ? sm2r[lane][tn - 2] : dly[lane][0];
      dly[lane][0] = d1; dly[lane][1] = d2;
    }
    // ---- S2: lane 0, five independent chains -------------------------------------------------------------------------------
    if (lane == 0) {
#pragma unroll 8
      for (int n = 0; n < tn; ++n) {
        amr[n] = L::smooth(c.s, c.amountT, smA);                                       // si.smoo (:93-94)
        ser[n] = L::smooth(c.s, c.sensT, smS);
        hfLvl = hfabs[n] * one_m_lv + c.hfLvl_a * hfLvl;                               // :131
        lvr[n] = hfLvl;
        const float cn = curvn[n];
        const float cf = (cn > env) ? c.catt : c.crel;                                 // si.onePoleSwitching (:160)
        env = (1.0f - cf) * cn + cf * env;
        envr[n] = env;
        airr[n] = L::tf22t(airz, c.air, nzr[n]);                                       // :177-178
      }
    }
    zf_wave_sync();
    // ---- M3: trigger, gains, mix, per frame -----------------------------------------------------------------------------------
    {
      const float amount = amr[lane], sens = ser[lane];
      const float maxExp_lin = L::db2linear(5.0f * amount);                            // :103-104
      const float airMix = 0.25f * amount;
      const float thrN = 0.18f - 0.13f * sens;
      const float uu = zf_min(1.0f, zf_max(0.0f, (lvr[lane] / (c.floorLin + eps) - 1.0f) / (2.0f - 1.0f)));
      const float gate = uu * uu * (3.0f - 2.0f * uu);                                 // :133
      const float u = zf_max(0.0f, envr[lane] / thrN - 1.0f);                          // :163-165
      const float t = (u / (1.0f + u)) * gate;
      const float t2 = zf_pow(zf_max(eps, t), 1.8f);
      const float g = 1.0f + t * (maxExp_lin - 1.0f);
      const float airGain = (t2 * c.airBase) * airMix;
      const float air = airr[lane];
      if (lane < tn) {
        out0[t0 + lane] = (xL + hfr[0][lane] * (g - 1.0f) + air * airGain) * 1.0f;     // :187-188
        out0[a.frame_stride + t0 + lane] = (xR + hfr[1][lane] * (g - 1.0f) + air * airGain) * 1.0f;
      }
    }
    zf_wave_sync();
  }
#define ZF_PUT(K, v) b.vars[(K) * b.var_se + inst * b.var_si] = (double)(v)
  if (lane < 2) {
    ZF_PUT(L::S_DET + 2 * lane, detz[0]); ZF_PUT(L::S_DET + 2 * lane + 1, detz[1]);
    ZF_PUT(L::S_HF + 4 * lane, hf1z[0]); ZF_PUT(L::S_HF + 4 * lane + 1, hf1z[1]);
    ZF_PUT(L::S_HF + 4 * lane + 2, hf2z[0]); ZF_PUT(L::S_HF + 4 * lane + 3, hf2z[1]);
    ZF_PUT(L::S_SM + 4 * lane, sm1); ZF_PUT(L::S_SM + 4 * lane + 1, sm2);
    ZF_PUT(L::S_SM + 4 * lane + 2, dly[lane][0]); ZF_PUT(L::S_SM + 4 * lane + 3, dly[lane][1]);
  }
  if (lane == 0) {
    ZF_PUT(L::S_SMA, smA); ZF_PUT(L::S_SMS, smS); ZF_PUT(L::S_HFLVL, hfLvl); ZF_PUT(L::S_ENV, env);
    ZF_PUT(L::S_AIR, airz[0]); ZF_PUT(L::S_AIR + 1, airz[1]);
    const int32_t rs = (int32_t)rnd;
    ZF_PUT(L::S_RND, (float)(rs >> 16)); ZF_PUT(L::S_RND + 1, (float)(rs & 0xffff));
  }
#undef ZF_PUT
}

static int32_t zf_var_applies(const ZabBatch*, const ZabAudio* a) { return a->frames > 0 ? 1 : 0; }
static hipError_t zf_var_launch(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  hipLaunchKernelGGL(zf_var_wave, dim3(b->n_inst), dim3(64), 0, st, *b, *a);
  return hipGetLastError();
}
