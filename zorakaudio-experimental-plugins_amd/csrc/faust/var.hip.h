// var.hip.h -- Restoration/VAR, restated from "Vocal Air Recovery (VAR).dsp" (reference:
// plugins/Restoration/VAR/src/Vocal Air Recovery (VAR).dsp; line numbers refer to it). f32 throughout.
// Library pieces as published in stdfaust.lib (see oracle/faust_ref.c for the list); parity unpinned.
#pragma once

#include "faust_lane.hip.h"

struct ZfVar {
  static constexpr int NCH = 2;
  static constexpr int NPARAM = 3;       // Air Amount [%], Sensitivity [%], Detector Floor [dB] (:93-95)
  // state (order = oracle's var_state): smAmount smSens | det[2]{s1,s2} | (hf1,hf2)[2]{s1,s2} | air{s1,s2} hfLvl |
  //                                     (sm1 sm2 d1 d2)[2] | env | rnd_hi rnd_lo
  static constexpr int S_SMA = 0, S_SMS = 1, S_DET = 2, S_HF = 6, S_AIR = 14, S_HFLVL = 16, S_SM = 17, S_ENV = 25, S_RND = 26;
  static constexpr int NSTATE = 28;
  static const char* const names[NSTATE];

  struct Biq { float b0, b1, b2, a1, a2; };
  struct Ctl {
    float s, amountT, sensT, floorLin, detSmooth_a, hfLvl_a, catt, crel, airBase;
    Biq det, hf, air;
  };
  ZF_FN static float safeFc(float fc, float SR) { return zf_min(fc, 0.45f * SR); }                 // :12
  ZF_FN static Biq rbj(int kind, float fc, float Q, float SR) {                                   // :15-86
    Biq c;
    const float f = safeFc(fc, SR), q = zf_max(0.001f, Q);
    const float w0 = 6.2831855f * f / SR;
    const float cw = zf_cos(w0), sw = zf_sin(w0);
    const float alpha = sw / (2.0f * q);
    float bb0, bb1, bb2;
    if (kind == 0) { bb0 = (1.0f + cw) / 2.0f; bb1 = -(1.0f + cw); bb2 = (1.0f + cw) / 2.0f; }
    else if (kind == 1) { bb0 = (1.0f - cw) / 2.0f; bb1 = 1.0f - cw; bb2 = (1.0f - cw) / 2.0f; }
    else { bb0 = sw / 2.0f; bb1 = 0.0f; bb2 = -sw / 2.0f; }
    const float aa0 = 1.0f + alpha, aa1 = -2.0f * cw, aa2 = 1.0f - alpha;
    c.b0 = bb0 / aa0; c.b1 = bb1 / aa0; c.b2 = bb2 / aa0; c.a1 = aa1 / aa0; c.a2 = aa2 / aa0;
    return c;
  }
  ZF_FN static float db2linear(float l) { return zf_pow(10.0f, l / 20.0f); }
  ZF_FN static Ctl control(const float* p, float SR) {
    Ctl c;
    c.s = 1.0f - 44.1f / SR;                                                                      // si.smoo
    c.amountT = p[0] / 100.0f; c.sensT = p[1] / 100.0f;
    c.floorLin = db2linear(p[2]);
    c.det = rbj(2, 9500.0f, 1.0f, SR); c.hf = rbj(0, 11500.0f, 0.707f, SR); c.air = rbj(2, 16000.0f, 1.2f, SR);
    c.detSmooth_a = zf_exp(-6.2831855f * safeFc(8500.0f, SR) / SR);                               // :112-113
    c.hfLvl_a = zf_exp(-1.0f / (SR * 0.14f));                                                     // :130
    c.catt = zf_exp(-1.0f / (0.0025f * SR)); c.crel = zf_exp(-1.0f / (0.080f * SR));              // ba.tau2pole
    c.airBase = db2linear(-34.0f);
    return c;
  }
  ZF_FN static float tf22t(float* z, const Biq& k, float x) {                                     // fi.tf22t
    const float y = k.b0 * x + z[0];
    z[0] = (k.b1 * x - k.a1 * y) + z[1];
    z[1] = k.b2 * x - k.a2 * y;
    return y;
  }
  ZF_FN static float smooth(float s, float x, float& y) { y = x * (1.0f - s) + s * y; return y; }

  ZF_FN static void frame(float* st, const Ctl& c, float* io) {
    const float eps = 1e-12f;
    const float amount = smooth(c.s, c.amountT, st[S_SMA]), sens = smooth(c.s, c.sensT, st[S_SMS]);
    const float maxExp_lin = db2linear(5.0f * amount);                                            // :103-104
    const float airMix = 0.25f * amount;
    const float thrN = 0.18f - 0.13f * sens;
    float detv[2], curv[2], hfv[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float x = io[ch];
      float* sm = st + S_SM + 4 * ch;                                                             // sm1 sm2 d1 d2
      detv[ch] = tf22t(st + S_DET + 2 * ch, c.det, x);
      sm[0] = detv[ch] * (1.0f - c.detSmooth_a) + c.detSmooth_a * sm[0];                          // :137-138
      const float s1prev = sm[2], s2prev = sm[3];
      sm[1] = sm[0] * (1.0f - c.detSmooth_a) + c.detSmooth_a * sm[1];
      const float s0 = sm[1];
      const float lap = s0 - 2.0f * s1prev + s2prev;                                              // :143-151
      const float denom = fabsf(s0) + 2.0f * fabsf(s1prev) + fabsf(s2prev) + eps;
      curv[ch] = fabsf(lap) / denom;
      sm[3] = s1prev; sm[2] = s0;
      hfv[ch] = tf22t(st + S_HF + 4 * ch + 2, c.hf, tf22t(st + S_HF + 4 * ch, c.hf, x));         // :167-168
    }
    const float hfAbs = 0.5f * (fabsf(detv[0]) + fabsf(detv[1]));                                 // :129-133
    st[S_HFLVL] = hfAbs * (1.0f - c.hfLvl_a) + c.hfLvl_a * st[S_HFLVL];
    const float uu = zf_min(1.0f, zf_max(0.0f, (st[S_HFLVL] / (c.floorLin + eps) - 1.0f) / (2.0f - 1.0f)));
    const float gate = uu * uu * (3.0f - 2.0f * uu);
    const float curvN = 0.5f * (curv[0] + curv[1]);
    const float cf = (curvN > st[S_ENV]) ? c.catt : c.crel;                                       // si.onePoleSwitching (:160)
    st[S_ENV] = (1.0f - cf) * curvN + cf * st[S_ENV];
    const float u = zf_max(0.0f, st[S_ENV] / thrN - 1.0f);                                        // :163-165
    const float t = (u / (1.0f + u)) * gate;
    const float t2 = zf_pow(zf_max(eps, t), 1.8f);
    const float g = 1.0f + t * (maxExp_lin - 1.0f);
    uint32_t r = ((uint32_t)(int32_t)st[S_RND] << 16) | ((uint32_t)(int32_t)st[S_RND + 1] & 0xffffu);   // no.noise
    r = r * 1103515245u + 12345u;
    const int32_t ri = (int32_t)r;
    st[S_RND] = (float)(ri >> 16); st[S_RND + 1] = (float)(ri & 0xffff);
    const float nz = (float)ri / 2147483647.0f;
    const float air = tf22t(st + S_AIR, c.air, nz);                                               // :177-178 (nL == nR)
    const float airGain = (t2 * c.airBase) * airMix;
    io[0] = (io[0] + hfv[0] * (g - 1.0f) + air * airGain) * 1.0f;                                 // :187-188
    io[1] = (io[1] + hfv[1] * (g - 1.0f) + air * airGain) * 1.0f;
  }
};
const char* const ZfVar::names[ZfVar::NSTATE] = {
    "smoo_amount", "smoo_sens", "detL.s1", "detL.s2", "detR.s1", "detR.s2", "hf1L.s1", "hf1L.s2", "hf2L.s1", "hf2L.s2",
    "hf1R.s1", "hf1R.s2", "hf2R.s1", "hf2R.s2", "air.s1", "air.s2", "hfLvl", "sm1L", "sm2L", "sm2L'", "sm2L''",
    "sm1R", "sm2R", "sm2R'", "sm2R''", "env", "noise.hi16", "noise.lo16"};
