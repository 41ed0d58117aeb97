// red.hip.h -- Dynamics/RED, restated from "Reverb Expanding Downwards (RED).dsp" (reference:
// plugins/Dynamics/RED/src/Reverb Expanding Downwards (RED).dsp; line numbers refer to it). 6 in / 6 out, f32.
#pragma once

#include "faust_lane.hip.h"

struct ZfRed {
  static constexpr int NCH = 6;
  static constexpr int NPARAM = 3;       // Amount (max duck dB), Sensitivity (%), Release (ms) (:8-10)
  static constexpr int S_WET = 0, S_REF = 1, S_OFF = 2, S_TGT = 3, S_DRY = 4, S_HOLD = 5, S_GRN = 6, S_GRF = 7;
  static constexpr int NSTATE = 8;
  static const char* const names[NSTATE];

  struct Ctl {
    float maxduck_dB, thr_db, ratio, knee, pole_rms, pole_tgt, pole_grace, pole_hold, pole_10, dry_on_lin, ref_off_lin,
        floor_lin, catt, crel, crel_in;
  };
  ZF_FN static float ms2pole(float ms, float SR) { return zf_exp(-1.0f / (SR * (ms / 1000.0f))); }   // :22
  ZF_FN static float clampf(float x, float lo, float hi) { return zf_max(lo, zf_min(hi, x)); }        // :15
  ZF_FN static float smoothstep01(float x) { const float x1 = clampf(x, 0.0f, 1.0f); return x1 * x1 * (3.0f - 2.0f * x1); }
  ZF_FN static Ctl control(const float* p, float SR) {
    Ctl c;
    const float sens = p[1] / 100.0f, rel_ms = p[2];
    c.maxduck_dB = p[0];
    c.thr_db = 18.0f - sens * 21.0f; c.ratio = 1.2f + sens * 3.0f;                                  // :49-51
    c.knee = zf_max(10.0f - sens * 6.0f, 0.001f);
    const float grace_ms = clampf(rel_ms * 0.25f, 60.0f, 200.0f);
    c.pole_rms = ms2pole(35.0f, SR); c.pole_tgt = ms2pole(25.0f, SR); c.pole_grace = ms2pole(grace_ms, SR);
    c.pole_hold = ms2pole(80.0f, SR); c.pole_10 = ms2pole(10.0f, SR);
    c.dry_on_lin = zf_pow(10.0f, -50.0f / 20.0f); c.ref_off_lin = zf_pow(10.0f, -60.0f / 20.0f);
    c.floor_lin = zf_pow(10.0f, -80.0f / 20.0f);
    c.catt = zf_exp(-1.0f / ((12.0f / 1000.0f) * SR));                                              // ba.tau2pole
    c.crel = zf_exp(-1.0f / ((rel_ms / 1000.0f) * SR)); c.crel_in = zf_exp(-1.0f / ((70.0f / 1000.0f) * SR));
    return c;
  }
  ZF_FN static float smooth(float s, float x, float& y) { y = x * (1.0f - s) + s * y; return y; }    // si.smooth
  ZF_FN static float ar(float catt, float crel, float x, float& y) {                                 // si.onePoleSwitching
    const float cf = (x > y) ? catt : crel;
    y = (1.0f - cf) * x + cf * y;
    return y;
  }
  ZF_FN static void frame(float* st, const Ctl& c, float* io) {
    const float eps = 1e-12f;
    const float wetL = io[0], wetR = io[1], refL = io[4], refR = io[5];
    const float wet_p = 0.5f * (wetL * wetL + wetR * wetR), ref_p = 0.5f * (refL * refL + refR * refR);   // :66-67
    const float wet_env2 = smooth(c.pole_rms, wet_p, st[S_WET]), ref_env2 = smooth(c.pole_rms, ref_p, st[S_REF]);
    const float Ey = zf_max(sqrtf(zf_max(wet_env2, 0.0f)), c.floor_lin), Ex = zf_max(sqrtf(zf_max(ref_env2, 0.0f)), c.floor_lin);
    const float dryA = (float)(Ex > c.dry_on_lin), offA = (float)(Ex <= c.ref_off_lin);              // :76-77
    const float offA_s = smooth(c.pole_grace, offA, st[S_OFF]);
    const float tail_w = (1.0f - offA) + offA * smoothstep01(offA_s);                                // :82
    const float rdB = 20.0f * zf_log10(zf_max((Ey + eps) / (Ex + eps), 1e-30f));                     // :85
    const float over = rdB - c.thr_db;
    const float over_eff = (over <= 0.0f) ? 0.0f : over * smoothstep01(clampf(over / c.knee, 0.0f, 1.0f));
    const float tgt0 = (over_eff > 0.0f) ? zf_min(c.maxduck_dB, over_eff * c.ratio) : 0.0f;          // :95-96
    const float tgt1 = tgt0 * tail_w;
    const float tgt_db = smooth(c.pole_tgt, tgt1, st[S_TGT]);                                        // :99
    const float dryA_s = smooth(c.pole_10, dryA, st[S_DRY]);                                         // :105
    const float tgt_hold = zf_max(tgt_db, smooth(c.pole_hold, tgt_db, st[S_HOLD]));                  // :110
    const float tgt_pin = (1.0f - dryA) * tgt_hold + dryA * tgt_db;
    const float gr_norm = ar(c.catt, c.crel, fabsf(tgt_pin), st[S_GRN]);                             // :118-119
    const float gr_fast = ar(c.catt, c.crel_in, fabsf(tgt_pin), st[S_GRF]);
    const float gr_db = (1.0f - dryA_s) * gr_norm + dryA_s * gr_fast;                                // :122
    const float g = zf_pow(10.0f, (0.0f - gr_db) / 20.0f);                                           // :125
    io[0] = wetL * g; io[1] = wetR * g;                                                              // channels 3..6 pass through
  }
};
const char* const ZfRed::names[ZfRed::NSTATE] = {"wet_env2", "ref_env2", "offA_s", "tgt_db", "dryA_s", "tgt_hold", "gr_norm", "gr_fast"};
