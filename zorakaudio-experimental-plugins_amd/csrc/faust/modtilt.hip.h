// modtilt.hip.h -- Dynamics/ModTilt, restated from ModTilt.dsp (reference: plugins/Dynamics/ModTilt/src/ModTilt.dsp).
// f32 throughout; per-sample log10 / pow are evaluated in f64 and rounded once (as the CPU restatement does).
#pragma once

#include "faust_lane.hip.h"

struct ZfModTilt {
  static constexpr int NCH = 2;
  static constexpr int NPARAM = 3;       // Tilt (dB), Pivot (Hz), Mix (:9-19)
  static constexpr int S_ENV2 = 0, S_BASE = 1, S_LP = 2, S_RS = 3, S_MEAN = 4;
  static constexpr int NSTATE = 5;
  static const char* const names[NSTATE];

  struct Ctl { float a_env, a_base, a_piv, a_trim, g_hi, g_lo, mix; };

  ZF_FN static float a_from_hz(float hz, float SR) {                                            // :25-26
    return 1.0f - zf_exp((-6.2831855f * zf_max(hz, 0.001f)) / SR);
  }
  ZF_FN static Ctl control(const float* p, float SR) {
    Ctl c;
    c.a_env = a_from_hz(25.0f, SR);                                                             // :34-39
    c.a_base = a_from_hz(1.0f, SR);
    c.a_piv = a_from_hz(p[1], SR);
    c.a_trim = a_from_hz(0.2f, SR);
    c.g_hi = zf_pow(10.0f, (p[0] * 0.5f) / 20.0f);                                              // :43-44  ba.db2linear
    c.g_lo = zf_pow(10.0f, ((0.0f - p[0]) * 0.5f) / 20.0f);
    c.mix = p[2];
    return c;
  }
  // onepole(a) = *(a) : (+ ~ *(1.0 - a))                                                        :29
  ZF_FN static float onepole(float a, float x, float& y) { y = x * a + y * (1.0f - a); return y; }

  ZF_FN static void frame(float* st, const Ctl& c, float* io) {
    const float xL = io[0], xR = io[1];
    const float x = 0.5f * (xL + xR);                                                           // :52
    const float env2 = onepole(c.a_env, x * x, st[S_ENV2]);                                     // :55-56
    const float env = sqrtf(zf_max(env2, 0.0f));                                                // :57
    const float base = onepole(c.a_base, env, st[S_BASE]);                                      // :60
    const float m = env - base;                                                                 // :63
    const float lp = onepole(c.a_piv, m, st[S_LP]);                                             // :66-68
    const float m_lo = lp, m_hi = m - lp;
    const float m2_tilt = m_lo * c.g_lo + m_hi * c.g_hi;                                        // :71-72
    const float m2 = m * 0.25f + m2_tilt * 0.75f;
    const float env_t = base + m2;                                                              // :75
    const float env_tp = zf_max(env_t, 0.05f * env);                                            // :78-80
    const float r0 = (env_tp + 1e-9f) / (env + 1e-9f);
    const float r0c = zf_min(zf_max(r0, 0.67f), 1.5f);                                          // :83
    const float r_s = 1.0f + onepole(0.05f, r0c - 1.0f, st[S_RS]);                              // :86-87
    const float rdb = 20.0f * zf_log10(zf_max(r_s, 1e-12f));                                    // :90-92
    const float mean_rdb = onepole(c.a_trim, rdb, st[S_MEAN]);
    const float trim = zf_pow(10.0f, (0.0f - mean_rdb) / 20.0f);
    const float yL = xL * r_s, yR = xR * r_s;                                                   // :95-99
    io[0] = (xL * (1.0f - c.mix) + yL * c.mix) * trim;
    io[1] = (xR * (1.0f - c.mix) + yR * c.mix) * trim;
  }
};
const char* const ZfModTilt::names[ZfModTilt::NSTATE] = {"env2", "base", "lp_piv", "r_smooth", "mean_rdb"};
