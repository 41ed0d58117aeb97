// clickbegone.hip.h -- Restoration/ClickBeGoneSG, restated from "Click-Be-Gone (SG).dsp" (reference:
// plugins/Restoration/ClickBeGoneSG/src/Click-Be-Gone (SG).dsp; line numbers below refer to it).
// f32 throughout (Faust -single). Expressions keep the source's left-to-right evaluation order; no FMA contraction.
#pragma once

#include "faust_lane.hip.h"

struct ZfClickBeGone {
  static constexpr int NCH = 2;
  static constexpr int NPARAM = 5;       // Amount, Sensitivity, HPF, Mode, Monitor (:11-15, declaration order)
  static constexpr int HIST = 30;        // x@1 .. x@30 (:51-57)
  // state: hpL hpR env base hold | L@1..L@30 | R@1..R@30
  static constexpr int S_HPL = 0, S_HPR = 1, S_ENV = 2, S_BASE = 3, S_HOLD = 4, S_HL = 5, S_HR = 5 + HIST;
  static constexpr int NSTATE = 5 + 2 * HIST;
  static const char* const names[NSTATE];

  struct Ctl {
    float a, env_rel, base_a, one_m_base_a, ratio_thr, err_thr, mix_max, relHold, range_eps;
    int mode, monitor;
  };

  ZF_FN static float sel3(int m, float a, float b, float c) { return m <= 0 ? a : (m == 1 ? b : c); }   // ba.selectn(3, mode, ..)

  ZF_FN static Ctl control(const float* p, float SR) {
    Ctl c;
    const float amount = p[0] / 100.0f, sens = p[1] / 100.0f, hpf_hz = p[2];                    // :11-13
    c.mode = (int)p[3];
    c.monitor = (int)p[4];
    const float eps = 1e-12f;
    const float ratio_thr0 = 6.0f - 4.0f * sens, err_thr0 = 0.25f - 0.17f * sens;               // :20-21
    const float ratio_mul = sel3(c.mode, 1.12f, 1.00f, 0.92f), err_mul = sel3(c.mode, 1.18f, 1.00f, 0.90f);
    const float mix_mul = sel3(c.mode, 0.85f, 1.00f, 1.08f), hold_mul = sel3(c.mode, 0.75f, 1.00f, 1.35f);
    const float env_rel_ms0 = 30.0f - 20.0f * sens, base_ms0 = 300.0f - 180.0f * sens;          // :28-29
    const float base_mul = sel3(c.mode, 0.85f, 1.00f, 1.10f), env_mul = sel3(c.mode, 0.85f, 1.00f, 1.10f);
    const float env_rel_ms = env_rel_ms0 * env_mul, base_ms = base_ms0 * base_mul;
    c.ratio_thr = ratio_thr0 * ratio_mul;
    c.err_thr = err_thr0 * err_mul;
    const float mix_max0 = 0.60f + 0.32f * amount;                                              // :39-40
    c.mix_max = zf_min(mix_max0 * mix_mul, 0.96f);
    const float holdN_base = 8.0f + amount * 32.0f;                                             // :42-43
    const float holdN = zf_max(holdN_base * hold_mul, 4.0f);
    c.env_rel = zf_exp(-1000.0f / (SR * env_rel_ms));                                           // :45-46
    c.base_a = 1.0f - zf_exp(-1000.0f / (SR * base_ms));
    c.one_m_base_a = 1.0f - c.base_a;
    c.a = zf_exp((-6.2831855f * hpf_hz) / SR);                                                  // :48
    c.relHold = zf_exp(-6.9077554f / (holdN + eps));                                            // :93-94, log(1e-3)
    c.range_eps = c.err_thr * 3.0f + eps;                                                       // :100-101
    return c;
  }

  // sig@(d): d = 0 is the current sample, h[d-1] the d-th previous one
  ZF_FN static float at(const float* h, float cur, int d) { return d == 0 ? cur : h[d - 1]; }

  ZF_FN static float sg11(const float* h, float x) {                                            // :51
    return (-36.0f * at(h, x, 20) + 9.0f * at(h, x, 19) + 44.0f * at(h, x, 18) + 69.0f * at(h, x, 17) + 84.0f * at(h, x, 16) +
            89.0f * at(h, x, 15) + 84.0f * at(h, x, 14) + 69.0f * at(h, x, 13) + 44.0f * at(h, x, 12) + 9.0f * at(h, x, 11) -
            36.0f * at(h, x, 10)) / 429.0f;
  }
  ZF_FN static float sg15(const float* h, float x) {                                            // :53
    return (-78.0f * at(h, x, 22) - 13.0f * at(h, x, 21) + 42.0f * at(h, x, 20) + 87.0f * at(h, x, 19) + 122.0f * at(h, x, 18) +
            147.0f * at(h, x, 17) + 162.0f * at(h, x, 16) + 167.0f * at(h, x, 15) + 162.0f * at(h, x, 14) + 147.0f * at(h, x, 13) +
            122.0f * at(h, x, 12) + 87.0f * at(h, x, 11) + 42.0f * at(h, x, 10) - 13.0f * at(h, x, 9) - 78.0f * at(h, x, 8)) / 1105.0f;
  }
  ZF_FN static float sg21(const float* h, float x) {                                            // :55
    return (-171.0f * at(h, x, 25) - 76.0f * at(h, x, 24) + 9.0f * at(h, x, 23) + 84.0f * at(h, x, 22) + 149.0f * at(h, x, 21) +
            204.0f * at(h, x, 20) + 249.0f * at(h, x, 19) + 284.0f * at(h, x, 18) + 309.0f * at(h, x, 17) + 324.0f * at(h, x, 16) +
            329.0f * at(h, x, 15) + 324.0f * at(h, x, 14) + 309.0f * at(h, x, 13) + 284.0f * at(h, x, 12) + 249.0f * at(h, x, 11) +
            204.0f * at(h, x, 10) + 149.0f * at(h, x, 9) + 84.0f * at(h, x, 8) + 9.0f * at(h, x, 7) - 76.0f * at(h, x, 6) -
            171.0f * at(h, x, 5)) / 3059.0f;
  }
  ZF_FN static float sg31(const float* h, float x) {                                            // :57
    return (-406.0f * at(h, x, 30) - 261.0f * at(h, x, 29) - 126.0f * at(h, x, 28) - 1.0f * at(h, x, 27) + 114.0f * at(h, x, 26) +
            219.0f * at(h, x, 25) + 314.0f * at(h, x, 24) + 399.0f * at(h, x, 23) + 474.0f * at(h, x, 22) + 539.0f * at(h, x, 21) +
            594.0f * at(h, x, 20) + 639.0f * at(h, x, 19) + 674.0f * at(h, x, 18) + 699.0f * at(h, x, 17) + 714.0f * at(h, x, 16) +
            719.0f * at(h, x, 15) + 714.0f * at(h, x, 14) + 699.0f * at(h, x, 13) + 674.0f * at(h, x, 12) + 639.0f * at(h, x, 11) +
            594.0f * at(h, x, 10) + 539.0f * at(h, x, 9) + 474.0f * at(h, x, 8) + 399.0f * at(h, x, 7) + 314.0f * at(h, x, 6) +
            219.0f * at(h, x, 5) + 114.0f * at(h, x, 4) - 1.0f * at(h, x, 3) - 126.0f * at(h, x, 2) - 261.0f * at(h, x, 1) -
            406.0f * at(h, x, 0)) / 9889.0f;
  }
  ZF_FN static float small_pred(int m, const float* h, float x) { return m <= 0 ? sg11(h, x) : (m == 1 ? sg15(h, x) : sg21(h, x)); }
  ZF_FN static float large_pred(int m, const float* h, float x) { return m <= 0 ? sg15(h, x) : (m == 1 ? sg21(h, x) : sg31(h, x)); }

  ZF_FN static void frame(float* st, const Ctl& c, float* x) {
    const float L = x[0], R = x[1];
    float* hL = st + S_HL;
    float* hR = st + S_HR;
    // hpf_jsfx(a) = (_ <: (_, _@1) : -) : *(a) : (+ ~ *(a))                                     :63
    const float hpL = c.a * (L - hL[0]) + c.a * st[S_HPL];
    const float hpR = c.a * (R - hR[0]) + c.a * st[S_HPR];
    st[S_HPL] = hpL; st[S_HPR] = hpR;
    const float ehf = zf_max(fabsf(hpL), fabsf(hpR));                                           // :72
    const float env = zf_max(st[S_ENV] * c.env_rel, ehf);                                       // :73  max ~ *(env_rel)
    st[S_ENV] = env;
    const float base = env * c.base_a + st[S_BASE] * c.one_m_base_a;                            // :74-75
    st[S_BASE] = base;
    const float ratio = env / (base + 1e-12f);                                                  // :77
    const float xC_L = at(hL, L, 15), xC_R = at(hR, R, 15);                                     // :79-80
    const float small_L = small_pred(c.mode, hL, L), small_R = small_pred(c.mode, hR, R);       // :82-85
    const float large_L = large_pred(c.mode, hL, L), large_R = large_pred(c.mode, hR, R);
    const float eA = zf_max(fabsf(xC_L - small_L), fabsf(xC_R - small_R)) / (zf_max(fabsf(small_L), fabsf(small_R)) + 1e-6f);
    const float eB = zf_max(fabsf(xC_L - large_L), fabsf(xC_R - large_R)) / (zf_max(fabsf(large_L), fabsf(large_R)) + 1e-6f);
    const bool useA = eA <= eB;                                                                 // :90-93
    const float pred_L = useA ? small_L : large_L, pred_R = useA ? small_R : large_R, e_norm = useA ? eA : eB;
    const float trig = (float)((int)(ratio > c.ratio_thr) * (int)(e_norm > c.err_thr));         // :95
    const float hold = zf_max(st[S_HOLD] * c.relHold, trig);                                    // :98-99
    st[S_HOLD] = hold;
    const bool active = hold > 1e-3f;                                                           // :100
    const float mix_base = active ? zf_min(zf_max((e_norm - c.err_thr) / c.range_eps, 0.0f), 1.0f) : 0.0f;   // :103-104
    const float mix = mix_base * c.mix_max;
    const float outL = xC_L * (1.0f - mix) + pred_L * mix;                                      // :107-108
    const float outR = xC_R * (1.0f - mix) + pred_R * mix;
    x[0] = c.monitor ? outL - xC_L : outL;                                                      // :110-114
    x[1] = c.monitor ? outR - xC_R : outR;
#pragma unroll
    for (int d = HIST - 1; d > 0; --d) { hL[d] = hL[d - 1]; hR[d] = hR[d - 1]; }
    hL[0] = L; hR[0] = R;
  }
};

#define ZF_N(s) s
const char* const ZfClickBeGone::names[ZfClickBeGone::NSTATE] = {
    "hpL", "hpR", "env", "base", "hold",
    "L@1", "L@2", "L@3", "L@4", "L@5", "L@6", "L@7", "L@8", "L@9", "L@10", "L@11", "L@12", "L@13", "L@14", "L@15",
    "L@16", "L@17", "L@18", "L@19", "L@20", "L@21", "L@22", "L@23", "L@24", "L@25", "L@26", "L@27", "L@28", "L@29", "L@30",
    "R@1", "R@2", "R@3", "R@4", "R@5", "R@6", "R@7", "R@8", "R@9", "R@10", "R@11", "R@12", "R@13", "R@14", "R@15",
    "R@16", "R@17", "R@18", "R@19", "R@20", "R@21", "R@22", "R@23", "R@24", "R@25", "R@26", "R@27", "R@28", "R@29", "R@30"};
#undef ZF_N
