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

  // sig@(d) accessors: registers (d = 0 is the current sample, h[d-1] the d-th previous one) or a contiguous LDS row
  struct RegHist { const float* h; float cur; ZF_FN float operator()(int d) const { return d == 0 ? cur : h[d - 1]; } };
  struct RowHist { const float* p; ZF_FN float operator()(int d) const { return p[-d]; } };

  template <class H> ZF_FN static float sg11(const H& at) {                                            // :51
    return (-36.0f * at(20) + 9.0f * at(19) + 44.0f * at(18) + 69.0f * at(17) + 84.0f * at(16) +
            89.0f * at(15) + 84.0f * at(14) + 69.0f * at(13) + 44.0f * at(12) + 9.0f * at(11) -
            36.0f * at(10)) / 429.0f;
  }
  template <class H> ZF_FN static float sg15(const H& at) {                                            // :53
    return (-78.0f * at(22) - 13.0f * at(21) + 42.0f * at(20) + 87.0f * at(19) + 122.0f * at(18) +
            147.0f * at(17) + 162.0f * at(16) + 167.0f * at(15) + 162.0f * at(14) + 147.0f * at(13) +
            122.0f * at(12) + 87.0f * at(11) + 42.0f * at(10) - 13.0f * at(9) - 78.0f * at(8)) / 1105.0f;
  }
  template <class H> ZF_FN static float sg21(const H& at) {                                            // :55
    return (-171.0f * at(25) - 76.0f * at(24) + 9.0f * at(23) + 84.0f * at(22) + 149.0f * at(21) +
            204.0f * at(20) + 249.0f * at(19) + 284.0f * at(18) + 309.0f * at(17) + 324.0f * at(16) +
            329.0f * at(15) + 324.0f * at(14) + 309.0f * at(13) + 284.0f * at(12) + 249.0f * at(11) +
            204.0f * at(10) + 149.0f * at(9) + 84.0f * at(8) + 9.0f * at(7) - 76.0f * at(6) -
            171.0f * at(5)) / 3059.0f;
  }
  template <class H> ZF_FN static float sg31(const H& at) {                                            // :57
    return (-406.0f * at(30) - 261.0f * at(29) - 126.0f * at(28) - 1.0f * at(27) + 114.0f * at(26) +
            219.0f * at(25) + 314.0f * at(24) + 399.0f * at(23) + 474.0f * at(22) + 539.0f * at(21) +
            594.0f * at(20) + 639.0f * at(19) + 674.0f * at(18) + 699.0f * at(17) + 714.0f * at(16) +
            719.0f * at(15) + 714.0f * at(14) + 699.0f * at(13) + 674.0f * at(12) + 639.0f * at(11) +
            594.0f * at(10) + 539.0f * at(9) + 474.0f * at(8) + 399.0f * at(7) + 314.0f * at(6) +
            219.0f * at(5) + 114.0f * at(4) - 1.0f * at(3) - 126.0f * at(2) - 261.0f * at(1) -
            406.0f * at(0)) / 9889.0f;
  }
  template <class H> ZF_FN static float small_pred(int m, const H& at) { return m <= 0 ? sg11(at) : (m == 1 ? sg15(at) : sg21(at)); }
  template <class H> ZF_FN static float large_pred(int m, const H& at) { return m <= 0 ? sg15(at) : (m == 1 ? sg21(at) : sg31(at)); }

  // ---- the sample function, split where the .dsp's recursions are (so the wave kernel can run the feed-forward parts
  //      with one lane per FRAME and only the recursions serially) ----------------------------------------------------
  struct Pred { float xC_L, xC_R, pred_L, pred_R, e_norm; };
  template <class H> ZF_FN static Pred predict(const Ctl& c, const H& aL, const H& aR) {          // :79-93 (feed-forward)
    return predict_with(c, small_pred(c.mode, aL), small_pred(c.mode, aR), large_pred(c.mode, aL), large_pred(c.mode, aR), aL, aR);
  }
  template <class H> ZF_FN static Pred predict_with(const Ctl& c, float small_L, float small_R, float large_L, float large_R,
                                                     const H& aL, const H& aR) {
    Pred q;
    q.xC_L = aL(15); q.xC_R = aR(15);
    const float eA = zf_max(fabsf(q.xC_L - small_L), fabsf(q.xC_R - small_R)) / (zf_max(fabsf(small_L), fabsf(small_R)) + 1e-6f);
    const float eB = zf_max(fabsf(q.xC_L - large_L), fabsf(q.xC_R - large_R)) / (zf_max(fabsf(large_L), fabsf(large_R)) + 1e-6f);
    const bool useA = eA <= eB;
    q.pred_L = useA ? small_L : large_L; q.pred_R = useA ? small_R : large_R; q.e_norm = useA ? eA : eB;
    return q;
  }
  // predict() where every lane of the wavefront serves ONE instance (the wave kernels: lane = frame), so the Mode is the same in
  // all of them: one scalar branch, and behind it the four smoothing sums of the two channels stand in ONE basic block -- four
  // independent chains of dependent additions (the source's left-to-right order is kept inside each) that the scheduler
  // interleaves, where the per-call mode tests of predict() leave it one chain at a time (role clock of the four-wavefront
  // kernel, round 4: the feed-forward wavefronts went from 5150 to 3950 cycles per 64-frame tick).
  template <int M, class H> ZF_FN static Pred predict_mode(const Ctl& c, const H& aL, const H& aR) {
    if constexpr (M <= 0) return predict_with(c, sg11(aL), sg11(aR), sg15(aL), sg15(aR), aL, aR);
    else if constexpr (M == 1) return predict_with(c, sg15(aL), sg15(aR), sg21(aL), sg21(aR), aL, aR);
    else return predict_with(c, sg21(aL), sg21(aR), sg31(aL), sg31(aR), aL, aR);
  }
  template <class H> ZF_FN static Pred predict_uniform(const Ctl& c, const H& aL, const H& aR) {
    const int m = __builtin_amdgcn_readfirstlane(c.mode);
    if (m <= 0) return predict_mode<0>(c, aL, aR);
    if (m == 1) return predict_mode<1>(c, aL, aR);
    return predict_mode<2>(c, aL, aR);
  }
  // recursion 1: hpf -> env -> base (:63-75); uL = L - L@1
  ZF_FN static void detect(float* st, const Ctl& c, float uL, float uR, float& env, float& base) {
    const float hpL = c.a * uL + c.a * st[S_HPL];
    const float hpR = c.a * uR + c.a * st[S_HPR];
    st[S_HPL] = hpL; st[S_HPR] = hpR;
    const float ehf = zf_max(fabsf(hpL), fabsf(hpR));
    env = zf_max(st[S_ENV] * c.env_rel, ehf);
    st[S_ENV] = env;
    base = env * c.base_a + st[S_BASE] * c.one_m_base_a;
    st[S_BASE] = base;
  }
  // detect() with the inputs' product a * u already formed (the wave kernel forms it frame-parallel); the two channels'
  // multiply and add are written as 2-vectors so that they issue as one packed instruction each (same IEEE operations)
  ZF_FN static void detect_scaled(float* st, const Ctl& c, float vL, float vR, float& env, float& base) {
#ifdef ZF_NO_PK
    struct { float x, y; } hp = {vL + c.a * st[S_HPL], vR + c.a * st[S_HPR]};
#else
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {vL, vR}, hp0 = {st[S_HPL], st[S_HPR]}, a2 = {c.a, c.a};
    const f2 hp = v + a2 * hp0;
#endif
    st[S_HPL] = hp.x; st[S_HPR] = hp.y;
    const float ehf = zf_max(fabsf(hp.x), fabsf(hp.y));
    env = zf_max(st[S_ENV] * c.env_rel, ehf);
    st[S_ENV] = env;
    base = env * c.base_a + st[S_BASE] * c.one_m_base_a;
    st[S_BASE] = base;
  }
  // detect_scaled() split where its own feed-forward part is, for the wave kernel (same IEEE operations, so the same bits):
  //   env_step   the HPFs and the envelope: env = max(env * rel, |hpL|, |hpR|) as ONE three-way maximum (max is exact and
  //              associative: v_max3_f32 with |.| modifiers gives the bits of max(env * rel, max(|hpL|, |hpR|)));
  //   (the product env * base_a of every frame is formed frame-parallel by the caller)
  //   base_step  base = eb + base * (1 - base_a).
  ZF_FN static float env_step(float* st, const Ctl& c, float vL, float vR) {
#ifdef ZF_NO_PK
    struct { float x, y; } hp = {vL + c.a * st[S_HPL], vR + c.a * st[S_HPR]};
#else
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {vL, vR}, hp0 = {st[S_HPL], st[S_HPR]}, a2 = {c.a, c.a};
    const f2 hp = v + a2 * hp0;
#endif
    st[S_HPL] = hp.x; st[S_HPR] = hp.y;
    const float env = __builtin_fmaxf(__builtin_fmaxf(st[S_ENV] * c.env_rel, fabsf(hp.x)), fabsf(hp.y));
    st[S_ENV] = env;
    return env;
  }
  ZF_FN static float base_step(float* st, const Ctl& c, float env_times_base_a) {
    const float base = env_times_base_a + st[S_BASE] * c.one_m_base_a;
    st[S_BASE] = base;
    return base;
  }
  // Recursion 1 over a whole 64-frame chunk, software-pipelined by hand (round 4): its three recurrences -- the HPF pair, the
  // envelope, the baseline -- each close over TWO dependent operations per frame, and written frame by frame they form one chain of
  // five dependent VALU operations whose latency (SQ counters: the serial wave issues 35 % of its cycles) is what a frame costs.
  // Here iteration n runs the HPFs of frame n, the envelope of frame n - 1 and the baseline of frame n - 2: three independent
  // instruction streams side by side. Same IEEE operations on the same operands as detect_scaled(), frame for frame.
  ZF_FN static void detect_chunk64(float* st, const Ctl& c, const float* vL, const float* vR, float* env_base_out) {
    float hpl = st[S_HPL], hpr = st[S_HPR], env = st[S_ENV], base = st[S_BASE];
#pragma unroll
    for (int n = 0; n < 64 + 2; ++n) {
      if (n >= 2) {                                             // baseline of frame n - 2 (env still holds that frame's envelope)
        base = env * c.base_a + base * c.one_m_base_a;
        env_base_out[2 * (n - 2) + 1] = base;
      }
      if (n >= 1 && n <= 64) {                                  // envelope of frame n - 1 (hp still holds that frame's HPF outputs)
        env = __builtin_fmaxf(__builtin_fmaxf(env * c.env_rel, fabsf(hpl)), fabsf(hpr));
        env_base_out[2 * (n - 1)] = env;
      }
      if (n < 64) {                                             // HPFs of frame n
        hpl = vL[n] + c.a * hpl;
        hpr = vR[n] + c.a * hpr;
      }
    }
    st[S_HPL] = hpl; st[S_HPR] = hpr; st[S_ENV] = env; st[S_BASE] = base;
  }
  ZF_FN static float trigger(const Ctl& c, float env, float base, float e_norm) {                // :77,95 (feed-forward)
    const float ratio = env / (base + 1e-12f);
    return (float)((int)(ratio > c.ratio_thr) * (int)(e_norm > c.err_thr));
  }
  ZF_FN static float hold_step(float* st, const Ctl& c, float trig) {                             // recursion 2 (:98-99)
    const float hold = zf_max(st[S_HOLD] * c.relHold, trig);
    st[S_HOLD] = hold;
    return hold;
  }
  ZF_FN static void mixdown(const Ctl& c, const Pred& q, float hold, float& oL, float& oR) {      // :100-114 (feed-forward)
    const bool active = hold > 1e-3f;
    const float mix_base = active ? zf_min(zf_max((q.e_norm - c.err_thr) / c.range_eps, 0.0f), 1.0f) : 0.0f;
    const float mix = mix_base * c.mix_max;
    const float outL = q.xC_L * (1.0f - mix) + q.pred_L * mix;
    const float outR = q.xC_R * (1.0f - mix) + q.pred_R * mix;
    oL = c.monitor ? outL - q.xC_L : outL;
    oR = c.monitor ? outR - q.xC_R : outR;
  }

  ZF_FN static void frame(float* st, const Ctl& c, float* x) {
    const float L = x[0], R = x[1];
    float* hL = st + S_HL;
    float* hR = st + S_HR;
    const RegHist aL{hL, L}, aR{hR, R};
    float env, base;
    detect(st, c, L - hL[0], R - hR[0], env, base);
    const Pred q = predict(c, aL, aR);
    const float hold = hold_step(st, c, trigger(c, env, base, q.e_norm));
    mixdown(c, q, hold, x[0], x[1]);
#pragma unroll
    for (int d = HIST - 1; d > 0; --d) { hL[d] = hL[d - 1]; hR[d] = hR[d - 1]; }
    hL[0] = L; hR[0] = R;
  }
};

const char* const ZfClickBeGone::names[ZfClickBeGone::NSTATE] = {
    "hpL", "hpR", "env", "base", "hold",
    "L@1", "L@2", "L@3", "L@4", "L@5", "L@6", "L@7", "L@8", "L@9", "L@10", "L@11", "L@12", "L@13", "L@14", "L@15",
    "L@16", "L@17", "L@18", "L@19", "L@20", "L@21", "L@22", "L@23", "L@24", "L@25", "L@26", "L@27", "L@28", "L@29", "L@30",
    "R@1", "R@2", "R@3", "R@4", "R@5", "R@6", "R@7", "R@8", "R@9", "R@10", "R@11", "R@12", "R@13", "R@14", "R@15",
    "R@16", "R@17", "R@18", "R@19", "R@20", "R@21", "R@22", "R@23", "R@24", "R@25", "R@26", "R@27", "R@28", "R@29", "R@30"};
