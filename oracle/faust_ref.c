/* oracle/faust_ref.c -- TEST INFRASTRUCTURE ONLY (used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline).
 *
 * CPU restatement of the Faust leaves' mydsp::compute(), written in the shape of Faust-generated code:
 * a state struct with fRec / fVec ring buffers and an IOTA counter, "fSlow" values computed once per compute() call from
 * the UI zones, then the sample loop. f32 throughout (Faust -single, src/faust_support_min.h:4-6).
 *
 * PARITY UNPINNED: the Faust compiler and stdfaust.lib (pinned only as FAUST_VERSION 2.81.2 for the Windows installer,
 * .github/workflows/release.yml:4) are not in the reference tree and no reference test holds an output of these leaves.
 * The library definitions used here are the published ones: si.smooth(s) = *(1-s) : + ~ *(s); ba.db2linear(x) = pow(10,
 * x/20); ba.if / ba.selectn = select2 chains; x@(n) = n-sample delay with zero history; max ~ *(r) = peak hold with decay.
 * Control-rate and per-sample transcendental functions are evaluated in double and rounded once to float (expf/log10f/powf
 * of the platform libm differ from that by at most the last bit in rare cases).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/faust_ref.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float f_exp(float x) { return (float)exp((double)x); }
static float f_log10(float x) { return (float)log10((double)x); }
static float f_pow(float x, float y) { return (float)pow((double)x, (double)y); }
static float f_min(float a, float b) { return fminf(a, b); }
static float f_max(float a, float b) { return fmaxf(a, b); }
static float ma_SR(int sr) { return f_min(192000.0f, f_max(1.0f, (float)sr)); }

/* ---------------------------------------------------------------------------------------------------------------------
 * Restoration/ClickBeGoneSG -- plugins/Restoration/ClickBeGoneSG/src/Click-Be-Gone (SG).dsp
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int sr;
  int IOTA;
  float fVecL[32], fVecR[32]; /* input delay lines, x@(0..30) */
  float fRecHpL[2], fRecHpR[2], fRecEnv[2], fRecBase[2], fRecHold[2];
} cbg_t;

static const int SG11_D0 = 10, SG15_D0 = 8, SG21_D0 = 5, SG31_D0 = 0;
/* coefficients listed from the LARGEST delay to the smallest, the order the .dsp writes its sums in (:51-57) */
static const float SG11[11] = {-36, 9, 44, 69, 84, 89, 84, 69, 44, 9, -36};
static const float SG15[15] = {-78, -13, 42, 87, 122, 147, 162, 167, 162, 147, 122, 87, 42, -13, -78};
static const float SG21[21] = {-171, -76, 9, 84, 149, 204, 249, 284, 309, 324, 329, 324, 309, 284, 249, 204, 149, 84, 9, -76, -171};
static const float SG31[31] = {-406, -261, -126, -1, 114, 219, 314, 399, 474, 539, 594, 639, 674, 699, 714, 719,
                               714,  699,  674,  639, 594, 539, 474, 399, 314, 219, 114, -1,  -126, -261, -406};

static float sg_pred(const float* vec, int iota, const float* coef, int n, int d0, float norm) {
  /* term k multiplies x@(d0 + n-1-k); terms are added left to right; a negative coefficient after the first term is the
     source's "- c * x" (same value as "+ (-c) * x" in IEEE arithmetic) */
  float acc = 0.0f;
  for (int k = 0; k < n; ++k) {
    const float x = vec[(iota - (d0 + n - 1 - k)) & 31];
    const float term = coef[k] * x;
    acc = (k == 0) ? term : acc + term;
  }
  return acc / norm;
}
static float pred_by_index(int which, const float* vec, int iota) {
  switch (which) {
    case 0: return sg_pred(vec, iota, SG11, 11, SG11_D0, 429.0f);
    case 1: return sg_pred(vec, iota, SG15, 15, SG15_D0, 1105.0f);
    case 2: return sg_pred(vec, iota, SG21, 21, SG21_D0, 3059.0f);
    default: return sg_pred(vec, iota, SG31, 31, SG31_D0, 9889.0f);
  }
}
static float selectn3(int i, float a, float b, float c) { return i <= 0 ? a : (i == 1 ? b : c); }

static void cbg_compute(cbg_t* d, const float* zone, int count, float** in, float** out) {
  const float SR = ma_SR(d->sr);
  const float amount = zone[0] / 100.0f;
  const float sensitivity = zone[1] / 100.0f;
  const float hpf_hz = zone[2];
  const int mode = (int)zone[3];
  const int monitor = (int)zone[4];
  const float eps = 1e-12f;
  const float ratio_thr = (6.0f - 4.0f * sensitivity) * selectn3(mode, 1.12f, 1.00f, 0.92f);
  const float err_thr = (0.25f - 0.17f * sensitivity) * selectn3(mode, 1.18f, 1.00f, 0.90f);
  const float mix_mul = selectn3(mode, 0.85f, 1.00f, 1.08f);
  const float hold_mul = selectn3(mode, 0.75f, 1.00f, 1.35f);
  const float env_rel_ms = (30.0f - 20.0f * sensitivity) * selectn3(mode, 0.85f, 1.00f, 1.10f);
  const float base_ms = (300.0f - 180.0f * sensitivity) * selectn3(mode, 0.85f, 1.00f, 1.10f);
  const float mix_max = f_min((0.60f + 0.32f * amount) * mix_mul, 0.96f);
  const float holdN = f_max((8.0f + amount * 32.0f) * hold_mul, 4.0f);
  const float env_rel = f_exp(-1000.0f / (SR * env_rel_ms));
  const float base_a = 1.0f - f_exp(-1000.0f / (SR * base_ms));
  const float a = f_exp((-6.2831855f * hpf_hz) / SR);
  const float T = 1e-3f;
  const float relHold = f_exp(-6.9077554f / (holdN + eps));
  const float range = err_thr * 3.0f;
  const int small_idx = mode <= 0 ? 0 : (mode == 1 ? 1 : 2);
  const int large_idx = small_idx + 1;
  for (int i = 0; i < count; ++i) {
    const float L = in[0][i], R = in[1][i];
    const int io = d->IOTA;
    d->fVecL[io & 31] = L;
    d->fVecR[io & 31] = R;
    d->fRecHpL[0] = a * (L - d->fVecL[(io - 1) & 31]) + a * d->fRecHpL[1];
    d->fRecHpR[0] = a * (R - d->fVecR[(io - 1) & 31]) + a * d->fRecHpR[1];
    const float ehf = f_max(fabsf(d->fRecHpL[0]), fabsf(d->fRecHpR[0]));
    d->fRecEnv[0] = f_max(d->fRecEnv[1] * env_rel, ehf);
    d->fRecBase[0] = d->fRecEnv[0] * base_a + d->fRecBase[1] * (1.0f - base_a);
    const float ratio = d->fRecEnv[0] / (d->fRecBase[0] + eps);
    const float xC_L = d->fVecL[(io - 15) & 31], xC_R = d->fVecR[(io - 15) & 31];
    const float small_L = pred_by_index(small_idx, d->fVecL, io), small_R = pred_by_index(small_idx, d->fVecR, io);
    const float large_L = pred_by_index(large_idx, d->fVecL, io), large_R = pred_by_index(large_idx, d->fVecR, io);
    const float eA = f_max(fabsf(xC_L - small_L), fabsf(xC_R - small_R)) / (f_max(fabsf(small_L), fabsf(small_R)) + 1e-6f);
    const float eB = f_max(fabsf(xC_L - large_L), fabsf(xC_R - large_R)) / (f_max(fabsf(large_L), fabsf(large_R)) + 1e-6f);
    const int useA = eA <= eB;
    const float pred_L = useA ? small_L : large_L, pred_R = useA ? small_R : large_R, e_norm = useA ? eA : eB;
    const int trig = (ratio > ratio_thr) * (e_norm > err_thr);
    d->fRecHold[0] = f_max(d->fRecHold[1] * relHold, (float)trig);
    const int active = d->fRecHold[0] > T;
    float mix_base = 0.0f;
    if (active) mix_base = f_min(f_max((e_norm - err_thr) / (range + eps), 0.0f), 1.0f);
    const float mix = mix_base * mix_max;
    const float outL = xC_L * (1.0f - mix) + pred_L * mix;
    const float outR = xC_R * (1.0f - mix) + pred_R * mix;
    out[0][i] = monitor ? outL - xC_L : outL;
    out[1][i] = monitor ? outR - xC_R : outR;
    d->fRecHpL[1] = d->fRecHpL[0]; d->fRecHpR[1] = d->fRecHpR[0];
    d->fRecEnv[1] = d->fRecEnv[0]; d->fRecBase[1] = d->fRecBase[0]; d->fRecHold[1] = d->fRecHold[0];
    d->IOTA = io + 1;
  }
}
/* state in the order the device module names it: hpL hpR env base hold L@1..L@30 R@1..R@30 */
static int cbg_state(const cbg_t* d, float* o) {
  int n = 0;
  o[n++] = d->fRecHpL[1]; o[n++] = d->fRecHpR[1]; o[n++] = d->fRecEnv[1]; o[n++] = d->fRecBase[1]; o[n++] = d->fRecHold[1];
  for (int k = 1; k <= 30; ++k) o[n++] = d->fVecL[(d->IOTA - k) & 31];
  for (int k = 1; k <= 30; ++k) o[n++] = d->fVecR[(d->IOTA - k) & 31];
  return n;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Dynamics/ModTilt -- plugins/Dynamics/ModTilt/src/ModTilt.dsp
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int sr;
  float fRecEnv2[2], fRecBase[2], fRecPiv[2], fRecRatio[2], fRecTrim[2];
} mt_t;

static float mt_a_from_hz(float hz, float SR) { return 1.0f - f_exp((-6.2831855f * f_max(hz, 0.001f)) / SR); }

static void mt_compute(mt_t* d, const float* zone, int count, float** in, float** out) {
  const float SR = ma_SR(d->sr);
  const float tilt_db = zone[0], pivotHz = zone[1], mix = zone[2];
  const float a_env = mt_a_from_hz(25.0f, SR), a_base = mt_a_from_hz(1.0f, SR), a_piv = mt_a_from_hz(pivotHz, SR);
  const float a_ratio = 0.05f, a_trim = mt_a_from_hz(0.2f, SR);
  const float depth = 0.75f;
  const float g_hi = f_pow(10.0f, (tilt_db * 0.5f) / 20.0f);
  const float g_lo = f_pow(10.0f, ((0.0f - tilt_db) * 0.5f) / 20.0f);
  for (int i = 0; i < count; ++i) {
    const float xL = in[0][i], xR = in[1][i];
    const float x = 0.5f * (xL + xR);
    d->fRecEnv2[0] = (x * x) * a_env + d->fRecEnv2[1] * (1.0f - a_env);
    const float env = sqrtf(f_max(d->fRecEnv2[0], 0.0f));
    d->fRecBase[0] = env * a_base + d->fRecBase[1] * (1.0f - a_base);
    const float base = d->fRecBase[0];
    const float m = env - base;
    d->fRecPiv[0] = m * a_piv + d->fRecPiv[1] * (1.0f - a_piv);
    const float m_lo = d->fRecPiv[0], m_hi = m - d->fRecPiv[0];
    const float m2_tilt = m_lo * g_lo + m_hi * g_hi;
    const float m2 = m * (1.0f - depth) + m2_tilt * depth;
    const float env_t = base + m2;
    const float env_tp = f_max(env_t, 0.05f * env);
    const float r0 = (env_tp + 1e-9f) / (env + 1e-9f);
    const float r0c = f_min(f_max(r0, 0.67f), 1.5f);
    d->fRecRatio[0] = (r0c - 1.0f) * a_ratio + d->fRecRatio[1] * (1.0f - a_ratio);
    const float r_s = 1.0f + d->fRecRatio[0];
    const float rdb = 20.0f * f_log10(f_max(r_s, 1e-12f));
    d->fRecTrim[0] = rdb * a_trim + d->fRecTrim[1] * (1.0f - a_trim);
    const float trim = f_pow(10.0f, (0.0f - d->fRecTrim[0]) / 20.0f);
    const float yL = xL * r_s, yR = xR * r_s;
    out[0][i] = (xL * (1.0f - mix) + yL * mix) * trim;
    out[1][i] = (xR * (1.0f - mix) + yR * mix) * trim;
    d->fRecEnv2[1] = d->fRecEnv2[0]; d->fRecBase[1] = d->fRecBase[0]; d->fRecPiv[1] = d->fRecPiv[0];
    d->fRecRatio[1] = d->fRecRatio[0]; d->fRecTrim[1] = d->fRecTrim[0];
  }
}
static int mt_state(const mt_t* d, float* o) {
  o[0] = d->fRecEnv2[1]; o[1] = d->fRecBase[1]; o[2] = d->fRecPiv[1]; o[3] = d->fRecRatio[1]; o[4] = d->fRecTrim[1];
  return 5;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * shared library pieces (published stdfaust.lib definitions)
 *   si.smooth(s)  = *(1 - s) : + ~ *(s)                      si.smoo = si.smooth(1 - 44.1/ma.SR)
 *   ba.db2linear(l) = pow(10, l/20)                           ba.tau2pole(t) = exp(-1/(t*ma.SR))
 *   si.onePoleSwitching(att, rel, x): y = (1-c)*x + c*y', c = x > y' ? tau2pole(att) : tau2pole(rel)
 *   an.amp_follower_ar(att, rel) = abs : si.onePoleSwitching(att, rel)
 *   fi.tf22t(b0,b1,b2,a1,a2): transposed direct form II: y = b0*x + s1'; s1 = (b1*x - a1*y) + s2'; s2 = b2*x - a2*y
 *   no.noise = (+(12345) ~ *(1103515245)) / 2147483647.0   (int32 wrap-around)
 * ------------------------------------------------------------------------------------------------------------------- */
static float f_expf(float x) { return expf(x); }          /* per-sample exp of GTS: platform expf, compared by tolerance */
static float smooth_step(float s, float x, float* y) { *y = x * (1.0f - s) + s * *y; return *y; }
static float db2linear(float l) { return f_pow(10.0f, l / 20.0f); }
typedef struct { float s1, s2; } tf22t_t;
static float tf22t(tf22t_t* z, float b0, float b1, float b2, float a1, float a2, float x) {
  const float y = b0 * x + z->s1;
  z->s1 = (b1 * x - a1 * y) + z->s2;
  z->s2 = b2 * x - a2 * y;
  return y;
}
static float one_pole_switching(float catt, float crel, float x, float* y) {
  const float c = (x > *y) ? catt : crel;
  *y = (1.0f - c) * x + c * *y;
  return *y;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Dynamics/GTS -- plugins/Dynamics/GTS/src/Gaussian Transient Shaper (GTS).dsp
 * The five UI values pass through si.smoo, so the Gaussian kernel (129 distinct taps, :24-38) is a per-sample signal.
 * ------------------------------------------------------------------------------------------------------------------- */
#define GTS_R 128
typedef struct {
  int sr;
  int IOTA;
  float fVec[2][512];            /* x@(0..256) per channel */
  float fRecSm[5];               /* si.smoo states: sigmaMs, attackDB, sustainDB, mix, outGain(lin) */
} gts_t;

static void gts_compute(gts_t* d, const float* zone, int count, float** in, float** out) {
  const float SR = ma_SR(d->sr);
  const float s = 1.0f - 44.1f / SR;
  const float target[5] = {zone[0], zone[1], zone[2], zone[3], db2linear(zone[4])};   /* :52-65 */
  float g[GTS_R + 1];
  for (int i = 0; i < count; ++i) {
    float sm[5];
    for (int k = 0; k < 5; ++k) sm[k] = smooth_step(s, target[k], &d->fRecSm[k]);
    const float sigmaSamples = f_max(0.25f, sm[0] * SR * 0.001f);                     /* :69-70 */
    for (int j = 0; j <= GTS_R; ++j) {                                                /* g(i) :27 */
      const float q = (float)j / sigmaSamples;
      g[j] = f_expf(-0.5f * (q * q));
    }
    float sumRest = 0.0f;
    for (int j = 1; j <= GTS_R; ++j) sumRest = (j == 1) ? g[1] : sumRest + g[j];      /* :31 */
    const float norm = 1.0f / (g[0] + 2.0f * sumRest + 1e-20f);                       /* :32 */
    const float aGain = db2linear(sm[1]), sGain = db2linear(sm[2]);                   /* :88-89 */
    const int io = d->IOTA;
    for (int c = 0; c < 2; ++c) {
      float* v = d->fVec[c];
      const float x = in[c][i];
      v[io & 511] = x;
      float sustain = 0.0f;                                                           /* fi.fir, taps in order k = 0..256 */
      for (int k = 0; k <= 2 * GTS_R; ++k) {
        const int off = k < GTS_R ? GTS_R - k : k - GTS_R;
        const float term = (norm * g[off]) * v[(io - k) & 511];
        sustain = (k == 0) ? term : sustain + term;
      }
      const float xAligned = v[(io - GTS_R) & 511];                                   /* :80 */
      const float attack = xAligned - sustain;
      const float shaped = aGain * attack + sGain * sustain;                          /* :91 */
      out[c][i] = ((sm[3] * shaped) + ((1.0f - sm[3]) * xAligned)) * sm[4];           /* :94 */
    }
    d->IOTA = io + 1;
  }
}
/* state: 5 smoothers, then L@1..L@256, R@1..R@256 */
static int gts_state(const gts_t* d, float* o) {
  int n = 0;
  for (int k = 0; k < 5; ++k) o[n++] = d->fRecSm[k];
  for (int c = 0; c < 2; ++c)
    for (int k = 1; k <= 256; ++k) o[n++] = d->fVec[c][(d->IOTA - k) & 511];
  return n;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Restoration/VAR -- plugins/Restoration/VAR/src/Vocal Air Recovery (VAR).dsp
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int sr;
  float smAmount, smSens;                      /* si.smoo (:93-94) */
  tf22t_t det[2], hf1[2], hf2[2], air;         /* biquads (:126-127,167-168,177-178; nL == nR is one shared signal) */
  float hfLvl;                                 /* :131 */
  float sm1[2], sm2[2], d1[2], d2[2];          /* two smoothing stages and their x', x'' (:137-148) */
  float env;                                   /* :160 */
  int32_t rnd;                                 /* no.noise */
} var_t;

typedef struct { float b0, b1, b2, a1, a2; } biq_t;
static float var_safe(float fc, float SR) { return f_min(fc, 0.45f * SR); }
static biq_t var_rbj(int kind, float fc, float Q, float SR) {     /* 0 HP, 1 LP, 2 BP constant skirt (:15-86) */
  biq_t c;
  const float f = var_safe(fc, SR), q = f_max(0.001f, Q);
  const float w0 = 6.2831855f * f / SR;
  const float cw = (float)cos((double)w0), sw = (float)sin((double)w0);
  const float alpha = sw / (2.0f * q);
  float bb0, bb1, bb2;
  if (kind == 0) { bb0 = (1.0f + cw) / 2.0f; bb1 = -(1.0f + cw); bb2 = (1.0f + cw) / 2.0f; }
  else if (kind == 1) { bb0 = (1.0f - cw) / 2.0f; bb1 = 1.0f - cw; bb2 = (1.0f - cw) / 2.0f; }
  else { bb0 = sw / 2.0f; bb1 = 0.0f; bb2 = -sw / 2.0f; }
  const float aa0 = 1.0f + alpha, aa1 = -2.0f * cw, aa2 = 1.0f - alpha;
  c.b0 = bb0 / aa0; c.b1 = bb1 / aa0; c.b2 = bb2 / aa0; c.a1 = aa1 / aa0; c.a2 = aa2 / aa0;
  return c;
}
static float var_smoothstep(float a, float b, float x) {
  const float u = f_min(1.0f, f_max(0.0f, (x - a) / (b - a)));
  return u * u * (3.0f - 2.0f * u);
}
static void var_compute(var_t* d, const float* zone, int count, float** in, float** out) {
  const float SR = ma_SR(d->sr);
  const float eps = 1e-12f;
  const float s = 1.0f - 44.1f / SR;
  const float amountT = zone[0] / 100.0f, sensT = zone[1] / 100.0f;
  const float floorLin = db2linear(zone[2]);
  const biq_t det = var_rbj(2, 9500.0f, 1.0f, SR), hf = var_rbj(0, 11500.0f, 0.707f, SR), airc = var_rbj(2, 16000.0f, 1.2f, SR);
  const float detSmooth_a = f_exp(-6.2831855f * var_safe(8500.0f, SR) / SR);
  const float hfLvl_a = f_exp(-1.0f / (SR * 0.14f));
  const float catt = f_exp(-1.0f / (0.0025f * SR)), crel = f_exp(-1.0f / (0.080f * SR));
  const float airBase = db2linear(-34.0f);
  for (int i = 0; i < count; ++i) {
    const float amount = smooth_step(s, amountT, &d->smAmount), sens = smooth_step(s, sensT, &d->smSens);
    const float maxExp_lin = db2linear(5.0f * amount);                               /* :103-104, per sample */
    const float airMix = 0.25f * amount;
    const float thrN = 0.18f - 0.13f * sens;
    float detv[2], curv[2], hfv[2];
    for (int c = 0; c < 2; ++c) {
      const float x = in[c][i];
      detv[c] = tf22t(&d->det[c], det.b0, det.b1, det.b2, det.a1, det.a2, x);
      d->sm1[c] = detv[c] * (1.0f - detSmooth_a) + detSmooth_a * d->sm1[c];          /* onePoleExp x2 (:137-138) */
      const float s1prev = d->d1[c], s2prev = d->d2[c];
      d->sm2[c] = d->sm1[c] * (1.0f - detSmooth_a) + detSmooth_a * d->sm2[c];
      const float s0 = d->sm2[c];
      const float lap = s0 - 2.0f * s1prev + s2prev;                                  /* :143-151 */
      const float denom = fabsf(s0) + 2.0f * fabsf(s1prev) + fabsf(s2prev) + eps;
      curv[c] = fabsf(lap) / denom;
      d->d2[c] = s1prev; d->d1[c] = s0;
      hfv[c] = tf22t(&d->hf2[c], hf.b0, hf.b1, hf.b2, hf.a1, hf.a2, tf22t(&d->hf1[c], hf.b0, hf.b1, hf.b2, hf.a1, hf.a2, x));
    }
    const float hfAbs = 0.5f * (fabsf(detv[0]) + fabsf(detv[1]));                     /* :129-133 */
    d->hfLvl = hfAbs * (1.0f - hfLvl_a) + hfLvl_a * d->hfLvl;
    const float gate = var_smoothstep(1.0f, 2.0f, d->hfLvl / (floorLin + eps));
    const float curvN = 0.5f * (curv[0] + curv[1]);
    const float env = one_pole_switching(catt, crel, curvN, &d->env);                /* :160 */
    const float u = f_max(0.0f, env / thrN - 1.0f);                                   /* :163-165 */
    const float t = (u / (1.0f + u)) * gate;
    const float t2 = f_pow(f_max(eps, t), 1.8f);
    const float g = 1.0f + t * (maxExp_lin - 1.0f);                                   /* :168 */
    d->rnd = (int32_t)((uint32_t)d->rnd * 1103515245u + 12345u);                      /* no.noise */
    const float nz = (float)d->rnd / 2147483647.0f;
    const float air = tf22t(&d->air, airc.b0, airc.b1, airc.b2, airc.a1, airc.a2, nz);
    const float airGain = (t2 * airBase) * airMix;                                    /* :183-185 */
    for (int c = 0; c < 2; ++c)
      out[c][i] = (in[c][i] + hfv[c] * (g - 1.0f) + air * airGain) * 1.0f;            /* :187-188, outGain = db2linear(0) */
  }
}
static int var_state(const var_t* d, float* o) {
  int n = 0;
  o[n++] = d->smAmount; o[n++] = d->smSens;
  for (int c = 0; c < 2; ++c) { o[n++] = d->det[c].s1; o[n++] = d->det[c].s2; }
  for (int c = 0; c < 2; ++c) { o[n++] = d->hf1[c].s1; o[n++] = d->hf1[c].s2; o[n++] = d->hf2[c].s1; o[n++] = d->hf2[c].s2; }
  o[n++] = d->air.s1; o[n++] = d->air.s2; o[n++] = d->hfLvl;
  for (int c = 0; c < 2; ++c) { o[n++] = d->sm1[c]; o[n++] = d->sm2[c]; o[n++] = d->d1[c]; o[n++] = d->d2[c]; }
  o[n++] = d->env; o[n++] = (float)(d->rnd >> 16); o[n++] = (float)(d->rnd & 0xffff);   /* 32 bits in two exact floats */
  return n;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Dynamics/RED -- plugins/Dynamics/RED/src/Reverb Expanding Downwards (RED).dsp   (6 in / 6 out)
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int sr;
  float wet_env2, ref_env2, offA_s, tgt_db, dryA_s, tgt_hold, gr_norm, gr_fast;
} red_t;
static float red_pole(float ms, float SR) { return f_exp(-1.0f / (SR * (ms / 1000.0f))); }   /* ms2pole :22 */
static float red_clamp(float x, float lo, float hi) { return f_max(lo, f_min(hi, x)); }
static float red_smoothstep01(float x) { const float x1 = red_clamp(x, 0.0f, 1.0f); return x1 * x1 * (3.0f - 2.0f * x1); }
static void red_compute(red_t* d, const float* zone, int count, float** in, float** out) {
  const float SR = ma_SR(d->sr);
  const float eps = 1e-12f;
  const float maxduck_dB = zone[0], sens = zone[1] / 100.0f, rel_ms = zone[2];
  const float thr_db = 18.0f - sens * 21.0f, ratio = 1.2f + sens * 3.0f, knee_db = 10.0f - sens * 6.0f;      /* :49-51 */
  const float grace_ms = red_clamp(rel_ms * 0.25f, 60.0f, 200.0f);
  const float pole_rms = red_pole(35.0f, SR), pole_tgt = red_pole(25.0f, SR), pole_grace = red_pole(grace_ms, SR);
  const float pole_hold = red_pole(80.0f, SR), pole_10 = red_pole(10.0f, SR);
  const float dry_on_lin = f_pow(10.0f, -50.0f / 20.0f), ref_off_lin = f_pow(10.0f, -60.0f / 20.0f);
  const float floor_lin = f_pow(10.0f, -80.0f / 20.0f);
  const float catt = f_exp(-1.0f / ((12.0f / 1000.0f) * SR));                                                 /* tau2pole */
  const float crel = f_exp(-1.0f / ((rel_ms / 1000.0f) * SR)), crel_in = f_exp(-1.0f / ((70.0f / 1000.0f) * SR));
  const float knee = f_max(knee_db, 0.001f);
  for (int i = 0; i < count; ++i) {
    const float wetL = in[0][i], wetR = in[1][i], refL = in[4][i], refR = in[5][i];
    const float wet_p = 0.5f * (wetL * wetL + wetR * wetR), ref_p = 0.5f * (refL * refL + refR * refR);       /* :66-67 */
    const float wet_env2 = smooth_step(pole_rms, wet_p, &d->wet_env2), ref_env2 = smooth_step(pole_rms, ref_p, &d->ref_env2);
    const float Ey = f_max(sqrtf(f_max(wet_env2, 0.0f)), floor_lin), Ex = f_max(sqrtf(f_max(ref_env2, 0.0f)), floor_lin);
    const float dryA = (float)(Ex > dry_on_lin), offA = (float)(Ex <= ref_off_lin);                           /* :76-77 */
    const float offA_s = smooth_step(pole_grace, offA, &d->offA_s);
    const float tail_w = (1.0f - offA) + offA * red_smoothstep01(offA_s);                                     /* :82 */
    const float rdB = 20.0f * f_log10(f_max((Ey + eps) / (Ex + eps), 1e-30f));                                /* :85 */
    const float over = rdB - thr_db;
    const float over_eff = (over <= 0.0f) ? 0.0f : over * red_smoothstep01(red_clamp(over / knee, 0.0f, 1.0f));
    const float tgt0 = (over_eff > 0.0f) ? f_min(maxduck_dB, over_eff * ratio) : 0.0f;                       /* :95-96 */
    const float tgt1 = tgt0 * tail_w;
    const float tgt_db = smooth_step(pole_tgt, tgt1, &d->tgt_db);                                             /* :99 */
    const float dryA_s = smooth_step(pole_10, dryA, &d->dryA_s);                                              /* :105 */
    const float tgt_hold = f_max(tgt_db, smooth_step(pole_hold, tgt_db, &d->tgt_hold));                       /* :110 */
    const float tgt_pin = (1.0f - dryA) * tgt_hold + dryA * tgt_db;
    const float gr_norm = one_pole_switching(catt, crel, fabsf(tgt_pin), &d->gr_norm);                        /* :118-119 */
    const float gr_fast = one_pole_switching(catt, crel_in, fabsf(tgt_pin), &d->gr_fast);
    const float gr_db = (1.0f - dryA_s) * gr_norm + dryA_s * gr_fast;                                         /* :122 */
    const float g = f_pow(10.0f, (0.0f - gr_db) / 20.0f);                                                     /* :125 */
    out[0][i] = wetL * g; out[1][i] = wetR * g;
    out[2][i] = in[2][i]; out[3][i] = in[3][i]; out[4][i] = refL; out[5][i] = refR;
  }
}
static int red_state(const red_t* d, float* o) {
  o[0] = d->wet_env2; o[1] = d->ref_env2; o[2] = d->offA_s; o[3] = d->tgt_db; o[4] = d->dryA_s; o[5] = d->tgt_hold;
  o[6] = d->gr_norm; o[7] = d->gr_fast;
  return 8;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * C API for oracle/faust_ref.py.  leaf: 0 = ClickBeGoneSG, 1 = ModTilt, 2 = GTS, 3 = VAR, 4 = RED
 * ------------------------------------------------------------------------------------------------------------------- */
int fref_state_bytes(int leaf) {
  switch (leaf) { case 0: return (int)sizeof(cbg_t); case 1: return (int)sizeof(mt_t); case 2: return (int)sizeof(gts_t);
                  case 3: return (int)sizeof(var_t); case 4: return (int)sizeof(red_t); default: return -1; }
}
int fref_channels(int leaf) { return leaf == 4 ? 6 : (leaf >= 0 && leaf <= 3) ? 2 : -1; }
/* mydsp::init(sample_rate): instanceConstants + instanceClear */
void fref_init(int leaf, void* st, int sr) {
  memset(st, 0, (size_t)fref_state_bytes(leaf));
  *(int*)st = sr;                       /* every state struct starts with `int sr` */
}
void fref_compute(int leaf, void* st, const float* zones, int count, float** in, float** out) {
  switch (leaf) {
    case 0: cbg_compute((cbg_t*)st, zones, count, in, out); break;
    case 1: mt_compute((mt_t*)st, zones, count, in, out); break;
    case 2: gts_compute((gts_t*)st, zones, count, in, out); break;
    case 3: var_compute((var_t*)st, zones, count, in, out); break;
    case 4: red_compute((red_t*)st, zones, count, in, out); break;
  }
}
int fref_state(int leaf, const void* st, float* o) {
  switch (leaf) {
    case 0: return cbg_state((const cbg_t*)st, o);
    case 1: return mt_state((const mt_t*)st, o);
    case 2: return gts_state((const gts_t*)st, o);
    case 3: return var_state((const var_t*)st, o);
    case 4: return red_state((const red_t*)st, o);
    default: return -1;
  }
}
