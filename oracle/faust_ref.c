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
 * C API for oracle/faust_ref.py.  leaf: 0 = ClickBeGoneSG, 1 = ModTilt
 * ------------------------------------------------------------------------------------------------------------------- */
int fref_state_bytes(int leaf) { return leaf == 0 ? (int)sizeof(cbg_t) : leaf == 1 ? (int)sizeof(mt_t) : -1; }
int fref_channels(int leaf) { return (leaf == 0 || leaf == 1) ? 2 : -1; }
/* mydsp::init(sample_rate): instanceConstants + instanceClear */
void fref_init(int leaf, void* st, int sr) {
  memset(st, 0, (size_t)fref_state_bytes(leaf));
  if (leaf == 0) ((cbg_t*)st)->sr = sr; else ((mt_t*)st)->sr = sr;
}
void fref_compute(int leaf, void* st, const float* zones, int count, float** in, float** out) {
  if (leaf == 0) cbg_compute((cbg_t*)st, zones, count, in, out);
  else if (leaf == 1) mt_compute((mt_t*)st, zones, count, in, out);
}
int fref_state(int leaf, const void* st, float* o) {
  return leaf == 0 ? cbg_state((const cbg_t*)st, o) : leaf == 1 ? mt_state((const mt_t*)st, o) : -1;
}
