// clickbegone_wave.hip.h -- hand-written gfx950 kernel for Restoration/ClickBeGoneSG (BASELINE config C5: 1024 instances per
// GPU). Same arithmetic, operation for operation, as the lane-per-instance kernel (clickbegone.hip.h) and therefore the
// same bits; what changes is WHERE each operation runs:
//
//   * the .dsp is feed-forward except for four short recursions (two HPFs -> envelope -> baseline, and the hold envelope).
//     Everything feed-forward -- the Savitzky-Golay predictors (36..52 taps per channel), the error norms with their
//     divisions, the trigger ratio, the final mix -- is computed with ONE LANE PER FRAME, 64 frames of one instance at a
//     time, reading the input history from an LDS row;
//   * the recursions run ONE LANE PER INSTANCE over the 64 frames of the chunk (2 x 64 dependent steps of a few flops),
//     exchanging per-frame values with the frame-parallel phases through LDS.
//   A workgroup is one wavefront serving G instances (G = 1 or 4, picked from the batch size): with 1024 instances
//   and G = 1 the chip runs 1024 wavefronts instead of the 16 a lane-per-instance mapping gives, and each spends ~12
//   serial VALU ops per frame instead of ~150.
//   * HBM: 8 B in + 8 B out per frame, 256-byte row segments per wave access; state is read and written once per launch.
#pragma once

#include "clickbegone.hip.h"

static char zf_cbg_kernel_name[24] = "zf_cbg_wave_quad";   /* the kernel the last launch took: zf_cbg_wave_quad | zf_cbg_wave */
#define ZF_CBG_FAST_NAME zf_cbg_kernel_name

template <int G>
__global__ void __launch_bounds__(64) zf_cbg_wave(ZabBatch b, ZabAudio a) {
  using L = ZfClickBeGone;
  // rows read or written by ONE lane over a chunk (the recursions) are 16-byte aligned when the wave serves one instance, so
  // that lane moves four frames per LDS instruction; with G > 1 the odd row length keeps the G lanes on different banks
  constexpr int RP = G == 1 ? 68 : 65;
  __shared__ float xs[G][2][96];        // [0..31]: the previous chunk's last 32 frames, [32..95]: this chunk
  __shared__ __attribute__((aligned(16))) float us[G][2][RP];   // HPF input, already scaled: a * (x - x@1)
  __shared__ float pp[G][5][64];        // Pred fields per frame
  __shared__ __attribute__((aligned(16))) float eb[G][2][RP];   // env, base per frame
  __shared__ __attribute__((aligned(16))) float th[G][RP];      // trigger, then hold, per frame
  __shared__ L::Ctl ctls[G];
  const int lane = threadIdx.x;
  const int inst0 = blockIdx.x * G;
  const int ng = (b.n_inst - inst0) < G ? (b.n_inst - inst0) : G;     // live instances of this wave (uniform)
  const float SR = zf_sr(b.srate);

  // ---- launch prologue: recursion state + constants in lane g, history rows in LDS ---------------------------------
  float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  L::Ctl myc = {};
  if (lane < ng) {
    const int inst = inst0 + lane;
#pragma unroll
    for (int k = 0; k < 5; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];
    float par[L::NPARAM];
#pragma unroll
    for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
    myc = L::control(par, SR);
    ctls[lane] = myc;
    b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  }
  for (int g = 0; g < ng; ++g) {
    const int inst = inst0 + g;
    if (lane < 32) {                                                  // xs[g][ch][32 - d] = x@d, d = 1..30 (lane = 32 - d)
      const int d = 32 - lane;
      float hl = 0.f, hr = 0.f;
      if (d >= 1 && d <= L::HIST) {
        hl = (float)b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si];
        hr = (float)b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si];
      }
      xs[g][0][lane] = hl; xs[g][1][lane] = hr;
    }
  }
  zf_wave_sync();

  // the HBM read of chunk k+1 is issued before chunk k is processed (registers), so its latency hides behind the phases
  float nxL[G], nxR[G];
  auto fetch = [&](int64_t t0) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool ok = g < ng && t0 + lane < a.frames;
      const float* in = a.in + (int64_t)(inst0 + g) * 2 * a.frame_stride + t0;
      nxL[g] = ok ? in[lane] : 0.0f;
      nxR[g] = ok ? in[a.frame_stride + lane] : 0.0f;
    }
  };
  fetch(0);
  for (int64_t t0 = 0; t0 < a.frames; t0 += 64) {
    const int tn = (int)((a.frames - t0 < 64) ? (a.frames - t0) : 64);
    // ---- A: this chunk's input rows -------------------------------------------------------------------------------
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) { xs[g][0][32 + lane] = nxL[g]; xs[g][1][32 + lane] = nxR[g]; }
    }
    fetch(t0 + 64);
    zf_wave_sync();
    // ---- B: feed-forward, lane = frame: predictors, error norms, HPF input ---------------------------------------------
    for (int g = 0; g < ng; ++g) {
      const L::Ctl c = ctls[g];
      const L::RowHist aL{&xs[g][0][32 + lane]}, aR{&xs[g][1][32 + lane]};
      const L::Pred q = L::predict_uniform(c, aL, aR);
      pp[g][0][lane] = q.xC_L; pp[g][1][lane] = q.xC_R; pp[g][2][lane] = q.pred_L; pp[g][3][lane] = q.pred_R; pp[g][4][lane] = q.e_norm;
      us[g][0][lane] = c.a * (aL(0) - aL(1));          // (the recursion's first product, frame-parallel: same operands, same bits)
      us[g][1][lane] = c.a * (aR(0) - aR(1));
    }
    zf_wave_sync();
    // ---- C: recursion 1, lane = instance. Its feed-forward part -- the product env * base_a inside the baseline's update --
    // is taken out of the serial chain (round 4): C1 runs the HPFs and the envelope over the chunk (4 dependent VALU
    // operations per frame: packed multiply, packed add, multiply, three-way maximum), C2 forms env * base_a with one lane
    // per frame, C3 runs the baseline's two remaining operations per frame. Same IEEE operations on the same operands.
    if (lane < ng) {
      if (tn == 64) {                                                 // full chunk: straight-line code, LDS reads up front
        float ul[64], ur[64];
#pragma unroll
        for (int n = 0; n < 64; ++n) { ul[n] = us[lane][0][n]; ur[n] = us[lane][1][n]; }
#pragma unroll
        for (int n = 0; n < 64; ++n) eb[lane][0][n] = L::env_step(st, myc, ul[n], ur[n]);
      } else {
        for (int n = 0; n < tn; ++n) eb[lane][0][n] = L::env_step(st, myc, us[lane][0][n], us[lane][1][n]);
      }
    }
    zf_wave_sync();
    for (int g = 0; g < ng; ++g) eb[g][1][lane] = eb[g][0][lane] * ctls[g].base_a;         // C2
    zf_wave_sync();
    if (lane < ng) {                                                                        // C3
      if (tn == 64) {
        float ev[64];
#pragma unroll
        for (int n = 0; n < 64; ++n) ev[n] = eb[lane][1][n];
#pragma unroll
        for (int n = 0; n < 64; ++n) eb[lane][1][n] = L::base_step(st, myc, ev[n]);
      } else {
        for (int n = 0; n < tn; ++n) eb[lane][1][n] = L::base_step(st, myc, eb[lane][1][n]);
      }
    }
    zf_wave_sync();
    // ---- D: feed-forward: trigger --------------------------------------------------------------------------------------
    for (int g = 0; g < ng; ++g) th[g][lane] = L::trigger(ctls[g], eb[g][0][lane], eb[g][1][lane], pp[g][4][lane]);
    zf_wave_sync();
    // ---- E: recursion 2, lane = instance -------------------------------------------------------------------------------
    if (lane < ng) {
      if (tn == 64) {
        float tr[64];
#pragma unroll
        for (int n = 0; n < 64; ++n) tr[n] = th[lane][n];
#pragma unroll
        for (int n = 0; n < 64; ++n) th[lane][n] = L::hold_step(st, myc, tr[n]);
      } else {
        for (int n = 0; n < tn; ++n) th[lane][n] = L::hold_step(st, myc, th[lane][n]);
      }
    }
    zf_wave_sync();
    // ---- F: feed-forward: mix and store; roll the history rows ----------------------------------------------------------
    for (int g = 0; g < ng; ++g) {
      const L::Ctl c = ctls[g];
      L::Pred q;
      q.xC_L = pp[g][0][lane]; q.xC_R = pp[g][1][lane]; q.pred_L = pp[g][2][lane]; q.pred_R = pp[g][3][lane]; q.e_norm = pp[g][4][lane];
      float oL, oR;
      L::mixdown(c, q, th[g][lane], oL, oR);
      float* out = a.out + (int64_t)(inst0 + g) * 2 * a.frame_stride + t0;
      if (lane < tn) { out[lane] = oL; out[a.frame_stride + lane] = oR; }
      // frames tn-32 .. tn-1 of the extended row become the next chunk's (or the next launch's) history
      float keepL = 0.f, keepR = 0.f;
      if (lane < 32) { keepL = xs[g][0][tn + lane]; keepR = xs[g][1][tn + lane]; }
      zf_wave_sync();
      if (lane < 32) { xs[g][0][lane] = keepL; xs[g][1][lane] = keepR; }
    }
    zf_wave_sync();
  }

  // ---- launch epilogue ----------------------------------------------------------------------------------------------------
  if (lane < ng) {
    const int inst = inst0 + lane;
#pragma unroll
    for (int k = 0; k < 5; ++k) b.vars[k * b.var_se + inst * b.var_si] = (double)st[k];
  }
  for (int g = 0; g < ng; ++g) {
    const int inst = inst0 + g;
    if (lane < 32) {
      const int d = 32 - lane;
      if (d >= 1 && d <= L::HIST) {
        b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si] = (double)xs[g][0][lane];
        b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si] = (double)xs[g][1][lane];
      }
    }
  }
}

#include "clickbegone_quad.hip.h"

static int zf_cbg_pick_g(int n_inst) {
  // measured on MI355X (48 000 frames, round 4: gpurun_out/s2_cbg_g_sweep.txt -> profiles/r04_cbg_g_sweep.txt), G = 1 / 2 / 4 / 8 / 16:
  //   N = 1024: 2.90 / 4.24 / 6.09 / 10.1 / 24.3 ms;  N = 4096: 6.38 / 5.60 / 6.57 / 10.7 / 24.3;  N = 8192: 12.1 / 10.4 / 7.98 / 10.8 / 25.2
  int g = n_inst <= 2560 ? 1 : (n_inst <= 6144 ? 2 : 4);
  if (const char* e = getenv("ZAB_CBG_G")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) g = v; }
  return g;
}
static int32_t zf_cbg_applies(const ZabBatch*, const ZabAudio* a) { return a->frames > 0 ? 1 : 0; }
static bool zf_cbg_use_quad(int n_inst) {
  // measured (48 000 frames; profiles/r04_cbg_quad_vs_wave.txt, second table): the four-wavefront kernel wins at every batch size
  // since its recursions are hand-ordered -- N = 1024: 1.44 ms against 2.58 (wave, G = 1); 4096: 3.27 against 5.22; 8192: 6.43
  // against 7.04; 16384: 12.6 against 13.8. (Its first cut lost beyond 4608 instances: 9.90 against 7.97 ms at 8192.)
  (void)n_inst;
  bool quad = true;                         // (ZAB_CBG_KERNEL = quad | wave pins it)
  if (const char* e = getenv("ZAB_CBG_KERNEL")) quad = e[0] == 'q';
  return quad;
}
static hipError_t zf_cbg_launch(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  const bool quad = zf_cbg_use_quad(b->n_inst);
  snprintf(zf_cbg_kernel_name, sizeof zf_cbg_kernel_name, quad ? "zf_cbg_wave_quad" : "zf_cbg_wave");
  if (quad) {
    hipLaunchKernelGGL(zf_cbg_wave_quad, dim3((b->n_inst + 3) / 4), dim3(256), 0, st, *b, *a);
    return hipGetLastError();
  }
  const int g = zf_cbg_pick_g(b->n_inst);
  const dim3 grid((b->n_inst + g - 1) / g), block(64);
  if (g == 1) hipLaunchKernelGGL(zf_cbg_wave<1>, grid, block, 0, st, *b, *a);
  else if (g == 2) hipLaunchKernelGGL(zf_cbg_wave<2>, grid, block, 0, st, *b, *a);
  else if (g == 4) hipLaunchKernelGGL(zf_cbg_wave<4>, grid, block, 0, st, *b, *a);
  else if (g == 8) hipLaunchKernelGGL(zf_cbg_wave<8>, grid, block, 0, st, *b, *a);
  else hipLaunchKernelGGL(zf_cbg_wave<16>, grid, block, 0, st, *b, *a);
  return hipGetLastError();
}
