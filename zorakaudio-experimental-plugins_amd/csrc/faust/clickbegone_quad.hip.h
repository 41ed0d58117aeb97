// clickbegone_quad.hip.h -- Restoration/ClickBeGoneSG, round 4: FOUR wavefronts for FOUR instances, specialised by role and
// pipelined a chunk apart (BASELINE config C5; the third kernel of this leaf after the lane-per-instance one and zf_cbg_wave).
//
// zf_cbg_wave gives every instance its own wavefront, which alternates between frame-parallel phases (64 lanes busy) and the
// recursions (ONE lane busy: two HPFs -> envelope -> baseline, and the hold envelope -- ~9 dependent VALU operations per frame, each
// a four-cycle issue slot whatever its lane count). At 1024 instances that is one wavefront per SIMD and nothing to overlap the
// serial slots with: they are about two thirds of the kernel. Here a workgroup is four wavefronts serving four instances:
//
//   waves 2, 3   B: the feed-forward front of two instances each, one lane per frame -- Savitzky-Golay predictors, error norms,
//                   the scaled HPF input -- for chunk t;
//   wave 0       S: recursion 1 of ALL FOUR instances in lanes 0..3 (the same instruction stream serves four instances) for
//                   chunk t - 1;
//   wave 1       D/E/F: trigger (lane = frame), recursion 2 of all four instances in lanes 0..3, mix and store for chunk t - 2.
//
// One workgroup barrier per 64-frame tick; the rows handed from role to role are double (us, eb) or triple (pp) buffered in LDS.
// Per tick the four SIMDs of a CU issue ~520 (B, two instances), ~520 (B), ~550 (S) and ~330 (D/E/F) instructions instead of four
// times ~910. Every IEEE operation is the one the other two kernels perform, on the same operands, in the same order per
// recursion: same bits (tests/test_faust.py runs all three against the restatement).
#pragma once

#include "clickbegone.hip.h"

ZF_FN void zf_quad_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // LDS only: the HBM prefetch stays in flight

__global__ void __launch_bounds__(256) zf_cbg_wave_quad(ZabBatch b, ZabAudio a) {
  using L = ZfClickBeGone;
  constexpr int G = 4;
  constexpr int RP = 68;      // rows one lane walks: 16-byte aligned (four frames per LDS instruction), lanes 0..3 land on banks 0, 4, 8, 12
  __shared__ float xs[G][2][96];                                      // [0..31]: the previous chunk's last 32 frames, [32..95]: this chunk
  constexpr int RP2 = 2 * 64 + 4;                                      // rows of (left, right) / (env, base) PAIRS, same alignment and banks
  __shared__ __attribute__((aligned(16))) float us[2][G][RP2];        // B -> S: HPF input of both channels, already scaled, frame-major
  __shared__ float pp[3][G][5][64];                                   // B -> D / F: Pred fields per frame
  __shared__ __attribute__((aligned(16))) float eb[2][G][RP2];        // S -> D: (env, base) per frame
  __shared__ __attribute__((aligned(16))) float th[G][RP];            // D -> E -> F (wave 1 only): trigger, then hold
  __shared__ L::Ctl ctls[G];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int inst0 = blockIdx.x * G;
  const int ng = (b.n_inst - inst0) < G ? (b.n_inst - inst0) : G;     // live instances of this workgroup (uniform)
  const float SR = zf_sr(b.srate);
  const int64_t frames = a.frames;
  const int64_t nchunks = (frames + 63) / 64;

  // ---- launch prologue -------------------------------------------------------------------------------------------------------
  float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // wave 0: hpL hpR env base; wave 1: hold -- of instance inst0 + lane
  L::Ctl myc = {};
  if (wave < 2 && lane < ng) {
    const int inst = inst0 + lane;
#pragma unroll
    for (int k = 0; k < 5; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];
    float par[L::NPARAM];
#pragma unroll
    for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
    myc = L::control(par, SR);
    if (wave == 0) { ctls[lane] = myc; b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY; }
  }
  const int g0 = (wave - 2) * 2;                 // waves 2, 3: their two instances
  if (wave >= 2) {
    for (int gi = 0; gi < 2; ++gi) {
      const int g = g0 + gi;
      if (g < ng && lane < 32) {                 // xs[g][ch][32 - d] = x@d, d = 1..30 (lane = 32 - d)
        const int inst = inst0 + g, d = 32 - lane;
        float hl = 0.f, hr = 0.f;
        if (d >= 1 && d <= L::HIST) {
          hl = (float)b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si];
          hr = (float)b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si];
        }
        xs[g][0][lane] = hl; xs[g][1][lane] = hr;
      }
    }
  }
  // the HBM read of chunk t + 1 is issued before chunk t is processed (waves 2, 3)
  float nxL[2] = {0.f, 0.f}, nxR[2] = {0.f, 0.f};
  auto fetch = [&](int64_t t0) __attribute__((always_inline)) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = g0 + gi;
      const bool ok = g < ng && t0 + lane < frames;
      const float* in = a.in + (int64_t)(inst0 + (g < ng ? g : 0)) * 2 * a.frame_stride + t0;
      nxL[gi] = ok ? in[lane] : 0.0f;
      nxR[gi] = ok ? in[a.frame_stride + lane] : 0.0f;
    }
  };
  if (wave >= 2) fetch(0);
  __syncthreads();

#ifdef ZF_QUAD_CLOCKS          // role clock (tools/cbg_quad_clocks.py): cycles each wavefront works per launch, of the launch's cycles
  uint64_t zc_busy = 0;
  const uint64_t zc_start = __builtin_readcyclecounter();
#endif
  for (int64_t tick = 0; tick < nchunks + 2; ++tick) {
#ifdef ZF_QUAD_CLOCKS
    const uint64_t zc_t0 = __builtin_readcyclecounter();
#endif
    if (wave >= 2) {
      // ---- B(tick): feed-forward front, lane = frame ------------------------------------------------------------------------
      const int64_t k = tick;
      if (k < nchunks) {
        const int64_t t0 = k * 64;
        const int tn = (int)((frames - t0 < 64) ? (frames - t0) : 64);
        const int ub = (int)(k & 1), pb = (int)(k % 3);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const int g = g0 + gi;
          if (g < ng) { xs[g][0][32 + lane] = nxL[gi]; xs[g][1][32 + lane] = nxR[gi]; }
        }
        fetch(t0 + 64);
        zf_wave_sync();
        for (int gi = 0; gi < 2; ++gi) {
          const int g = g0 + gi;
          if (g >= ng) break;
          const L::Ctl c = ctls[g];
          const L::RowHist aL{&xs[g][0][32 + lane]}, aR{&xs[g][1][32 + lane]};
          const L::Pred q = L::predict(c, aL, aR);
          pp[pb][g][0][lane] = q.xC_L; pp[pb][g][1][lane] = q.xC_R; pp[pb][g][2][lane] = q.pred_L; pp[pb][g][3][lane] = q.pred_R;
          pp[pb][g][4][lane] = q.e_norm;
          us[ub][g][2 * lane] = c.a * (aL(0) - aL(1));
          us[ub][g][2 * lane + 1] = c.a * (aR(0) - aR(1));
          // frames tn-32 .. tn-1 of the extended row become the next chunk's (or the next launch's) history
          float keepL = 0.f, keepR = 0.f;
          if (lane < 32) { keepL = xs[g][0][tn + lane]; keepR = xs[g][1][tn + lane]; }
          zf_wave_sync();
          if (lane < 32) { xs[g][0][lane] = keepL; xs[g][1][lane] = keepR; }
        }
      }
    } else if (wave == 0) {
      // ---- S(tick - 1): recursion 1 of the four instances, lane = instance ----------------------------------------------------
      const int64_t k = tick - 1;
      if (k >= 0 && k < nchunks && lane < ng) {
        const int64_t t0 = k * 64;
        const int tn = (int)((frames - t0 < 64) ? (frames - t0) : 64);
        const int ub = (int)(k & 1);
        if (tn == 64) {                                               // full chunk: straight-line code, LDS reads up front
          float ul[64], ur[64];
#pragma unroll
          for (int n = 0; n < 64; ++n) { ul[n] = us[ub][lane][2 * n]; ur[n] = us[ub][lane][2 * n + 1]; }
          L::detect_chunk64(st, myc, ul, ur, &eb[ub][lane][0]);
        } else {
          for (int n = 0; n < tn; ++n) {
            float env, base;
            L::detect_scaled(st, myc, us[ub][lane][2 * n], us[ub][lane][2 * n + 1], env, base);
            eb[ub][lane][2 * n] = env; eb[ub][lane][2 * n + 1] = base;
          }
        }
      }
    } else {
      // ---- D / E / F (tick - 2): trigger, recursion 2, mix and store ---------------------------------------------------------------
      const int64_t k = tick - 2;
      if (k >= 0 && k < nchunks) {
        const int64_t t0 = k * 64;
        const int tn = (int)((frames - t0 < 64) ? (frames - t0) : 64);
        const int ub = (int)(k & 1), pb = (int)(k % 3);
        for (int g = 0; g < ng; ++g) th[g][lane] = L::trigger(ctls[g], eb[ub][g][2 * lane], eb[ub][g][2 * lane + 1], pp[pb][g][4][lane]);
        zf_wave_sync();
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
        for (int g = 0; g < ng; ++g) {
          const L::Ctl c = ctls[g];
          L::Pred q;
          q.xC_L = pp[pb][g][0][lane]; q.xC_R = pp[pb][g][1][lane]; q.pred_L = pp[pb][g][2][lane]; q.pred_R = pp[pb][g][3][lane];
          q.e_norm = pp[pb][g][4][lane];
          float oL, oR;
          L::mixdown(c, q, th[g][lane], oL, oR);
          float* out = a.out + (int64_t)(inst0 + g) * 2 * a.frame_stride + t0;
          if (lane < tn) { out[lane] = oL; out[a.frame_stride + lane] = oR; }
        }
        zf_wave_sync();
      }
    }
#ifdef ZF_QUAD_CLOCKS
    zc_busy += __builtin_readcyclecounter() - zc_t0;
#endif
    zf_quad_barrier();
  }
#ifdef ZF_QUAD_CLOCKS
  if (blockIdx.x == 5 && lane == 0)
    printf("zf_cbg_wave_quad role %d: busy %llu of %llu cycles over %lld ticks\n", wave, (unsigned long long)zc_busy,
           (unsigned long long)(__builtin_readcyclecounter() - zc_start), (long long)(nchunks + 2));
#endif

  // ---- launch epilogue ----------------------------------------------------------------------------------------------------------
  if (wave == 0 && lane < ng) {
    const int inst = inst0 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) b.vars[k * b.var_se + inst * b.var_si] = (double)st[k];
  }
  if (wave == 1 && lane < ng) b.vars[L::S_HOLD * b.var_se + (inst0 + lane) * b.var_si] = (double)st[L::S_HOLD];
  if (wave >= 2) {
    for (int gi = 0; gi < 2; ++gi) {
      const int g = g0 + gi;
      if (g < ng && lane < 32) {
        const int inst = inst0 + g, d = 32 - lane;
        if (d >= 1 && d <= L::HIST) {
          b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si] = (double)xs[g][0][lane];
          b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si] = (double)xs[g][1][lane];
        }
      }
    }
  }
}
