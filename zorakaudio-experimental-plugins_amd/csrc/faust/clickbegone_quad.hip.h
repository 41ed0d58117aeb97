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
//   wave 0       S + H: BOTH recursions of ALL FOUR instances in lanes 0..3 (one instruction stream serves four instances):
//                   recursion 1 (HPFs -> envelope -> baseline) for chunk t - 1 and recursion 2 (the hold envelope) for chunk
//                   t - 3, as four instruction streams interleaved by hand (zf_quad_recursions below);
//   wave 1       D / F: the trigger of chunk t - 2 and the mix and store of chunk t - 4, lane = frame.
//
// One workgroup barrier per 64-frame tick; the rows handed from role to role are double (us, eb), triple (th) or five-fold (pp)
// buffered in LDS. Every IEEE operation is the one the other two kernels perform, on the same operands, in the same order per
// recursion: same bits (tests/test_faust.py runs all three against the restatement).
//
// Role clock (tools/cbg_quad_clocks.py, 1024 instances, cycles per tick at the 1.86 GHz the kernel runs at): the first cut of this
// kernel (hold recursion in wave 1, recursions as compiled) B 5150 / S 3950 / DEF 4400 of 5500; with the predictors' four sums in
// one basic block (predict_uniform) B 3950 of 4750; the figures of this cut: DESIGN.md section 4.4.
#pragma once

#include "clickbegone.hip.h"

ZF_FN void zf_quad_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // LDS only: the HBM prefetch stays in flight

// The serial wavefront's chunk: recursion 1 over the 64 frames of one chunk and recursion 2 over the 64 frames of an earlier one,
// every lane its own instance. Each recurrence closes over two dependent operations per frame (multiply, then add or maximum);
// compiled frame by frame they make one chain whose every link waits for the one before (a dependent VALU operation issues
// ~6.6 cycles after its producer, an independent one after 4), and the SLP vectoriser packs the channel pair into v_pk_* forms that
// need register shuffles. Here the four recurrences run a frame apart -- iteration n does the HPFs of frame n, the envelope of
// frame n - 1, the baseline of frame n - 2 and the hold envelope of frame n -- and each iteration issues its independent
// multiplies first, then the operations that consume them: written as volatile asm so that this order is the order issued.
// The operations are the compiler's own for detect_scaled() / hold_step(): v_mul_f32, v_add_f32, v_max3_f32 with |.| modifiers
// (= max(env * rel, max(|hpL|, |hpR|)): max is exact and associative), v_max_f32. No FMA.
#define ZF_VMUL(d, x, y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define ZF_VADD(d, x, y) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define ZF_VMAX(d, x, y) asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define ZF_VMAX3ABS(d, x, y, z) asm volatile("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(x), "v"(y), "v"(z))
// rec1: state hpL hpR env base in st[0..3], scaled HPF input pairs in `v` (frame-major), (env, base) pairs out;
// rec2: state hold in st[4], trigger row in, hold row out (in place). Either may be absent (pipeline fill and drain).
template <bool REC1, bool REC2>
ZF_FN void zf_quad_recursions(float* st, const ZfClickBeGone::Ctl& c, const float* v, float* env_base_out, float* trig_hold) {
  using L = ZfClickBeGone;
  float hpl = st[L::S_HPL], hpr = st[L::S_HPR], env = st[L::S_ENV], base = st[L::S_BASE], hold = st[L::S_HOLD];
  float vv[128], tr[64];
  if (REC1) {
#pragma unroll
    for (int n = 0; n < 128; ++n) vv[n] = v[n];
  }
  if (REC2) {
#pragma unroll
    for (int n = 0; n < 64; ++n) tr[n] = trig_hold[n];
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  float e_[64], b_[64], h_[64];          // results on their way to LDS: written four floats at a time (both rows are 16-byte aligned)
#pragma unroll
  for (int n = 0; n < 64 + 2; ++n) {
    float m1, m2, m3, m4, m5, m6;
    // independent multiplies of this iteration's four streams
    if (REC1 && n >= 2) { ZF_VMUL(m1, env, c.base_a); ZF_VMUL(m2, base, c.one_m_base_a); }      // baseline of frame n - 2
    if (REC1 && n >= 1 && n <= 64) ZF_VMUL(m3, env, c.env_rel);                                   // envelope of frame n - 1
    if (REC1 && n < 64) { ZF_VMUL(m4, c.a, hpl); ZF_VMUL(m5, c.a, hpr); }                         // HPFs of frame n
    if (REC2 && n < 64) ZF_VMUL(m6, hold, c.relHold);                                             // hold envelope of frame n
    // their consumers (env before the HPFs move on: it reads frame n - 1's outputs)
    if (REC1 && n >= 2) { ZF_VADD(base, m1, m2); b_[n - 2] = base; }
    if (REC1 && n >= 1 && n <= 64) { ZF_VMAX3ABS(env, m3, hpl, hpr); e_[n - 1] = env; }
    if (REC1 && n < 64) { ZF_VADD(hpl, vv[2 * n], m4); ZF_VADD(hpr, vv[2 * n + 1], m5); }
    if (REC2 && n < 64) { ZF_VMAX(hold, m6, tr[n]); h_[n] = hold; }
    if (REC1 && n >= 3 && (n - 3) % 2 == 0) {                   // frames n - 3 and n - 2 are complete
      const int m = n - 3;
      *(f4*)&env_base_out[2 * m] = f4{e_[m], b_[m], e_[m + 1], b_[m + 1]};
    }
    if (REC2 && n < 64 && n % 4 == 3) *(f4*)&trig_hold[n - 3] = f4{h_[n - 3], h_[n - 2], h_[n - 1], h_[n]};
  }
  st[L::S_HPL] = hpl; st[L::S_HPR] = hpr; st[L::S_ENV] = env; st[L::S_BASE] = base; st[L::S_HOLD] = hold;
}

__global__ void __launch_bounds__(256) zf_cbg_wave_quad(ZabBatch b, ZabAudio a) {
  using L = ZfClickBeGone;
  constexpr int G = 4;
  constexpr int RP = 68;      // rows one lane walks: 16-byte aligned (four frames per LDS instruction), lanes 0..3 land on banks 0, 4, 8, 12
  __shared__ float xs[G][2][96];                                      // [0..31]: the previous chunk's last 32 frames, [32..95]: this chunk
  constexpr int RP2 = 2 * 64 + 4;                                      // rows of (left, right) / (env, base) PAIRS, same alignment and banks
  __shared__ __attribute__((aligned(16))) float us[2][G][RP2];        // B -> S: HPF input of both channels, already scaled, frame-major
  __shared__ float pp[5][G][5][64];                                   // B -> D (two ticks later) / F (four ticks later): Pred fields per frame
  __shared__ __attribute__((aligned(16))) float eb[2][G][RP2];        // S -> D: (env, base) per frame
  __shared__ __attribute__((aligned(16))) float th[3][G][RP];         // D -> H -> F: trigger, then (in place) hold
  __shared__ L::Ctl ctls[G];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int inst0 = blockIdx.x * G;
  const int ng = (b.n_inst - inst0) < G ? (b.n_inst - inst0) : G;     // live instances of this workgroup (uniform)
  const float SR = zf_sr(b.srate);
  const int64_t frames = a.frames;
  const int64_t nchunks = (frames + 63) / 64;

  // ---- launch prologue -------------------------------------------------------------------------------------------------------
#ifdef ZF_QUAD_HOLD_IN_S
  constexpr int HW = 0;                         // the wavefront that runs recursion 2
#else
  constexpr int HW = 1;
#endif
  float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // wave 0: hpL hpR env base, wave HW: hold -- of instance inst0 + lane
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
  for (int64_t tick = 0; tick < nchunks + 4; ++tick) {
#ifdef ZF_QUAD_CLOCKS
    const uint64_t zc_t0 = __builtin_readcyclecounter();
#endif
    if (wave >= 2) {
      // ---- B(tick): feed-forward front, lane = frame ------------------------------------------------------------------------
      const int64_t k = tick;
      if (k < nchunks) {
        const int64_t t0 = k * 64;
        const int tn = (int)((frames - t0 < 64) ? (frames - t0) : 64);
        const int ub = (int)(k & 1), pb = (int)(k % 5);
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
          const L::Pred q = L::predict_uniform(c, aL, aR);
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
      // ---- S(tick - 1) and H(tick - 3): the recursions of the four instances, lane = instance -------------------------------
      const int64_t k1 = tick - 1, k2 = tick - 3;
      const bool do1 = k1 >= 0 && k1 < nchunks, do2 = HW == 0 && k2 >= 0 && k2 < nchunks;
      if (lane < ng && (do1 || do2)) {
        const int tn1 = do1 ? (int)((frames - k1 * 64 < 64) ? (frames - k1 * 64) : 64) : 0;
        const int tn2 = do2 ? (int)((frames - k2 * 64 < 64) ? (frames - k2 * 64) : 64) : 0;
        const int ub = (int)(k1 & 1), hb = (int)((k2 + 3) % 3);
        float* const eo = &eb[ub][lane][0];
        float* const hrow = &th[hb][lane][0];
        const float* const vin = &us[ub][lane][0];
        if (tn1 == 64 && tn2 == 64) zf_quad_recursions<true, true>(st, myc, vin, eo, hrow);        // steady state: straight-line code
        else {
          if (tn1 == 64) zf_quad_recursions<true, false>(st, myc, vin, eo, hrow);
          else {
            for (int n = 0; n < tn1; ++n) {
              float env, base;
              L::detect_scaled(st, myc, vin[2 * n], vin[2 * n + 1], env, base);
              eo[2 * n] = env; eo[2 * n + 1] = base;
            }
          }
          if (tn2 == 64) zf_quad_recursions<false, true>(st, myc, vin, eo, hrow);
          else {
            for (int n = 0; n < tn2; ++n) hrow[n] = L::hold_step(st, myc, hrow[n]);
          }
        }
      }
    } else {
      // ---- D(tick - 2): trigger; F(tick - 4): mix and store; lane = frame. All G rows are computed (rows past ng hold stale
      // numbers that nothing reads), so the LDS reads of the four instances overlap instead of queueing behind four branches.
      const int64_t kd = tick - 2, kh = tick - 3, kf = tick - 4;
      if (HW == 1 && kh >= 0 && kh < nchunks && lane < ng) {       // H(tick - 3): recursion 2, lane = instance
        const int tnh = (int)((frames - kh * 64 < 64) ? (frames - kh * 64) : 64);
        float* const hrow = &th[(int)(kh % 3)][lane][0];
        if (tnh == 64) zf_quad_recursions<false, true>(st, myc, hrow, hrow, hrow);
        else {
          for (int n = 0; n < tnh; ++n) hrow[n] = L::hold_step(st, myc, hrow[n]);
        }
      }
      if (kd >= 0 && kd < nchunks) {
        const int ub = (int)(kd & 1), pb = (int)(kd % 5), hb = (int)(kd % 3);
#pragma unroll
        for (int g = 0; g < G; ++g)
          th[hb][g][lane] = L::trigger(ctls[g], eb[ub][g][2 * lane], eb[ub][g][2 * lane + 1], pp[pb][g][4][lane]);
      }
      if (kf >= 0 && kf < nchunks) {
        const int64_t t0 = kf * 64;
        const int tn = (int)((frames - t0 < 64) ? (frames - t0) : 64);
        const int pb = (int)(kf % 5), hb = (int)(kf % 3);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const L::Ctl c = ctls[g];
          L::Pred q;
          q.xC_L = pp[pb][g][0][lane]; q.xC_R = pp[pb][g][1][lane]; q.pred_L = pp[pb][g][2][lane]; q.pred_R = pp[pb][g][3][lane];
          q.e_norm = pp[pb][g][4][lane];
          float oL, oR;
          L::mixdown(c, q, th[hb][g][lane], oL, oR);
          float* out = a.out + (int64_t)(inst0 + (g < ng ? g : 0)) * 2 * a.frame_stride + t0;
          if (g < ng && lane < tn) { out[lane] = oL; out[a.frame_stride + lane] = oR; }
        }
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
           (unsigned long long)(__builtin_readcyclecounter() - zc_start), (long long)(nchunks + 4));
#endif

  // ---- launch epilogue ----------------------------------------------------------------------------------------------------------
  if (wave == 0 && lane < ng) {
    const int inst = inst0 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) b.vars[k * b.var_se + inst * b.var_si] = (double)st[k];
  }
  if (wave == HW && lane < ng) b.vars[L::S_HOLD * b.var_se + (inst0 + lane) * b.var_si] = (double)st[L::S_HOLD];
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
