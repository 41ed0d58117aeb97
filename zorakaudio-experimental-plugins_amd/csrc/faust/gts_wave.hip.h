// gts_wave.hip.h -- Dynamics/GTS, restated from "Gaussian Transient Shaper (GTS).dsp" (reference:
// plugins/Dynamics/GTS/src/Gaussian Transient Shaper (GTS).dsp; line numbers refer to it). f32 throughout.
//
// The five UI values go through si.smoo, so sigma -- and with it the whole 257-tap Gaussian kernel (129 distinct taps,
// each an exp) -- is a per-SAMPLE signal: ~129 exp + 2 x 257 multiply-adds per frame. Apart from the five one-pole
// smoothers (linear recursions on the parameters, independent of the audio) the leaf is feed-forward, so the kernel runs
// ONE WAVEFRONT PER INSTANCE with ONE LANE PER FRAME: lanes 0..4 advance one smoother each over the 64 frames of a chunk
// (serially, so their rounding is the serial one), then every lane builds the kernel of its own frame in registers and
// runs the FIR over an LDS row holding the last 256 + 64 input frames of each channel.
#pragma once

#include "faust_lane.hip.h"

struct ZfGts {
  static constexpr int NCH = 2;
  static constexpr int NPARAM = 5;       // Sigma [ms], Attack Gain [dB], Sustain Gain [dB], Mix, Output Gain [dB] (:52-65)
  static constexpr int R = 128;          // GAUSS_RADIUS (:13)
  static constexpr int HIST = 2 * R;     // x@1 .. x@256
  static constexpr int S_SM = 0, S_HL = 5, S_HR = 5 + HIST;
  static constexpr int NSTATE = 5 + 2 * HIST;
  static const char* const* names;
};

#define ZF_GTS_KERNEL_NAME "zf_gts_wave"

__device__ __forceinline__ float zf_readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__global__ void __launch_bounds__(64) zf_gts_prepare(ZabBatch b) {      // dsp->init(): instanceClear
  const int inst = blockIdx.x;
  for (int k = threadIdx.x; k < ZfGts::NSTATE; k += 64) b.vars[k * b.var_se + inst * b.var_si] = 0.0;
  if (threadIdx.x == 0) b.flags[inst] = ZAB_FLAG_PREPARED;
}

__global__ void __launch_bounds__(64) zf_gts_wave(ZabBatch b, ZabAudio a) {
  using L = ZfGts;
  __shared__ float xs[2][L::HIST + 64];      // [0..255]: x@256..x@1 of the chunk's first frame, [256..319]: this chunk
  __shared__ float sm[5][64];                // smoothed parameters per frame
  __shared__ float gsh[ZfGts::R + 1];        // the Gaussian taps of a chunk whose 64 frames share one sigma
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  const float SR = zf_sr(b.srate);
  const float s = 1.0f - 44.1f / SR;                                    // si.smoo = si.smooth(1 - 44.1/ma.SR)
  // lanes 0..4: one smoother each (state + target)
  float y = 0.0f, target = 0.0f;
  if (lane < 5) {
    y = (float)b.vars[(L::S_SM + lane) * b.var_se + inst * b.var_si];
    target = (float)b.sliders[lane * b.sl_se + inst * b.sl_si];
    if (lane == 4) target = zf_pow(10.0f, target / 20.0f);              // outGain: ba.db2linear before si.smoo (:64-65)
  }
  if (lane == 0) b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  for (int d = 1 + lane; d <= L::HIST; d += 64) {
    xs[0][L::HIST - d] = (float)b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si];
    xs[1][L::HIST - d] = (float)b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si];
  }
  const float* in0 = a.in + (int64_t)inst * 2 * a.frame_stride;
  float* out0 = a.out + (int64_t)inst * 2 * a.frame_stride;
  float nx0 = lane < a.frames ? in0[lane] : 0.0f, nx1 = lane < a.frames ? in0[a.frame_stride + lane] : 0.0f;
  zf_wave_sync();

  for (int64_t t0 = 0; t0 < a.frames; t0 += 64) {
    const int tn = (int)((a.frames - t0 < 64) ? (a.frames - t0) : 64);
    xs[0][L::HIST + lane] = nx0; xs[1][L::HIST + lane] = nx1;
    {                                                                   // next chunk's HBM read, a chunk ahead
      const int64_t t = t0 + 64 + lane;
      nx0 = t < a.frames ? in0[t] : 0.0f;
      nx1 = t < a.frames ? in0[a.frame_stride + t] : 0.0f;
    }
    if (lane < 5) {                                                     // si.smoo recursions, serial in time
      for (int n = 0; n < tn; ++n) {
        y = target * (1.0f - s) + s * y;
        sm[lane][n] = y;
      }
    }
    zf_wave_sync();
    // ---- lane = frame ----------------------------------------------------------------------------------------------
    const float sigmaMs = sm[0][lane], attackDB = sm[1][lane], sustainDB = sm[2][lane], mix = sm[3][lane], outGain = sm[4][lane];
    const float sigmaSamples = zf_max(0.25f, sigmaMs * SR * 0.001f);    // :69-70
    float g[L::R + 1];
    // Once the sigma smoother has settled (f32: it stops moving after a few thousand frames) all 64 frames of a chunk have
    // the same 129 taps: then each is computed once, by one lane, and read back by all (the same expf of the same argument,
    // so the same bits as computing all of them in every lane) -- 129 exp per chunk instead of 129 per frame.
    const bool same_sigma = __ballot(sigmaSamples != zf_readlane_f32(sigmaSamples, 0)) == 0ull;
    if (same_sigma) {
      for (int j = lane; j <= L::R; j += 64) {
        const float q = (float)j / sigmaSamples;
        gsh[j] = expf(-0.5f * (q * q));
      }
      zf_wave_sync();
#pragma unroll
      for (int j = 0; j <= L::R; ++j) g[j] = gsh[j];
    } else {
#pragma unroll
      for (int j = 0; j <= L::R; ++j) {                                 // g(i) = exp(-0.5 * (i/sigma)^2)   (:27)
        const float q = (float)j / sigmaSamples;
        g[j] = expf(-0.5f * (q * q));
      }
    }
    float sumRest = g[1];                                               // :31
#pragma unroll
    for (int j = 2; j <= L::R; ++j) sumRest = sumRest + g[j];
    const float norm = 1.0f / (g[0] + 2.0f * sumRest + 1e-20f);         // :32
    const float aGain = zf_pow(10.0f, attackDB / 20.0f), sGain = zf_pow(10.0f, sustainDB / 20.0f);   // :88-89
    float o[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float* row = &xs[ch][L::HIST + lane];                       // row[-k] = x@k
      float sustain = (norm * g[L::R]) * row[0];                        // fi.fir: taps k = 0..256 in order (:41)
#pragma unroll
      for (int k = 1; k <= 2 * L::R; ++k) {
        const int off = k < L::R ? L::R - k : k - L::R;
        sustain = sustain + (norm * g[off]) * row[-k];
      }
      const float xAligned = row[-L::R];                                // de.delay(128, 128)   (:80)
      const float attack = xAligned - sustain;
      const float shaped = aGain * attack + sGain * sustain;            // :91
      o[ch] = ((mix * shaped) + ((1.0f - mix) * xAligned)) * outGain;   // :94
    }
    if (lane < tn) { out0[t0 + lane] = o[0]; out0[a.frame_stride + t0 + lane] = o[1]; }
    // roll the history: frames tn-256 .. tn-1 of the extended row become x@256..x@1 of the next chunk
    float keep[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { keep[0][q] = xs[0][tn + lane + 64 * q]; keep[1][q] = xs[1][tn + lane + 64 * q]; }
    zf_wave_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) { xs[0][lane + 64 * q] = keep[0][q]; xs[1][lane + 64 * q] = keep[1][q]; }
    zf_wave_sync();
  }
  if (lane < 5) b.vars[(L::S_SM + lane) * b.var_se + inst * b.var_si] = (double)y;
  for (int d = 1 + lane; d <= L::HIST; d += 64) {
    b.vars[(L::S_HL + d - 1) * b.var_se + inst * b.var_si] = (double)xs[0][L::HIST - d];
    b.vars[(L::S_HR + d - 1) * b.var_se + inst * b.var_si] = (double)xs[1][L::HIST - d];
  }
}

static hipError_t zf_gts_launch_prepare(const ZabBatch* b, hipStream_t st) {
  hipLaunchKernelGGL(zf_gts_prepare, dim3(b->n_inst), dim3(64), 0, st, *b);
  return hipGetLastError();
}
static hipError_t zf_gts_launch_process(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  if (a->frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(zf_gts_wave, dim3(b->n_inst), dim3(64), 0, st, *b, *a);
  return hipGetLastError();
}
static hipError_t zf_gts_launch_slider(const ZabBatch*, hipStream_t) { return hipSuccess; }

// state names: 5 smoothers, then the two delay lines
static const char* const* zf_gts_names() {
  static char text[ZfGts::NSTATE][12];
  static const char* ptr[ZfGts::NSTATE];
  static const char* const sm[5] = {"smoo_sigma", "smoo_attack", "smoo_sustain", "smoo_mix", "smoo_outgain"};
  for (int k = 0; k < 5; ++k) ptr[k] = sm[k];
  for (int c = 0; c < 2; ++c)
    for (int d = 1; d <= ZfGts::HIST; ++d) {
      const int k = 5 + c * ZfGts::HIST + d - 1;
      snprintf(text[k], sizeof text[k], "%c@%d", c ? 'R' : 'L', d);
      ptr[k] = text[k];
    }
  return ptr;
}
#define ZF_DEFINE_GTS_MODULE(KEY)                                                                                        \
  extern "C" const ZabModule* zab_module_get(void) {                                                                    \
    static const ZabModule m = {ZAB_MODULE_ABI, KEY, ZfGts::NSTATE, 2, 2, 2, 1, 0, 0, 1, 0, 64, zf_gts_names(), 0, 0, 0, 0, 0, \
                                zf_gts_launch_prepare, zf_gts_launch_process, zf_gts_launch_slider, nullptr, nullptr,   \
                                nullptr, ZF_GTS_KERNEL_NAME, nullptr, nullptr};                                                  \
    return &m;                                                                                                          \
  }
