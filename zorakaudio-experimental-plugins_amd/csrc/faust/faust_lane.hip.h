// faust_lane.hip.h -- batch kernels for the Faust leaves: FaustJuceProcessor::processBlock ->
// mydsp::compute(count, inputs, outputs)  (src/FaustJuceProcessor.cpp:462-482, ABI src/faust_support_min.h:41-53).
//
// The reference compiles each .dsp with the Faust compiler (-single: FAUSTFLOAT = float, all arithmetic f32). That compiler
// and stdfaust.lib are not part of the reference tree, so every leaf here is a hand-written restatement of its .dsp
// (csrc/faust/<leaf>.hip.h, citing the .dsp lines) -- "parity unpinned", see DESIGN.md.
//
// Mapping: ONE LANE PER INSTANCE, b.ipw (<= 64, see zab_generic.hip.h) instances per single-wave workgroup; the leaf's recursive state (a few floats and short
// delay lines) stays in registers for the whole launch. Audio is instance-major planar, so a wave stages a
// [NCH][64 instances][TT frames] tile through LDS exactly like the JSFX generic kernel (zab_generic.hip.h): HBM accesses
// are 128-byte row segments, the per-lane walk is bank-conflict free (+1 padding). In and out alias (in-place), as in
// FaustJuceProcessor::processBlock.
//
// A leaf type L provides:
//   NCH, NSTATE, NPARAM, names[], struct Ctl, control(params, sr) -> Ctl  (the "fSlow"/"fConst" values of a Faust compute()),
//   frame(st, ctl, x)  -- one sample, x[NCH] in place.
// State lives in ZabBatch::vars between launches (floats widened to f64: exact), sliders[k] is the k-th UI zone.
#pragma once

#include "../zab_module.h"

#define ZF_FN __device__ __forceinline__

// Phase boundary inside a SINGLE-WAVE workgroup (the wave kernels): a wave's LDS accesses execute in order, so other lanes'
// LDS writes are visible once the compiler is kept from reordering; __syncthreads() would also drain the HBM prefetch of
// the next chunk and the output stores (s_waitcnt vmcnt(0)) at every phase boundary.
ZF_FN void zf_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// control-rate transcendental functions: evaluated in f64 and rounded once (the CPU restatement does the same), so the
// coefficients on both sides are the same floats
ZF_FN float zf_exp(float x) { return (float)exp((double)x); }
ZF_FN float zf_log10(float x) { return (float)log10((double)x); }
ZF_FN float zf_pow(float x, float y) { return (float)pow((double)x, (double)y); }
ZF_FN float zf_sin(float x) { return (float)sin((double)x); }
ZF_FN float zf_cos(float x) { return (float)cos((double)x); }
ZF_FN float zf_min(float a, float b) { return fminf(a, b); }
ZF_FN float zf_max(float a, float b) { return fmaxf(a, b); }
// ma.SR = min(192000, max(1, fSampleRate))
ZF_FN float zf_sr(double srate) { return fminf(192000.0f, fmaxf(1.0f, (float)(int)srate)); }

template <class L>
__global__ void __launch_bounds__(64) zf_prepare(ZabBatch b) {   // dsp->init(sampleRate): instanceClear
  const int inst = blockIdx.x * b.ipw + threadIdx.x;
  if ((int)threadIdx.x >= b.ipw || inst >= b.n_inst) return;
#pragma unroll
  for (int k = 0; k < L::NSTATE; ++k) b.vars[k * b.var_se + inst * b.var_si] = 0.0;
  b.flags[inst] = ZAB_FLAG_PREPARED;
}

template <class L, int TT>
__global__ void __launch_bounds__(64) zf_process(ZabBatch b, ZabAudio a) {
  __shared__ float tile[L::NCH][64][TT + 1];
  const int lane = threadIdx.x;
  const int ipw = b.ipw;
  const int inst0 = blockIdx.x * ipw;
  const int inst = inst0 + lane;
  const bool active = lane < ipw && inst < b.n_inst;
  float st[L::NSTATE];
  float par[L::NPARAM];
  typename L::Ctl ctl;
  if (active) {
#pragma unroll
    for (int k = 0; k < L::NSTATE; ++k) st[k] = (float)b.vars[k * b.var_se + inst * b.var_si];
#pragma unroll
    for (int k = 0; k < L::NPARAM; ++k) par[k] = (float)b.sliders[k * b.sl_se + inst * b.sl_si];
    ctl = L::control(par, zf_sr(b.srate));     // zones are pushed before every compute(): per-launch constants
    b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
  }
  for (int64_t t0 = 0; t0 < a.frames; t0 += TT) {
    const int tn = (int)((a.frames - t0 < TT) ? (a.frames - t0) : TT);
    // eight HBM reads in flight per lane (one read per trip costs a full memory latency per tile element)
    for (int idx0 = lane; idx0 < ipw * L::NCH * TT; idx0 += 8 * 64) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + 64 * u;
        const int t = idx % TT, rc = idx / TT, ch = rc % L::NCH, row = rc / L::NCH;
        xv[u] = 0.0f;
        if (idx < ipw * L::NCH * TT && t < tn && inst0 + row < b.n_inst)
          xv[u] = a.in[((int64_t)(inst0 + row) * L::NCH + ch) * a.frame_stride + t0 + t];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + 64 * u;
        const int t = idx % TT, rc = idx / TT, ch = rc % L::NCH, row = rc / L::NCH;
        if (idx < ipw * L::NCH * TT) tile[ch][row][t] = xv[u];
      }
    }
    zf_wave_sync();
    if (active) {
      for (int t = 0; t < tn; ++t) {
        float x[L::NCH];
#pragma unroll
        for (int ch = 0; ch < L::NCH; ++ch) x[ch] = tile[ch][lane][t];
        L::frame(st, ctl, x);
#pragma unroll
        for (int ch = 0; ch < L::NCH; ++ch) tile[ch][lane][t] = x[ch];
      }
    }
    zf_wave_sync();
    for (int idx = lane; idx < ipw * L::NCH * TT; idx += 64) {
      const int t = idx % TT, rc = idx / TT, ch = rc % L::NCH, row = rc / L::NCH;
      if (t < tn && inst0 + row < b.n_inst)
        a.out[((int64_t)(inst0 + row) * L::NCH + ch) * a.frame_stride + t0 + t] = tile[ch][row][t];
    }
    zf_wave_sync();
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < L::NSTATE; ++k) b.vars[k * b.var_se + inst * b.var_si] = (double)st[k];
  }
}

template <class L> static hipError_t zf_launch_prepare(const ZabBatch* b, hipStream_t st) {
  hipLaunchKernelGGL(zf_prepare<L>, dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), 0, st, *b);
  return hipGetLastError();
}
template <class L> static hipError_t zf_launch_process(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  constexpr int TT = L::NCH <= 2 ? 64 : (L::NCH <= 4 ? 32 : 16);
  hipLaunchKernelGGL((zf_process<L, TT>), dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), 0, st, *b, *a);
  return hipGetLastError();
}
template <class L> static hipError_t zf_launch_slider(const ZabBatch*, hipStream_t) { return hipSuccess; }   // constants are per launch

#define ZF_DEFINE_MODULE(L, KEY)                                                                                        \
  static const ZabModule zf_module = {                                                                                  \
      ZAB_MODULE_ABI, KEY, L::NSTATE, L::NCH, L::NCH, L::NCH, 1, 0, 0, 1, 0, 64, L::names, 0, 0, 0, 0, 0,                     \
      zf_launch_prepare<L>, zf_launch_process<L>, zf_launch_slider<L>, nullptr, nullptr, nullptr, "zf_process", nullptr, nullptr};     \
  extern "C" const ZabModule* zab_module_get(void) { return &zf_module; }
// the same with a hand-written leaf kernel as the fast path
#define ZF_DEFINE_MODULE_FAST(L, KEY, APPLIES, LAUNCH, NAME)                                                            \
  static const ZabModule zf_module = {                                                                                  \
      ZAB_MODULE_ABI, KEY, L::NSTATE, L::NCH, L::NCH, L::NCH, 1, 0, 0, 1, 0, 64, L::names, 0, 0, 0, 0, 0,                     \
      zf_launch_prepare<L>, zf_launch_process<L>, zf_launch_slider<L>, APPLIES, LAUNCH, NAME, "zf_process", nullptr, nullptr};            \
  extern "C" const ZabModule* zab_module_get(void) { return &zf_module; }
