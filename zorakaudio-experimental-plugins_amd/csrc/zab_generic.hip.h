// zab_generic.hip.h -- translator back end: the generic gfx950 kernels every leaf module gets.
//
// Included at the end of a generated module source, after the ZA_* defines, zart.h and the zajit section code.
// Mapping: ONE LANE PER INSTANCE, b.ipw (<= 64) instances per single-wave workgroup. A wave costs the same whether 1 or 64
// of its lanes are live, so small batches are spread thin (ipw = 1 at 1024 instances: 1024 wavefronts, one per SIMD of
// the chip, instead of 16 full ones) and only batches beyond ~64 Ki instances fill every lane. Each lane keeps its DSPJSFX_State
// (vars / used sliders / used spl) in registers for the whole launch and walks the host blocks serially, exactly
// as jsfx_process_block does (dsp_jsfx_aot.py:5713-5905):
//     samplesblock = n; @block; if any pending slider mask -> @slider; for each frame: f32->f64 spl[], @sample,
//     f64->f32 out.
// Audio is instance-major planar (include/zabatch.h), so a wave stages a [64 instances][NCH][TT frames] tile
// through LDS: global reads/writes are 128-byte row segments (coalesced), the per-lane walk over its own row is
// conflict-free thanks to the +1 padding (bank = (row*(TT+1)+t) mod 32).
// vars/sliders/mem use the strides in ZabBatch, so the same code serves the interleaved layout (lanes of a wave
// touch consecutive addresses whenever instances share control flow and indices) and the instance-major one.
#pragma once

#include "zab_module.h"

#include <mutex>

#ifndef ZA_NCH
#error "generated defines missing"
#endif

#if ZA_NCH <= 4
#define ZA_TT 32
#elif ZA_NCH <= 8
#define ZA_TT 16
#elif ZA_NCH <= 16
#define ZA_TT 8
#elif ZA_NCH <= 32
#define ZA_TT 4
#else
#define ZA_TT 2
#endif

typedef ZaState<ZA_NV> ZaS;
#if ZA_USES_LMEM
typedef ZaState<ZA_NV, true> ZaSLm;     // the process kernel's second body: mem[0, lmem_words) in LDS (zart.h)
#endif

// Phase boundary inside a single-wave workgroup: a wave's LDS accesses execute in order, so only the compiler has to be
// kept from reordering them; __syncthreads() would also wait for the tile's output stores to reach memory (vmcnt(0)).
__device__ __forceinline__ void za_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Leaves with FFT builtins: their transforms are latency-bound (one wavefront per instance walks a chain of dependent memory and
// LDS round trips), so resident wavefronts are what hides it. Left alone the compiler takes 270 - 290 registers for these
// kernels -- one wavefront per SIMD; capped at 256 it is two (measured: tools/fft_bench.py).
#if ZA_USES_FFT && !defined(ZA_WAVES_PER_EU)
#define ZA_WAVES_PER_EU ZA_FFT_WAVES_PER_EU      /* zart_fft.h */
#endif
#if defined(ZA_WAVES_PER_EU) && ZA_WAVES_PER_EU > 0
#define ZA_OCC __attribute__((amdgpu_waves_per_eu(ZA_WAVES_PER_EU)))
#else
#define ZA_OCC
#endif

#ifndef ZA_KERNEL_ENTRY
#define ZA_KERNEL_ENTRY() (void)0      /* leaves with FFT builtins reset their LDS twiddle flag here (zart_fft.h) */
#endif

template <class SS>
__device__ __forceinline__ void za_state_bind(SS& s, const ZabBatch& b, int inst) {
  s.srate = b.srate;
  s.samplesblock = 0.0;
  s.midi_bus = 0.0;
  s.ext_midi_bus = 0.0;
  s.mem = b.mem + (int64_t)inst * b.mem_si;
#if defined(ZA_MEM_STRIDE1)
  // (replica-lane leaves are always laid out instance-major -- zajit/build.py sets this for them, zabatch.hip keeps the layout:
  //  the 64-bit multiply of every arena address by a run-time stride of 1 is four quarter-rate instructions per access)
  s.mem_stride = 1;
#else
  s.mem_stride = b.mem_se;
#endif
  s.mem_cap = b.mem_cap;
  s.mem_high = b.mem_high[inst];
  s.mem_need = b.mem_need[inst];
  s.mt = b.mt + (int64_t)inst * b.mt_si;
  s.mt_stride = b.mt_se;
  s.mti = b.mti[inst];
  s.err = b.err[inst];
  s.pend_change = b.pend[inst];
  s.pend_automate = b.pend[b.n_pad + inst];
  s.pend_automate_end = b.pend[2 * b.n_pad + inst];
  s.vis_mask = b.vis_mask[inst];
  s.vis_init = b.vis_init[inst];
  s.block_size = 0;
  s.gmem = (const ZaGmemView*)b.gmem;
  s.pool = (const ZaPoolView*)b.pool;
  s.instance_id = b.first_id + (uint64_t)inst;
  s.sink = 0.0;
  s.memtop = ZA_MEMTOP;
  s.fft = b.fft ? b.fft + (int64_t)inst * b.fft_si : nullptr;
#if defined(ZA_MEM_STRIDE1)
  s.fft_stride = 1;
#else
  s.fft_stride = b.fft_se;
#endif
  s.fft_cap = b.fft ? b.fft_cap : 0;
  s.gmem_attached = b.gmem_att ? b.gmem_att[inst] : 0;
  s.replica = 0;
  s.bus = (const ZaBusView*)b.bus;
  s.inst_index = (uint32_t)inst;
  s.files = (const ZaFileView*)b.files;
  s.fh = b.fh ? b.fh + (int64_t)inst * b.fh_si : nullptr;
  s.fh_stride = b.fh_se;
  s.lm_words = s.lm_stride = s.lm_off = 0;
  s.rep_i = 0; s.rep_n = 1; s.rep_stride = 64;
}

template <class SS>
__device__ __forceinline__ void za_state_load(SS& s, const ZabBatch& b, int inst) {
  za_state_bind(s, b, inst);
  // (Beyond ~1000 variables the compiler declines to unroll this loop and the one in za_state_store -- its limit for "#pragma
  // unroll" is 16 K instructions -- and one dynamic index into the state object keeps all of it in scratch memory. Measured on
  // Texture, 1087 variables: forcing the unroll (-mllvm -pragma-unroll-threshold) costs 14 minutes of compile time and the
  // register-allocated form with its 23 KB of spills runs SLOWER, 973 vs 601 ms at 192 instances. Left as it is.)
#pragma unroll
  for (int k = 0; k < ZA_NV; ++k) s.v[k] = b.vars[k * b.var_se + inst * b.var_si];
#define ZA_X(k) s.sl[k] = b.sliders[(k) * b.sl_se + inst * b.sl_si];
  ZA_FOR_USED_SL(ZA_X)
#undef ZA_X
#define ZA_X(k) s.spl[k] = b.spl[(k) * b.sl_se + inst * b.sl_si];
  ZA_FOR_USED_SPL(ZA_X)
#undef ZA_X
}

template <class SS>
__device__ __forceinline__ void za_state_store(const SS& s, const ZabBatch& b, int inst) {
#pragma unroll
  for (int k = 0; k < ZA_NV; ++k) b.vars[k * b.var_se + inst * b.var_si] = s.v[k];
#define ZA_X(k) b.sliders[(k) * b.sl_se + inst * b.sl_si] = s.sl[k];
  ZA_FOR_USED_SL(ZA_X)
#undef ZA_X
#define ZA_X(k) b.spl[(k) * b.sl_se + inst * b.sl_si] = s.spl[k];
  ZA_FOR_USED_SPL(ZA_X)
#undef ZA_X
  b.mem_high[inst] = s.mem_high;
  b.mem_need[inst] = s.mem_need;
  b.mti[inst] = s.mti;
  b.err[inst] = s.err;
  b.pend[inst] = s.pend_change;
  b.pend[b.n_pad + inst] = s.pend_automate;
  b.pend[2 * b.n_pad + inst] = s.pend_automate_end;
  b.vis_mask[inst] = s.vis_mask;
  b.vis_init[inst] = s.vis_init;
  if (b.gmem_att) b.gmem_att[inst] = s.gmem_attached;
}

// sliderN:var=... aliases: the host keeps the named var equal to the slider (src/JSFXJuceProcessor.cpp:9349-9353)
template <class SS>
__device__ __forceinline__ void za_alias_sync(SS& s) {
#define ZA_X(sl_idx, var_idx) s.v[var_idx] = s.sl[sl_idx];
  ZA_FOR_ALIAS(ZA_X)
#undef ZA_X
}

// prepareToPlay(): resetStateStructOnly (vars/spl zeroed, mem kept) -> sliders already pushed -> @init ->
// alias re-apply -> @slider.   src/JSFXJuceProcessor.cpp:3251,3302-3318
extern "C" __global__ void __launch_bounds__(64) ZA_KERNEL(prepare)(ZabBatch b) {
  ZA_KERNEL_ENTRY();
  const int inst = blockIdx.x * b.ipw + threadIdx.x;
  if ((int)threadIdx.x >= b.ipw || inst >= b.n_inst) return;
  ZaS s;
  za_state_bind(s, b, inst);
#pragma unroll
  for (int k = 0; k < ZA_NV; ++k) s.v[k] = 0.0;
#define ZA_X(k) s.sl[k] = b.sliders[(k) * b.sl_se + inst * b.sl_si];
  ZA_FOR_USED_SL(ZA_X)
#undef ZA_X
#define ZA_X(k) s.spl[k] = 0.0;
  ZA_FOR_USED_SPL(ZA_X)
#undef ZA_X
  s.mti = 0;
  s.pend_change = s.pend_automate = s.pend_automate_end = 0;
  s.vis_mask = 0;
  s.vis_init = 0;
  s.gmem_attached = ZA_GMEM_AUTOATTACH;
  za_alias_sync(s);
  za_section_init(s);
  za_alias_sync(s);
  za_section_slider(s);
  za_state_store(s, b, inst);
  b.flags[inst] = ZAB_FLAG_PREPARED;
}

#if (ZA_USES_FFT || ZA_USES_COOP) && !ZA_USES_GMEM
#define ZA_REPLICAS 1
#endif
// The audio tile: [NCH][rows][TT + 1] floats in LDS. Lane-per-instance leaves hold 64 rows (static). A replica-lane leaf only
// ever fills ipw rows -- two, typically -- and its wavefronts also hold a transform buffer, so its tile is sized at launch
// (dynamic LDS, behind the arena window if there is one): 17 KB -> 0.5 KB per wavefront, six waves per CU instead of three.
#ifdef ZA_REPLICAS
#define ZA_TILE_ROWS ipw
#else
#define ZA_TILE_ROWS 64
#endif
#define ZA_TILE(ch, row, t) tile[((ch) * ZA_TILE_ROWS + (row)) * (ZA_TT + 1) + (t)]
template <class SS>
__device__ __forceinline__ void za_process_body(const ZabBatch& b, const ZabAudio& a, float* tile) {
  const int lane = threadIdx.x;
  const int ipw = b.ipw;
  const int inst0 = blockIdx.x * ipw;
#ifdef ZA_REPLICAS
  // REPLICA LANES: a thin wavefront (ipw < 64) runs every instance on 64 / ipw lanes at once -- identical state, identical
  // control flow, identical (hence harmless) stores. Nothing is gained for the serial code, but every lane now reaches the
  // FFT builtins, whose wave-cooperative form (zart_fft.h) spreads ONE instance's transform over all of them.
  const int row = lane % ipw;
  const bool primary = lane < ipw;
  const int inst = inst0 + row;
  const bool active = inst < b.n_inst;
#else
  const int row = lane;
  const bool primary = true;
  const int inst = inst0 + lane;
  const bool active = lane < ipw && inst < b.n_inst;
#endif
  SS s;
  uint64_t pend_seen = 0;           // slider masks the script raised in any block of this launch (host: consumeDspSliderChanges)
  if (active) {
    za_state_load(s, b, inst);
    s.replica = primary ? 0u : 1u;
#ifdef ZA_REPLICAS
    s.rep_i = (uint32_t)(lane / ipw); s.rep_n = (uint32_t)(64 / ipw); s.rep_stride = (uint32_t)ipw;    // (accumulation loops, zart.h)
#endif
#if ZA_USES_LMEM
    if (SS::kLm) {                  // LDS window over mem[0, lmem_words) (zart.h): load it, eight reads in flight
      s.lm_stride = (uint32_t)ipw;
      s.lm_off = (uint32_t)row;         // (replica lanes share their instance's words)
      const uint32_t K = (uint32_t)b.lmem_words;      // multiple of 8 (runtime), <= mem_cap
      for (uint32_t a0 = 0; a0 < K; a0 += 8) {
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = s.mem[(int64_t)(a0 + u) * s.mem_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) za_lmem[(a0 + u) * s.lm_stride + s.lm_off] = w[u];
      }
      s.lm_words = K;
    }
#endif
    if (b.flags[inst] & ZAB_FLAG_SLIDER_DIRTY) {      // processBlock: sliders changed -> jsfx_slider (:3545-3547)
      za_alias_sync(s);
      za_section_slider(s);
      b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
    }
  }
  for (int64_t pos = 0; pos < a.frames; pos += a.block) {
    const int n = (int)((a.frames - pos < a.block) ? (a.frames - pos) : a.block);
    if (active) {
      s.samplesblock = (double)n;
      s.block_size = n;
#if ZA_USES_MSG
      za_msg_begin_block(s);        // DspJsfxRuntime::beginBlock: this block's ready inbox
#endif
      za_section_block(s);
      if (s.pend_change | s.pend_automate | s.pend_automate_end) za_section_slider(s);
    }
#if ZA_HAS_SAMPLE && ZA_NCH > 0
    for (int t0 = 0; t0 < n; t0 += ZA_TT) {
      const int tn = (n - t0 < ZA_TT) ? (n - t0) : ZA_TT;
      // eight HBM reads in flight per lane (one read per trip costs a full memory latency per tile element)
      for (int idx0 = lane; idx0 < ipw * ZA_NCH * ZA_TT; idx0 += 8 * 64) {
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = idx0 + 64 * u;
          const int t = idx % ZA_TT, rc = idx / ZA_TT, ch = rc % ZA_NCH, row = rc / ZA_NCH;
          xv[u] = 0.0f;
          if (idx < ipw * ZA_NCH * ZA_TT && t < tn && inst0 + row < b.n_inst)
            xv[u] = a.in[((int64_t)(inst0 + row) * ZA_NCH + ch) * a.frame_stride + pos + t0 + t];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = idx0 + 64 * u;
          const int t = idx % ZA_TT, rc = idx / ZA_TT, ch = rc % ZA_NCH, row = rc / ZA_NCH;
          if (idx < ipw * ZA_NCH * ZA_TT) ZA_TILE(ch, row, t) = xv[u];
        }
      }
      za_wave_sync();
      if (active) {
        for (int t = 0; t < tn; ++t) {
#define ZA_X(ch) s.spl[ch] = (double)ZA_TILE(ch, row, t);
          ZA_FOR_CH(ZA_X)
#undef ZA_X
          za_section_sample(s);
#define ZA_X(ch) ZA_TILE(ch, row, t) = (float)s.spl[ch];
          ZA_FOR_CH(ZA_X)
#undef ZA_X
        }
      }
      za_wave_sync();
      for (int idx = lane; idx < ipw * ZA_NCH * ZA_TT; idx += 64) {
        const int t = idx % ZA_TT, rc = idx / ZA_TT, ch = rc % ZA_NCH, row = rc / ZA_NCH;
        if (t < tn && inst0 + row < b.n_inst)
          a.out[((int64_t)(inst0 + row) * ZA_NCH + ch) * a.frame_stride + pos + t0 + t] = ZA_TILE(ch, row, t);
      }
      za_wave_sync();
    }
#endif
    if (active) {                   // consumeDspSliderChanges (:3745): the host mirror collects them after the launch
      pend_seen |= s.pend_change | s.pend_automate | s.pend_automate_end;
      s.pend_change = s.pend_automate = s.pend_automate_end = 0;
    }
  }
#if ZA_USES_LMEM
  if (SS::kLm && active && primary) {   // write the stored part of the window back (words at or above mem_high never changed)
    const int64_t top = s.mem_high < (int64_t)s.lm_words ? s.mem_high : (int64_t)s.lm_words;
    for (int64_t a = 0; a < top; ++a) s.mem[a * s.mem_stride] = za_lmem[(uint32_t)a * s.lm_stride + s.lm_off];
  }
#endif
  if (active && primary) {
    za_state_store(s, b, inst);
    if (pend_seen) b.pend[3 * (int64_t)b.n_pad + inst] |= pend_seen;
  }
}

extern "C" __global__ void __launch_bounds__(64) ZA_OCC ZA_KERNEL(process)(ZabBatch b, ZabAudio a) {
  ZA_KERNEL_ENTRY();
#if ZA_NCH > 0 && defined(ZA_REPLICAS)
  extern __shared__ double za_dyn_lds[];                 // [arena window: lmem_words x ipw doubles][tile]
  float* const tp = (float*)(za_dyn_lds + (size_t)b.lmem_words * (size_t)b.ipw);
#elif ZA_NCH > 0
  __shared__ float tile[ZA_NCH * 64 * (ZA_TT + 1)];
  float* const tp = tile;
#else
  float* const tp = nullptr;
#endif
#if ZA_USES_LMEM
  if (b.lmem_words > 0) { za_process_body<ZaSLm>(b, a, tp); return; }     // (uniform: a launch runs one body or the other)
#endif
  za_process_body<ZaS>(b, a, tp);
}

// processBlock prologue for the hand-written kernels: instances whose sliders changed run @slider first (:3545-3547).
extern "C" __global__ void __launch_bounds__(64) ZA_KERNEL(slider)(ZabBatch b) {
  ZA_KERNEL_ENTRY();
  const int inst = blockIdx.x * b.ipw + threadIdx.x;
  if ((int)threadIdx.x >= b.ipw || inst >= b.n_inst) return;
  if (!(b.flags[inst] & ZAB_FLAG_SLIDER_DIRTY)) return;
  ZaS s;
  za_state_load(s, b, inst);
  za_alias_sync(s);
  za_section_slider(s);
  za_state_store(s, b, inst);
  b.flags[inst] &= ~ZAB_FLAG_SLIDER_DIRTY;
}

// Single-section entry for the jsfx_init/jsfx_slider/jsfx_block/jsfx_sample compatibility shim (include/zabatch.h,
// zab_run_section): the raw section on the state as it stands, no reset, no float conversion -- what a direct call of
// the reference's generated section function does (dsp_jsfx_aot.py:4188).
extern "C" __global__ void __launch_bounds__(64) ZA_KERNEL(section)(ZabBatch b, int which, double samplesblock) {
  ZA_KERNEL_ENTRY();
  const int inst = blockIdx.x * b.ipw + threadIdx.x;
  if ((int)threadIdx.x >= b.ipw || inst >= b.n_inst) return;
  ZaS s;
  za_state_load(s, b, inst);
  s.samplesblock = samplesblock;
  s.block_size = (int)samplesblock;
  switch (which) {
    case 0: za_section_init(s); break;
    case 1: za_section_slider(s); break;
    case 2: za_section_block(s); break;
    default: za_section_sample(s); break;
  }
  za_state_store(s, b, inst);
}
static void za_fft_tables_once(hipStream_t st);
static hipError_t za_launch_section(const ZabBatch* b, int which, double samplesblock, hipStream_t st) {
  za_fft_tables_once(st);
  hipLaunchKernelGGL(ZA_KERNEL(section), dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), 0, st, *b, which, samplesblock);
  return hipGetLastError();
}

#if ZA_USES_MSG
// DspJsfxRuntime::endBlock of every instance, in instance order (one wavefront: the ring order must be the serial one)
extern "C" __global__ void __launch_bounds__(64) ZA_KERNEL(msgflush)(ZabBatch b) { za_msg_flush_all((const ZaBusView*)b.bus); }
static hipError_t za_launch_msg_flush(const ZabBatch* b, hipStream_t st) {
  hipLaunchKernelGGL(ZA_KERNEL(msgflush), dim3(1), dim3(64), 0, st, *b);
  return hipGetLastError();
}
#endif

static hipError_t za_launch_slider(const ZabBatch* b, hipStream_t st) {
  za_fft_tables_once(st);
  hipLaunchKernelGGL(ZA_KERNEL(slider), dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), 0, st, *b);
  return hipGetLastError();
}
// The FFT builtins' twiddle / permutation tables are __device__ globals of the module (one copy per GPU), filled once per device
// before the first kernel that can reach a builtin. EVERY launch path asks: the single-instance jsfx_* shim enters through
// za_launch_section / za_launch_process and never calls prepare -- with the tables only filled there, a host whose first FFT
// leaf ran through the shim transformed with zeros (round 4: tests/test_shim.py[DOT] run on its own; in the whole suite an
// engine of the same module had always prepared first).
static void za_fft_tables_once(hipStream_t st) {
#if ZA_USES_FFT
  static ZaPerDevice za_fft_once;
  za_fft_once.once([st] {       // (waited for here: another engine of this leaf on the same device runs on its own stream and must
                                //  not read the tables before this kernel has written them; once per device and module)
    hipLaunchKernelGGL(za_fft_table_kernel, dim3(ZA_FFT_MAX / 2 / 256), dim3(256), 0, st);
    (void)hipStreamSynchronize(st);
  });
#else
  (void)st;
#endif
}
static hipError_t za_launch_prepare(const ZabBatch* b, hipStream_t st) {
  za_fft_tables_once(st);
  hipLaunchKernelGGL(ZA_KERNEL(prepare), dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), 0, st, *b);
  return hipGetLastError();
}
static hipError_t za_launch_process(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {
  za_fft_tables_once(st);
  size_t lds = 0;
#if ZA_USES_LMEM
  lds = (size_t)b->lmem_words * (size_t)b->ipw * sizeof(double);
#endif
#if ZA_NCH > 0 && defined(ZA_REPLICAS)
  lds += (size_t)ZA_NCH * (size_t)b->ipw * (ZA_TT + 1) * sizeof(float);      // the audio tile (za_process_body)
#endif
#if ZA_USES_LMEM
  static size_t za_lds_allowed[256];                // dynamic LDS beyond the default limit has to be asked for, per device
  static std::mutex za_lds_mu;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(za_lds_mu);
    if (lds > za_lds_allowed[dev & 255]) {
      const hipError_t e = hipFuncSetAttribute((const void*)ZA_KERNEL(process), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      za_lds_allowed[dev & 255] = lds;
    }
  }
#endif
  hipLaunchKernelGGL(ZA_KERNEL(process), dim3((b->n_inst + b->ipw - 1) / b->ipw), dim3(64), lds, st, *b, *a);
  return hipGetLastError();
}
