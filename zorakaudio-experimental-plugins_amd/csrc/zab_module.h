// zab_module.h -- internal contract between libzabatch.so (host runtime, C ABI) and a plugin module
// (libzab_<leaf>.so: zajit-generated section code + generic kernels [+ a hand-written leaf kernel]).
// Not part of the public boundary (include/zabatch.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

// One-time initialisations of a module are per DEVICE, not per process: __device__ tables and function attributes belong to the
// device that was current when they were set, and one process may hold engines on several GPUs (zab_group_*). Runs f() the
// first time it is reached with a given device current; f() must have COMPLETED its work when it returns (the device is marked
// done under the lock: a second engine on another stream takes the mark as "ready to use").
struct ZaPerDevice {
  std::mutex mu;
  uint64_t done[4] = {0, 0, 0, 0};
  template <class F> void once(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    uint64_t& w = done[(dev >> 6) & 3];
    if (!((w >> (dev & 63)) & 1ull)) { f(); w |= 1ull << (dev & 63); }
  }
};

#define ZAB_MODULE_ABI 14

enum { ZAB_FLAG_SLIDER_DIRTY = 1u, ZAB_FLAG_PREPARED = 2u };

// Device-resident state of N instances of one leaf. Plain strides so one generated kernel serves both layouts:
//   interleaved   (generic default): element e of instance i at base[e * n_pad + i]        (se = n_pad, si = 1)
//   instance-major (leaf kernels):   element e of instance i at base[i * extent + e]       (se = 1, si = extent)
struct ZabBatch {
  int32_t n_inst;        // live instances
  int32_t n_pad;         // allocation width, multiple of 64
  int32_t instance_major;
  int32_t nvars;
  double* vars;   int64_t var_se, var_si;      // DSPJSFX_State::vars
  double* sliders; int64_t sl_se, sl_si;       // DSPJSFX_State::sliders[64]
  double* spl;                                 // DSPJSFX_State::spl[64], strides as sliders
  double* mem;    int64_t mem_se, mem_si, mem_cap;   // DSPJSFX_State::mem / memN
  uint32_t* mt;   int64_t mt_se, mt_si;        // randMT[624]
  uint32_t* mti;                               // randIndex            [n_pad]
  int64_t* mem_high;                           // write high-water     [n_pad]
  int64_t* mem_need;                           // capacity wanted      [n_pad]
  uint32_t* err;                               // ZA_ERR_* bits        [n_pad]
  uint32_t* flags;                             // ZAB_FLAG_*           [n_pad]
  uint64_t* pend;                              // pendingSlider{Change,Automate,AutomateEnd}Mask  [3][n_pad], then [n_pad]:
                                               // their OR over the blocks since the host last consumed it
  uint64_t* vis_mask;                          // sliderVisibleMask    [n_pad]
  int32_t* vis_init;                           // sliderVisibilityInit [n_pad]
  double srate;
  uint64_t first_id;
  const void* gmem;                            // ZaGmemView* (device) or null
  const void* pool;                            // ZaPoolView* (device) or null
  double* fft;    int64_t fft_se, fft_si, fft_cap;   // FFT builtin scratch (null / 0 when unused)
  uint32_t* gmem_att;                          // per-instance "gmem attached" flag [n_pad] (null when unused)
  uint64_t epoch;                              // bumped by the runtime whenever host calls may have changed state
  const void* files;                           // ZaFileView* (device) or null
  int64_t* fh;    int64_t fh_se, fh_si;        // per-instance file handle words [ZA_FH_WORDS] (null when unused)
  const void* bus;                             // ZaBusView* (device) or null
  int32_t ipw;                                 // instances per wavefront of the lane-per-instance kernels (1..64, power of
                                               // two): instance i runs in lane i % ipw of workgroup i / ipw
  int32_t lmem_words;                          // process kernel keeps mem[0, lmem_words) of its instances in LDS (0 = off);
                                               // chosen by the runtime after prepare from the instances' arena footprint
  int64_t* resume;                             // [n_pad] frames of the current launch already done by a time-parallel kernel
                                               // (zajit/tpar.py): == frames normally; less when it handed an instance back to
                                               // the generic code, which then finishes the launch from there
};

struct ZabAudio {
  const float* in;       // planar [n_inst][nch][frame_stride]
  float* out;
  int64_t frames;
  int64_t frame_stride;
  int32_t block;         // host block size (jsfx_process_block numSamples), last block short
};

struct ZabModule {
  int32_t abi;
  const char* name;
  int32_t nvars, nch, n_in, n_out;
  int32_t has_init, has_slider, has_block, has_sample;
  int32_t prefer_instance_major;
  int64_t default_mem_cap;
  const char* const* var_names;      // [nvars], index order
  int64_t fft_scratch_doubles;       // per-instance scratch the runtime must provide (0: leaf has no FFT builtins)
  int32_t uses_gmem;                 // runtime must provide a gmem segment; 2 = instances start attached (options:gmem=)
  int32_t uses_pool;                 // leaf reads the sample pool (zab_pool_upload provides it)
  int32_t uses_files;                // leaf calls file_*(): runtime provides slots + per-instance handle words
  int32_t uses_msg;                  // leaf calls msg_*(): runtime provides the bus, runs host block by host block and
                                     // calls launch_msg_flush after each (at most 256 instances per engine)
  // generic (translator-generated) kernels
  hipError_t (*launch_prepare)(const ZabBatch*, hipStream_t);
  hipError_t (*launch_process)(const ZabBatch*, const ZabAudio*, hipStream_t);
  hipError_t (*launch_slider)(const ZabBatch*, hipStream_t);   // @slider on instances flagged ZAB_FLAG_SLIDER_DIRTY
  // optional hand-written leaf kernel; applies() decides from host-visible config, launch may still defer
  // per instance to the generic kernel through `fallback_mask` semantics documented by the leaf
  int32_t (*fast_applies)(const ZabBatch*, const ZabAudio*);
  hipError_t (*launch_fast)(const ZabBatch*, const ZabAudio*, hipStream_t);
  const char* fast_kernel_name;
  const char* generic_kernel_name;
  // one raw section (0 init, 1 slider, 2 block, 3 sample) on every instance; null for leaves without sections (Faust)
  hipError_t (*launch_section)(const ZabBatch*, int which, double samplesblock, hipStream_t);
  hipError_t (*launch_msg_flush)(const ZabBatch*, hipStream_t);   // end of a host block: every outbox to the bus ring
  int32_t lmem_ok;                   // process kernel honours ZabBatch::lmem_words (leaf touches mem[] only through zart.h)
};

extern "C" const ZabModule* zab_module_get(void);
