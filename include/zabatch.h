/* zabatch.h -- C ABI of the MI355X batch audio-DSP engine (libzabatch.so).
 *
 * Drop-in boundary for ONE path of ZorakAudio-Experimental-Plugins: the generated-object interface
 * `jsfx_init / jsfx_slider / jsfx_block / jsfx_sample / jsfx_process_block (DSPJSFX_State*, ...)`
 * that dsp_jsfx_aot.py emits per plugin (prototype: dsp_jsfx_aot.py:6090-6102; struct: :5991-6025) and that
 * JSFXJuceProcessor::prepareToPlay / processBlock call (src/JSFXJuceProcessor.cpp:3305,3318,3547,3733).
 * The reference has no batch axis -- one DSPJSFX_State per plugin instance, one audio thread each. Here N
 * instances of one leaf are one object on one GPU and every entry point acts on all of them:
 *
 *   reference (per instance)                                  this ABI (N instances)
 *   --------------------------------------------------------  ------------------------------------------
 *   DSPJSFX_State st; calloc mem (JSFXJuceProcessor.cpp:8958)  zab_create()
 *   pushParamsToStateSliders() (:9286-9357)                    zab_set_sliders()
 *   jsfx_init(&st); alias re-apply; jsfx_slider(&st) (:3305-18) zab_prepare()
 *   jsfx_slider(&st) when sliders changed (:3545-3547)         implicit in zab_process() for touched instances
 *   jsfx_process_block(&st, in, out, nCh, n) (:3733)           zab_process()
 *   st.vars[DSPJSFX_VARS[i].index], st.mem[...]                zab_read_vars() / zab_read_mem()
 *
 * All functions return ZAB_OK (0) or a negative ZAB_E_* code; zab_last_error() gives the text. Nothing here
 * takes or returns a torch/JUCE/C++ type. Device pointers are plain `void*` HIP device addresses.
 * A compute call made without a GPU, or with a missing plugin module, fails loudly -- there is no CPU fallback.
 */
#ifndef ZABATCH_H
#define ZABATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zab_engine zab_engine;

enum {
  ZAB_OK = 0,
  ZAB_E_ARG = -1,          /* bad argument */
  ZAB_E_MODULE = -2,       /* plugin module missing / not loadable / ABI mismatch */
  ZAB_E_HIP = -3,          /* HIP runtime error (no device, allocation, launch) */
  ZAB_E_MEM_OVERFLOW = -4, /* a script stored to mem[] beyond the arena: recreate with a larger mem_cap */
  ZAB_E_UNSUPPORTED = -5,  /* script reached a host-only builtin (MIDI, file, msg_*) on the device */
  ZAB_E_LOOP_CAP = -6,     /* a loop()/while ran past the safety cap and was cut */
  ZAB_E_STATE = -7         /* call sequence error (process before prepare, ...) */
};

/* which kernels zab_process may use */
enum {
  ZAB_PATH_AUTO = 0,       /* hand-written leaf kernel when the module has one and it applies, else generic */
  ZAB_PATH_GENERIC = 1,    /* translator-generated serial-per-instance kernel (one lane per instance) */
  ZAB_PATH_FAST = 2        /* require the hand-written kernel; error if it does not apply */
};

/* audio buffer placement for zab_process */
enum {
  ZAB_BUF_DEVICE = 0,      /* in/out are device pointers (HBM resident) */
  ZAB_BUF_HOST = 1         /* in/out are host pointers; staged over PCIe by the call */
};

typedef struct zab_config {
  int32_t n_instances;     /* batch size N (> 0) */
  int32_t device;          /* HIP device ordinal */
  double srate;            /* engine sample rate (st.srate) */
  int32_t max_block;       /* largest host block `block` that zab_process will be given */
  int32_t path;            /* ZAB_PATH_* */
  int64_t mem_cap;         /* doubles of mem[] per instance; 0 = module default (65536, the reference's initial
                              calloc, src/JSFXJuceProcessor.cpp:8958-8963) */
  uint64_t first_instance_id; /* instance_id() of instance 0; instance i reports first_instance_id + i */
} zab_config;

typedef struct zab_info {
  char name[64];
  int32_t nvars;           /* entries of vars[] (DSPJSFX_State::vars) */
  int32_t n_channels;      /* channels processed per instance (DSPJSFX_PROCESS_CHANNELS) */
  int32_t n_inputs, n_outputs;
  int32_t has_init, has_slider, has_block, has_sample;
  int32_t has_fast_path;
  int64_t mem_cap;
  int32_t n_instances;
  int32_t layout_instance_major; /* 1: mem/vars of one instance contiguous; 0: interleaved across instances */
} zab_info;

const char* zab_last_error(void);
int zab_abi_version(void);        /* the plugin-module ABI (runtime <-> libzab_<leaf>.so) */
int zab_host_abi_version(void);   /* the host ABI: ZAB_HOST_ABI of the header this library was built from */

/* Load plugin module `libzab_<name>.so` (path or bare leaf name resolved next to libzabatch.so) and allocate
 * state for cfg->n_instances instances: vars/sliders/spl zeroed, mem zeroed (resetStateStructOnly + calloc). */
int zab_create(const char* module, const zab_config* cfg, zab_engine** out);
int zab_destroy(zab_engine* e);
int zab_get_info(zab_engine* e, zab_info* out);

/* name <-> vars[] index table of the leaf (DSPJSFX_VARS, dsp_jsfx_aot.py:6028-6046). */
int zab_var_count(zab_engine* e);
const char* zab_var_name(zab_engine* e, int index);
int zab_var_index(zab_engine* e, const char* name);   /* -1 if unknown */

/* Slider values as the host bridge would write them into st.sliders[] (already clamped/quantised by the caller,
 * zajit.sliders mirrors src/JSFXJuceProcessor.cpp:5556-5596). values = count x 64 doubles for instances
 * [first, first+count); count = 0 with first = 0 broadcasts one row of 64 to every instance.
 * Instances whose values differ from what the host pushed last (the reference's lastSliders comparison in
 * pushParamsToStateSliders, :9286-9357 -- not from what a script may have written into its sliders meanwhile) run @slider
 * at the start of the next zab_process (processBlock :3545-3547). The values always overwrite the device's. */
int zab_set_sliders(zab_engine* e, int32_t first, int32_t count, const double* values);
int zab_get_sliders(zab_engine* e, int32_t first, int32_t count, double* values);
/* consumeDspSliderChanges() (src/JSFXJuceProcessor.cpp:5665-5739) for instances [first, first+count): masks[i] receives the
 * OR of the pendingSlider{Change,Automate,AutomateEnd} masks the scripts raised (sliderchange / slider_automate) since the
 * previous call and the device-side record is cleared; values receives the 64 current slider values of each instance. For the
 * sliders named in a mask the engine also takes the script's value as "last pushed" (lastSliders / internalSliderShadow in
 * the reference), so that a host which pushes it back does not trigger @slider again. Either pointer may be null. */
int zab_consume_slider_changes(zab_engine* e, int32_t first, int32_t count, uint64_t* masks, double* values);

/* prepareToPlay(): @init, slider-alias re-apply, @slider on every instance (sliders must be set before). */
int zab_prepare(zab_engine* e);

/* processBlock() x ceil(frames/block): for each host block of `block` frames (last one short) run
 * jsfx_process_block semantics on every instance. Audio is planar float32, instance-major:
 *   sample (i, ch, t) at buf[(i * n_channels + ch) * frame_stride + t],  0 <= t < frames <= frame_stride.
 * out may equal in (Faust-style in-place). placement = ZAB_BUF_*. Asynchronous on the engine's stream when
 * placement is ZAB_BUF_DEVICE; zab_sync() waits and reports device-side errors. */
int zab_process(zab_engine* e, const void* in, void* out, int64_t frames, int64_t frame_stride, int32_t block,
                int32_t placement);
int zab_sync(zab_engine* e);

/* State read-back (synchronises). vars: [count][nvars]; mem: [count][n] doubles starting at element `start`. */
int zab_read_vars(zab_engine* e, int32_t first, int32_t count, double* dst);
int zab_read_mem(zab_engine* e, int32_t first, int32_t count, int64_t start, int64_t n, double* dst);
int zab_write_mem(zab_engine* e, int32_t first, int32_t count, int64_t start, int64_t n, const double* src);
int zab_read_mem_high(zab_engine* e, int32_t first, int32_t count, int64_t* dst); /* write high-water marks */

/* Single-instance state exchange in the layout of the reference's DSPJSFX_State (dsp_jsfx_aot.py:5991-6025) and raw
 * section calls: what a per-leaf shim exporting jsfx_init / jsfx_slider / jsfx_block / jsfx_sample / jsfx_process_block
 * (prototypes dsp_jsfx_aot.py:6088-6102) needs so that src/JSFXJuceProcessor.cpp can link against this engine unchanged
 * (zajit/shim.py generates that shim). Null pointers skip a field. Both calls synchronise.
 * MIDI queues and runtimeOpaque are host-side objects and are not mirrored. */
/* ZAB_HOST_ABI counts layout changes of the structs a HOST passes by pointer (zab_config, zab_info, zab_host_state). A client
 * compiled against another header must not call into this library: compare with zab_host_abi_version() once after loading
 * (the generated shims do, zajit/shim.py). zab_host_state also says how large the caller's struct is, so that fields added at
 * its end are only touched when the caller has them. */
#define ZAB_HOST_ABI 2
typedef struct zab_host_state {
  uint64_t struct_size;           /*        sizeof(zab_host_state) as the CALLER compiled it (0 is refused) */
  double* spl;                    /* [64]   DSPJSFX_State::spl */
  double* sliders;                /* [64]   ::sliders */
  double* vars;                   /* [nvars] ::vars */
  double* mem;                    /* [mem_n] ::mem (host allocation, src/JSFXJuceProcessor.cpp:8958-8963) */
  int64_t mem_n;                  /*        ::memN; at most zab_info.mem_cap cells are exchanged */
  int64_t* pending_masks;         /* [3]    pendingSlider{Change,Automate,AutomateEnd}Mask */
  uint32_t* rand_mt;              /* [624]  ::randMT */
  uint32_t* rand_index;           /*        ::randIndex */
  int64_t* slider_visible_mask;   /*        ::sliderVisibleMask */
  int32_t* slider_visibility_init;/*        ::sliderVisibilityInit */
  /* engine-side bookkeeping with no counterpart in DSPJSFX_State (null = derive): the write high-water mark of mem[]
   * (upload: null means mem_n, i.e. the whole host image counts as written) and the ZAB_FLAG_* word of the instance
   * (bit 0: sliders changed, @slider runs at the next zab_process; upload: null leaves it alone). */
  int64_t* mem_high;
  uint32_t* flags;
  /* slider masks the script raised in launches so far that zab_consume_slider_changes has not handed to the host yet (the
   * reference collects them per processBlock, consumeDspSliderChanges src/JSFXJuceProcessor.cpp:3745; a checkpoint taken
   * between a launch and the host's next look must carry them). upload: null leaves the word alone. */
  uint64_t* slider_changes;
} zab_host_state;
/* zab_state_upload replaces the instance's whole image: arena cells above the uploaded prefix that an earlier run had stored
 * to are zeroed (the host image holds nothing there), the high-water mark becomes *mem_high (or mem_n). */
int zab_state_upload(zab_engine* e, int32_t instance, const zab_host_state* h);
int zab_state_download(zab_engine* e, int32_t instance, zab_host_state* h);
enum { ZAB_SECTION_INIT = 0, ZAB_SECTION_SLIDER = 1, ZAB_SECTION_BLOCK = 2, ZAB_SECTION_SAMPLE = 3 };
/* Run one section on every instance exactly as a direct call of the generated section function would: no state
 * reset, no slider-alias sync, spl[] untouched by the wrapper. samplesblock = st->samplesblock for the call. */
int zab_run_section(zab_engine* e, int32_t section, int32_t samplesblock);

/* gmem[] segment of the engine (leaves that use gmem; reference: DspJsfxGmemAttachment, src/DspJsfxGmem.cpp).
 * One segment of 1 Mi cells per engine, shared by all its instances. read/write move raw doubles; seq returns the
 * write-sequence counter of a 1024-cell page, or the global one for page < 0. */
int zab_gmem_read(zab_engine* e, int64_t start, int64_t n, double* dst);
int zab_gmem_write(zab_engine* e, int64_t start, int64_t n, const double* src);
int zab_gmem_seq(zab_engine* e, int64_t page, uint64_t* out);

/* File slots (leaves that call file_open / file_riff / file_avail / file_mem / file_var ...; reference: the processor's
 * runtime file handles, src/JSFXJuceProcessor.cpp:4893-5215). The host decodes whatever is assigned to a slot into a flat
 * array of doubles (interleaved audio items) and hands it over; every instance of the engine sees the same slots through
 * its own handles and cursors. items == NULL unassigns the slot (file_open returns -1, as for an empty slot). */
int zab_file_slot_set(zab_engine* e, int32_t slot, int32_t channels, double sample_rate, const double* items, int64_t n_items);

/* Sample pool (leaves that call sample_read* / sample_export_mem): one immutable generation per upload, entries are
 * 1-based sample ids in array order (DspJsfxSamplePoolEntry / Generation, src/DspJsfxSamplePool.h:55-79). audio is the
 * packed float32 arena, each entry interleaved by channel starting at offset_items. Decode/resample is host work. */
typedef struct zab_pool_entry {
  int64_t offset_items;
  int32_t frames, sample_rate, channels;
  float peak, rms;
} zab_pool_entry;
int zab_pool_upload(zab_engine* e, int32_t n_entries, const zab_pool_entry* entries, const float* audio, int64_t audio_items);

/* File ingestion. The reference decodes what is assigned to a file slot / imported into the sample pool with
 * JUCE's format readers (src/JSFXJuceProcessor.cpp:15207-15305 file_* entry points over the processor's decoded slots;
 * src/DspJsfxSamplePool.cpp:473-751 worker decode), i.e. into float samples. This library reads RIFF/WAVE itself: PCM 8 (unsigned),
 * 16, 24 and 32 bit, IEEE float 32 and 64 bit, plain or WAVE_FORMAT_EXTENSIBLE, any channel count; integer samples become
 * floats as value / 2^(bits-1) (what a JUCE reader hands out). No resampling: the data keeps the file's rate, which the script
 * sees (file_riff's srate, sample_info).
 *   zab_wav_read      decodes into a malloc'ed interleaved float array the caller frees with zab_wav_free
 *   zab_file_slot_load_wav   = zab_wav_read + zab_file_slot_set (items = the floats widened to double)
 *   zab_pool_upload_wav      = one pool generation from n files, entries in argument order (1-based sample ids) */
typedef struct zab_wav_info {
  int32_t channels, sample_rate, bits, is_float;
  int64_t frames;
} zab_wav_info;
int zab_wav_read(const char* path, zab_wav_info* info, float** interleaved);
void zab_wav_free(float* interleaved);
int zab_file_slot_load_wav(zab_engine* e, int32_t slot, const char* path, zab_wav_info* info);
int zab_pool_upload_wav(zab_engine* e, int32_t n_files, const char* const* paths);

/* Device buffer helpers so hosts without a HIP binding (ctypes, cgo, JNI) can keep audio HBM-resident. */
int zab_device_alloc(zab_engine* e, int64_t bytes, void** out);
int zab_device_free(zab_engine* e, void* p);
int zab_device_upload(zab_engine* e, void* dst, const void* src, int64_t bytes);
int zab_device_download(zab_engine* e, void* dst, const void* src, int64_t bytes);
/* Fill a planar [n_instances][n_channels][frame_stride] float buffer with the benchmark's white noise
 * (xorshift64 per instance) on the device; the noise id of instance i is id_offset + i. */
int zab_device_noise(zab_engine* e, void* dst, int64_t frames, int64_t frame_stride, uint64_t id_offset);

/* Timing of the most recent zab_process: HIP events recorded on the engine's stream around the kernel launches.
 * kernel_ms = sum over launches; launches = number of kernel launches. */
int zab_last_timing(zab_engine* e, double* kernel_ms, int32_t* launches);
/* Device durations (ms) of the most recent zab_process launches, oldest first, at most 64 kept; returns the count. */
int zab_timing_history(zab_engine* e, double* kernel_ms, int32_t max_entries);
void* zab_stream(zab_engine* e);   /* hipStream_t of the engine */
int zab_used_fast_path(zab_engine* e); /* 1 if the most recent zab_process ran the leaf's hand-written kernel */
/* Name of the kernel the most recent zab_process launched (as rocprofv3 --kernel-trace lists it; templated kernels by the
 * substring before the template arguments). Valid until the next zab_process on this engine. */
const char* zab_last_kernel_name(zab_engine* e);
/* Page-locked host memory for audio buffers handed to zab_process(ZAB_BUF_HOST): with it the copies of the chunked
 * host-buffer pipeline (copy-in, kernels and copy-out of consecutive time chunks on three HIP streams) are truly
 * asynchronous. Any host memory works; pageable buffers are staged by the HIP runtime and overlap less.
 * The reference's host owns its channel buffers (juce::AudioBuffer, src/JSFXJuceProcessor.cpp:3435). */
int zab_host_alloc(size_t bytes, void** out);
int zab_host_free(void* p);
/* Launch shape of the lane-per-instance kernels as of the most recent zab_process: instances per wavefront and the
 * number of mem[] words per instance held in LDS for the length of a launch (0: the arena is read in place). */
int zab_launch_shape(zab_engine* e, int32_t* instances_per_wave, int32_t* lds_mem_words);
/* A leaf's generated time-parallel kernel hands an instance back to the serial section code when a chunk breaks one of the
 * lowering's run-time conditions (a delay line that does not advance by one cell per frame, reads that fall into a freshly written
 * span, events thicker than one frame in sixteen ...): results are the same, time is not. For the most recent zab_process call
 * (waits for it): how many launches of an instance were finished serially, and how many frames that was. Zero for leaves without
 * such a kernel. No reference counterpart: the reference has one execution path. */
int zab_handback_stats(zab_engine* e, uint64_t* instances, uint64_t* frames);

/* ---- one job over several GPUs of a node -------------------------------------------------------------------
 * The reference runs one plugin instance per audio thread; a batch job is N independent instances, so it shards by instance:
 * shard k of n owns the contiguous range [k*N/n + min(k, N%n), ...) -- sizes differ by at most one, earlier shards take the
 * extras -- on devices[k], with its own engine, stream and host thread. There is NO collective on the data path; the only
 * exchange is zab_group_reduce(), a few doubles at the end of a run, over RCCL (xGMI) when the shards sit on distinct GPUs.
 * Leaves coupled through gmem / msg_*() are "replicas only": every shard is its own communication domain.
 * cfg->n_instances is the total N, cfg->device is ignored, cfg->first_instance_id applies to global instance 0. */
typedef struct zab_group zab_group;
typedef struct zab_group_stats {
  double max_kernel_ms;    /* slowest shard's device time of its most recent zab_process (the job is as slow as that) */
  double sum_kernel_ms;
  double units;            /* samples processed by the most recent zab_group_process: sum over shards of count x channels x frames */
  double max_value;        /* max over shards of the caller's per-shard value (e.g. worst null-test residual); 0 without one */
  int32_t n_shards;
  int32_t used_rccl;       /* 1: reduced with RCCL all-reduces on the shards' streams; 0: shards share a device, reduced on the host */
} zab_group_stats;
int zab_group_create(const char* module, const zab_config* cfg, const int32_t* devices, int32_t n_devices, zab_group** out);
int zab_group_destroy(zab_group* g);
int zab_group_size(zab_group* g);                                    /* number of shards */
int zab_group_shard(zab_group* g, int32_t k, zab_engine** engine, int32_t* first, int32_t* count);
/* as zab_set_sliders, [first, first+count) in GLOBAL instance numbers (count = 0 with first = 0 broadcasts one row) */
int zab_group_set_sliders(zab_group* g, int32_t first, int32_t count, const double* values);
int zab_group_prepare(zab_group* g);
/* in[k] / out[k]: shard k's planar buffer [count_k][n_channels][frame_stride] -- device memory of devices[k] (ZAB_BUF_DEVICE:
 * every GPU is launched and the call returns; zab_group_sync waits) or host memory (ZAB_BUF_HOST: one thread per shard stages
 * its own PCIe pipeline, the call returns when all are done). */
int zab_group_process(zab_group* g, const void* const* in, void* const* out, int64_t frames, int64_t frame_stride, int32_t block,
                      int32_t placement);
int zab_group_sync(zab_group* g);
/* End-of-run statistics of the whole job; shard_values: one caller-supplied number per shard to take the maximum of, or NULL. */
int zab_group_reduce(zab_group* g, const double* shard_values, zab_group_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* ZABATCH_H */
