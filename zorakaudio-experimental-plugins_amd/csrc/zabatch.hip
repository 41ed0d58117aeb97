// zabatch.hip -- libzabatch.so: the C ABI of include/zabatch.h over HIP.
//
// Host side only orchestrates: it owns device allocations, loads a plugin module (libzab_<leaf>.so), launches the
// module's kernels on one stream and surfaces device-latched errors. All DSP -- @init/@slider/@block/@sample --
// runs on the GPU; there is no CPU execution path in this library.
//
// Reference call sites this replaces: src/JSFXJuceProcessor.cpp:3239-3342 (prepareToPlay),
// :3435-3772 (processBlock), :8958-8971 (mem calloc), :9286-9357 (slider push).
#include "../../include/zabatch.h"
#include "zab_module.h"

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>      /* types only: the library is loaded on demand (zab_group_reduce), a single-GPU host never needs it */

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(ZAB_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

std::string lib_dir() {
  Dl_info info;
  if (dladdr((void*)&lib_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
  }
  return ".";
}

// ---- white-noise generator for benchmarks / tests (host twin: zajit/noise.py) ----------
__global__ void zab_k_noise(float* dst, int n_inst, int nch, int64_t frames, int64_t stride, uint64_t id0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inst) return;
  uint64_t x = 0x9E3779B97F4A7C15ull ^ ((id0 + (uint64_t)i) * 0xD1B54A32D192ED03ull);
  if (x == 0) x = 0x9E3779B97F4A7C15ull;
  float* base = dst + (int64_t)i * nch * stride;
  for (int64_t t = 0; t < frames; ++t)
    for (int c = 0; c < nch; ++c) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
      base[(int64_t)c * stride + t] = (float)((u * 2.0 - 1.0) * 0.5);
    }
}

// ---- one instance's strided state <-> a contiguous staging buffer (zab_state_upload / zab_state_download) ----------
template <class T>
__global__ void zab_k_gather(const T* base, int64_t se, int64_t n, T* staging) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) staging[k] = base[k * se];
}
template <class T>
__global__ void zab_k_scatter(T* base, int64_t se, int64_t n, const T* staging) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) base[k * se] = staging[k];
}

template <class T>
__global__ void zab_k_fill(T* base, int64_t se, int64_t n, T v) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) base[k * se] = v;
}

}  // namespace

// mirrors ZaFileSlot / ZaFileView of csrc/zart_file.h (the runtime does not include the device headers)
struct ZabFileSlotDev { const double* items; int64_t n_items; int32_t channels; int32_t assigned; double srate; };
struct ZabFileViewDev { ZabFileSlotDev slot[16]; };
enum { kFileHandleWords = 34 };      // ZA_FH_WORDS (csrc/zart_file.h)
// mirrors ZaMsg / ZaBusView of csrc/zart_msg.h
struct ZabMsgDev { uint64_t seq, chan, src, target; double tag, a, b, c, d; uint32_t kind, pad, blen, pad2; };
struct ZabBusViewDev {
  ZabMsgDev* ring; uint64_t* global_seq; uint64_t* domain; uint64_t* ch_hash; uint32_t* ch_flags; uint64_t* ch_caps;
  uint64_t* ch_dropped; uint64_t* last_read; ZabMsgDev* outbox; uint32_t* out_count; ZabMsgDev* inbox; uint32_t* in_count;
  uint32_t n_inst, pad; uint64_t first_id;
  double *out_pay, *ring_pay, *in_pay; uint32_t *out_cells, *last_len;      // buffer messages (uses_msg == 2), else null
};
enum { kMsgRing = 4096, kMsgChannels = 24, kMsgOutbox = 1024, kMsgInbox = 1024, kMsgMaxInstances = 256, kMsgPay = 64 };
static const uint64_t kMsgDefaultDomain = 0x9ae16a3b2f90404full;

struct zab_engine {
  void* dl = nullptr;
  const ZabModule* mod = nullptr;
  zab_config cfg{};
  ZabBatch b{};
  hipStream_t stream = nullptr;
  static const int kTimingSlots = 64;
  hipEvent_t ev0[kTimingSlots] = {}, ev1[kTimingSlots] = {};
  uint64_t n_process = 0;
  bool prepared = false;
  bool sliders_dirty = false;
  bool timing_valid = false;
  bool lmem_stale = true;    // LDS window of the generic process kernel to be (re)chosen before the next launch
  int ipw0 = 64;             // instances per wavefront by policy; the LDS window may thin the waves further
  int launches = 0;
  bool used_fast = false;
  std::string kernel_name;              // of the most recent zab_process (copied: modules format theirs into a shared buffer)
  std::vector<double> last_pushed;      // [n_inst][64] slider values as the host pushed them last (reference: lastSliders)
  std::vector<uint8_t> pushed_valid;    // [n_inst]
  std::vector<void*> owned;
  unsigned long long *gmem_cells = nullptr, *gmem_page_seq = nullptr, *gmem_global_seq = nullptr;
  uint64_t gmem_cell_count = 0;
  uint32_t pool_generation = 0;
  ZabBusViewDev* d_bus = nullptr;         // message bus view (device) + host copy of the pointers
  ZabBusViewDev h_bus{};
  ZabFileViewDev* d_files = nullptr;      // file slot table (device) + its host copy
  ZabFileViewDev h_files{};
  void* state_stage = nullptr;      // device staging for zab_state_upload/download
  int64_t state_stage_bytes = 0;
  float* stage_in = nullptr;
  float* stage_out = nullptr;
  int64_t stage_bytes = 0;
  // host-buffer pipeline (zab_process, ZAB_BUF_HOST): copy-in / compute / copy-out of consecutive time chunks overlap
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t in_ready[2] = {}, k_done[2] = {}, out_done[2] = {};
  float* pipe_in[2] = {};
  float* pipe_out[2] = {};
  int64_t pipe_bytes = 0;

  template <class T>
  int alloc(T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) return fail(ZAB_E_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    e = hipMemsetAsync(q, 0, count * sizeof(T), stream);
    if (e != hipSuccess) return fail(ZAB_E_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    owned.push_back(q);
    *p = (T*)q;
    return ZAB_OK;
  }
};

// gmem segment of the engine (default size of the reference: 1 Mi cells, pages of 1024; src/DspJsfxGmem.h:17-18)
struct ZabGmemView { unsigned long long *cells, *page_seq, *page_writer, *global_seq; uint64_t cell_count, page_count; };
static int setup_gmem(zab_engine* e) {
  const uint64_t cells = 1024ull * 1024ull, pages = cells / 1024ull;
  ZabGmemView v{};
  int rc;
  if ((rc = e->alloc(&v.cells, cells)) || (rc = e->alloc(&v.page_seq, pages)) || (rc = e->alloc(&v.page_writer, pages)) ||
      (rc = e->alloc(&v.global_seq, 1)) || (rc = e->alloc(&e->b.gmem_att, (size_t)e->b.n_pad)))
    return rc;
  v.cell_count = cells; v.page_count = pages;
  ZabGmemView* dv = nullptr;
  if ((rc = e->alloc(&dv, 1))) return rc;
  if (hipMemcpyAsync(dv, &v, sizeof v, hipMemcpyHostToDevice, e->stream) != hipSuccess) return fail(ZAB_E_HIP, "gmem view upload failed");
  e->b.gmem = dv;
  e->gmem_cells = v.cells; e->gmem_page_seq = v.page_seq; e->gmem_global_seq = v.global_seq; e->gmem_cell_count = cells;
  return ZAB_OK;
}

// message bus of an engine whose leaf calls msg_*(): every instance registered in the default domain, nothing subscribed
static int reset_bus(zab_engine* e) {
  const ZabBusViewDev& v = e->h_bus;
  const size_t n = (size_t)e->b.n_pad;
  HIP_TRY(hipMemsetAsync(v.ring, 0, sizeof(ZabMsgDev) * kMsgRing, e->stream));
  HIP_TRY(hipMemsetAsync(v.global_seq, 0, 8, e->stream));
  HIP_TRY(hipMemsetAsync(v.ch_hash, 0, 8 * n * kMsgChannels, e->stream));
  HIP_TRY(hipMemsetAsync(v.ch_flags, 0, 4 * n * kMsgChannels, e->stream));
  HIP_TRY(hipMemsetAsync(v.ch_caps, 0, 8 * n * kMsgChannels, e->stream));
  HIP_TRY(hipMemsetAsync(v.ch_dropped, 0, 8 * n * kMsgChannels, e->stream));
  HIP_TRY(hipMemsetAsync(v.last_read, 0, 8 * n, e->stream));
  HIP_TRY(hipMemsetAsync(v.out_count, 0, 4 * n, e->stream));
  HIP_TRY(hipMemsetAsync(v.in_count, 0, 4 * n, e->stream));
  if (v.out_cells) {
    HIP_TRY(hipMemsetAsync(v.out_cells, 0, 4 * n, e->stream));
    HIP_TRY(hipMemsetAsync(v.last_len, 0, 4 * n, e->stream));
  }
  std::vector<uint64_t> dom(n, kMsgDefaultDomain);
  HIP_TRY(hipMemcpyAsync(v.domain, dom.data(), 8 * n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return ZAB_OK;
}
static int setup_bus(zab_engine* e) {
  if (e->b.n_inst > kMsgMaxInstances)
    return fail(ZAB_E_ARG, "%s exchanges messages: at most %d instances per engine (the reference's per-domain instance table)",
                e->mod->name, (int)kMsgMaxInstances);
  ZabBusViewDev& v = e->h_bus;
  const size_t n = (size_t)e->b.n_pad;
  int rc;
  if ((rc = e->alloc(&v.ring, kMsgRing)) || (rc = e->alloc(&v.global_seq, 1)) || (rc = e->alloc(&v.domain, n)) ||
      (rc = e->alloc(&v.ch_hash, n * kMsgChannels)) || (rc = e->alloc(&v.ch_flags, n * kMsgChannels)) ||
      (rc = e->alloc(&v.ch_caps, n * kMsgChannels)) || (rc = e->alloc(&v.ch_dropped, n * kMsgChannels)) ||
      (rc = e->alloc(&v.last_read, n)) || (rc = e->alloc(&v.outbox, n * kMsgOutbox)) || (rc = e->alloc(&v.out_count, n)) ||
      (rc = e->alloc(&v.inbox, n * kMsgInbox)) || (rc = e->alloc(&v.in_count, n)) || (rc = e->alloc(&e->d_bus, 1)))
    return rc;
  if (e->mod->uses_msg == 2 &&         // payload rows of the buffer messages (msg_send_buf / msg_recv_buf)
      ((rc = e->alloc(&v.out_pay, n * kMsgOutbox * kMsgPay)) || (rc = e->alloc(&v.ring_pay, (size_t)kMsgRing * kMsgPay)) ||
       (rc = e->alloc(&v.in_pay, n * kMsgInbox * kMsgPay)) || (rc = e->alloc(&v.out_cells, n)) || (rc = e->alloc(&v.last_len, n))))
    return rc;
  v.n_inst = (uint32_t)e->b.n_inst; v.pad = 0; v.first_id = e->b.first_id;
  if (hipMemcpyAsync(e->d_bus, &v, sizeof v, hipMemcpyHostToDevice, e->stream) != hipSuccess) return fail(ZAB_E_HIP, "bus view upload failed");
  e->b.bus = e->d_bus;
  return reset_bus(e);
}

static hipError_t create_events(zab_engine* e) {
  for (int i = 0; i < zab_engine::kTimingSlots; ++i) {
    hipError_t he = hipEventCreate(&e->ev0[i]);
    if (he != hipSuccess) return he;
    he = hipEventCreate(&e->ev1[i]);
    if (he != hipSuccess) return he;
  }
  return hipSuccess;
}

extern "C" {

const char* zab_last_error(void) { return g_err.c_str(); }
int zab_abi_version(void) { return ZAB_MODULE_ABI; }
int zab_host_abi_version(void) { return ZAB_HOST_ABI; }

// zab_host_state grows at its end: a field is there only if the caller's struct reaches it
#define ZAB_HS_HAS(h, field) ((h)->struct_size >= offsetof(zab_host_state, field) + sizeof((h)->field))
static bool host_state_ok(const zab_host_state* h) { return h && ZAB_HS_HAS(h, flags) && h->struct_size <= 4096; }

int zab_create(const char* module, const zab_config* cfg, zab_engine** out) {
  if (!module || !cfg || !out) return fail(ZAB_E_ARG, "zab_create: null argument");
  if (cfg->n_instances <= 0) return fail(ZAB_E_ARG, "zab_create: n_instances must be > 0");
  if (!(cfg->srate > 0.0)) return fail(ZAB_E_ARG, "zab_create: srate must be > 0");
  *out = nullptr;
  int ndev = 0;
  hipError_t de = hipGetDeviceCount(&ndev);
  if (de != hipSuccess || ndev <= 0)
    return fail(ZAB_E_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                de == hipSuccess ? "device count 0" : hipGetErrorString(de));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(ZAB_E_ARG, "device %d out of range (%d devices)", cfg->device, ndev);

  std::string path = module;
  if (path.find('/') == std::string::npos && path.find(".so") == std::string::npos)
    path = lib_dir() + "/libzab_" + path + ".so";
  void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!dl) return fail(ZAB_E_MODULE, "cannot load plugin module %s: %s", path.c_str(), dlerror());
  typedef const ZabModule* (*getter)(void);
  getter g = (getter)dlsym(dl, "zab_module_get");
  // (a loaded module is never dlclose'd: its code object is registered with the HIP runtime, and unloading it under the
  // runtime's feet has crashed the process)
  if (!g) return fail(ZAB_E_MODULE, "%s exports no zab_module_get", path.c_str());
  const ZabModule* m = g();
  if (!m || m->abi != ZAB_MODULE_ABI) {
    return fail(ZAB_E_MODULE, "%s: module ABI %d, runtime ABI %d", path.c_str(), m ? m->abi : -1, ZAB_MODULE_ABI);
  }

  zab_engine* e = new zab_engine();
  e->dl = dl; e->mod = m; e->cfg = *cfg;
  if (e->cfg.max_block <= 0) e->cfg.max_block = 512;
  int rc = ZAB_OK;
  hipError_t he;
  // (a message-bus leaf runs one small launch per host block plus a publish kernel: thousands of launches per second of audio,
  //  each waiting for its turn when other engines keep long kernels in flight on the same device -- a 10 s mixed-catalog run
  //  was measured spending 94 s in the 1 876 launches of one such engine. Its stream takes the device's highest priority.)
  auto make_stream = [&]() -> hipError_t {
    int least = 0, greatest = 0;
    if (m->uses_msg && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
      return hipStreamCreateWithPriority(&e->stream, hipStreamDefault, greatest);
    return hipStreamCreate(&e->stream);
  };
  if ((he = hipSetDevice(cfg->device)) != hipSuccess || (he = make_stream()) != hipSuccess ||
      (he = create_events(e)) != hipSuccess) {
    rc = fail(ZAB_E_HIP, "HIP stream/event setup failed: %s", hipGetErrorString(he));
    zab_destroy(e);
    return rc;
  }
  ZabBatch& b = e->b;
  b.n_inst = cfg->n_instances;
  b.n_pad = (cfg->n_instances + 63) / 64 * 64;
  b.nvars = m->nvars;
  b.instance_major = m->prefer_instance_major ? 1 : 0;
  {
    // lanes per wavefront given to instances by the lane-per-instance kernels. A wavefront's run time does not depend on
    // how many of its lanes are live, so a batch is spread over more wavefronts (shorter audio-tile staging per wave)
    // until there are two per SIMD (256 CUs x 4 SIMDs); 16 is the floor: 16 lanes x 8 B = one 128-byte line per access
    // to the interleaved state arrays.
    // Measured on the catalog at 1024 instances (profiles/README.md): thinning pays for leaves with many channels (tile
    // staging is NCH x ipw loads per frame: NeuroCV 18 ch 200 -> 87 ms, RED 6 ch 74 -> 39 ms) and is neutral to slightly
    // negative for 2..4-channel leaves, so it is applied from 6 channels up.
    // Leaves with FFT builtins on the audio path (prefer_instance_major == 2) run 8 instances per wavefront, each on 8
    // replica lanes (zab_generic.hip.h): the cooperative transforms (zart_fft.h) then queue 8 requests per wave instead
    // of 64 and the chip runs 8x the wavefronts. Measured at 1024-2048 instances (ipw 64 / 16 / 8 / 4 / 1): FFT round trip
    // 11.4 / 2.9 / 1.45 / 1.45 / 1.46 ms, STFT fixture 443 / 276 / 248 / 232 / 780 ms, DOT 642 / 600 / 593 / 593 / 2200 ms --
    // below ~4 the serial parts pay for 64x the memory instructions chip-wide.
    int ipw = m->prefer_instance_major >= 2 ? 8 : (m->nch >= 6 ? 16 : 64);
    // Replica-lane leaves (FFT builtins or accumulation loops on the audio path) want a fixed number of wavefronts rather than
    // a fixed width: ~256 when each wavefront holds a 64 KB LDS transform buffer (two per CU), ~1024 otherwise. Measured:
    // 4096-pt round trips, 256 buffers: 5.6 / 2.8 / 1.4 ms at ipw 8 / 4 / 2, 2048 buffers: 1.44 ms at any; DOT x4096: 133 / 90 /
    // 75 ms at ipw 4 / 8 / 16; TSEQ x1024: 417 / 274 / 294 / 486 ms at ipw 1 / 2 / 4 / 8, x4096: 329 / 227 / 328 at 2 / 4 / 8,
    // x16384: 457 / 385 / 340 / 480 at 4 / 8 / 16 / 32.
    if (m->prefer_instance_major >= 2) {
      // An FFT leaf built with the 1024-point LDS buffer (zart_fft.h, the default since round 2) holds 24 KB per wavefront, six
      // fit a CU, and it gets the 1024 target too. Measured (2048 buffers, fft + permute + ipermute + ifft round trip, target
      // 256 / 1024 / 2048 / 4096 wavefronts): 1024 points 349 / 126 / 126 / 127 us, 4096 points (sliced) 2564 / 991 / 1001 / 992 us;
      // DOT x1024 161 / 144 / 144 / 145 ms; the STFT fixtures and PsychoConvolver do not move (their serial script loops
      // dominate). ZAB_FFT_WAVES overrides the target (experiments).
      int lds_points = 4096;
      if (auto fp = (int (*)(void))dlsym(dl, "zab_module_fft_lds_points")) lds_points = fp();
      const char* tw = getenv("ZAB_FFT_WAVES");
      const int64_t target = m->fft_scratch_doubles > 0 ? (lds_points <= 1024 ? (tw ? atoi(tw) : 1024) : 256) : 1024;
      // From one instance per wavefront (all 64 lanes on its cooperative transforms, loops and copies) since the audio tile of
      // these leaves is sized at launch (zab_generic.hip.h: 25 KB of LDS per wavefront, six per CU = 1536 resident) and a loop
      // needs 64 trips, not two per lane, to be shared (zart.h za_coop_min). Measured at 1024 instances, 1 vs 2 per wavefront:
      // STFT fixtures 18.3 / 24.1 ms per 16 384 frames, DOT 134 / 159 and TSEQ 1838 / 2188 ms per 48 000, PsychoConvolver with
      // an impulse response x256 45 / 59 ms; Contour x384 88 / 82, SpectralStabilizer and Texture unchanged; the FFT round trip of
      // 2048 buffers (beyond the target either way) 1056 / 977 us. Before those two changes one per wavefront lost everywhere
      // (17 KB tile: three waves per CU; DOT's 100-tap FIR fell back to 64 copies of the serial loop: 749 ms).
      // (3: FFT builtins on the audio path of a leaf WITHOUT replica lanes -- gmem users, i.e. Sample: two per wavefront as
      //  before; one lone lane per wavefront took it from 3.8 to 7.9 s.)
      ipw = m->prefer_instance_major == 2 ? 1 : 2;
      while (ipw < 64 && (int64_t)cfg->n_instances > target * ipw) ipw <<= 1;
    }
    while (ipw < 64 && (int64_t)cfg->n_instances > 2048ll * ipw) ipw <<= 1;
    if (const char* f = getenv("ZAB_IPW")) { const int v = atoi(f); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) ipw = v; }
    b.ipw = ipw;
    e->ipw0 = ipw;
  }
  if (const char* f = getenv("ZAB_FORCE_LAYOUT")) {     // experiments only: "im" / "il" overrides the module's preference
    if (!strcmp(f, "im")) b.instance_major = 1;
    if (!strcmp(f, "il") && m->prefer_instance_major < 2) b.instance_major = 0;     // (>= 2: compiled for contiguous arenas, ZA_MEM_STRIDE1)
  }
  b.mem_cap = cfg->mem_cap > 0 ? cfg->mem_cap : m->default_mem_cap;
  if (b.mem_cap > 2147483520ll - 1024) {   // device addresses are converted with one 32-bit instruction (zart.h za_addr1)
    rc = fail(ZAB_E_ARG, "zab_create: mem_cap %lld: at most 2^31 - 1152 doubles per instance", (long long)b.mem_cap);
    zab_destroy(e);
    return rc;
  }
  b.srate = cfg->srate;
  b.first_id = cfg->first_instance_id ? cfg->first_instance_id : 1;
  {
    // epochs are unique across engines of the process (device pointers get recycled, so modules that cache plans
    // per batch must not be able to confuse a new engine with a destroyed one)
    static std::atomic<uint64_t> next_engine{1};
    b.epoch = next_engine.fetch_add(1) << 32;
  }
  const int64_t P = b.n_pad;
  if (b.instance_major) {
    b.var_se = 1; b.var_si = m->nvars;
    b.sl_se = 1; b.sl_si = 64;
    b.mem_se = 1; b.mem_si = b.mem_cap;
    b.mt_se = 1; b.mt_si = 624;
    b.fft_se = 1; b.fft_si = m->fft_scratch_doubles;
    b.fh_se = 1; b.fh_si = kFileHandleWords;
  } else {
    b.var_se = P; b.var_si = 1;
    b.sl_se = P; b.sl_si = 1;
    b.mem_se = P; b.mem_si = 1;
    b.mt_se = P; b.mt_si = 1;
    b.fft_se = P; b.fft_si = 1;
    b.fh_se = P; b.fh_si = 1;
  }
  b.fft_cap = m->fft_scratch_doubles;
  if ((rc = e->alloc(&b.vars, (size_t)P * m->nvars)) || (rc = e->alloc(&b.sliders, (size_t)P * 64)) ||
      (rc = e->alloc(&b.spl, (size_t)P * 64)) || (rc = e->alloc(&b.mem, (size_t)P * b.mem_cap)) ||
      (rc = e->alloc(&b.mt, (size_t)P * 624)) || (rc = e->alloc(&b.mti, (size_t)P)) ||
      (rc = e->alloc(&b.mem_high, (size_t)P)) || (rc = e->alloc(&b.mem_need, (size_t)P)) ||
      (rc = e->alloc(&b.err, (size_t)P)) || (rc = e->alloc(&b.flags, (size_t)P)) ||
      (rc = e->alloc(&b.pend, (size_t)P * 4)) ||      /* change / automate / automate-end, + their OR since the host last looked */ (rc = e->alloc(&b.vis_mask, (size_t)P)) ||
      (rc = e->alloc(&b.vis_init, (size_t)P)) || (rc = e->alloc(&b.resume, (size_t)P + 8)) ||     /* + the two hand-back counters, zab_handback_stats */
      (m->fft_scratch_doubles > 0 && (rc = e->alloc(&b.fft, (size_t)P * m->fft_scratch_doubles))) ||
      (m->uses_gmem && (rc = setup_gmem(e))) ||
      (m->uses_files && ((rc = e->alloc(&e->d_files, 1)) || (rc = e->alloc(&b.fh, (size_t)P * kFileHandleWords)))) ||
      (m->uses_msg && (rc = setup_bus(e)))) {
    zab_destroy(e);
    return rc;
  }
  b.files = e->d_files;                    // zero-filled: every slot unassigned until zab_file_slot_set
  if ((he = hipStreamSynchronize(e->stream)) != hipSuccess) {
    rc = fail(ZAB_E_HIP, "state clear failed: %s", hipGetErrorString(he));
    zab_destroy(e);
    return rc;
  }
  *out = e;
  return ZAB_OK;
}

int zab_destroy(zab_engine* e) {
  if (!e) return ZAB_OK;
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->s_in) hipStreamSynchronize(e->s_in);          // (a failed pipelined call may have left copies in flight)
  if (e->s_out) hipStreamSynchronize(e->s_out);
  for (void* p : e->owned) hipFree(p);
  if (e->stage_in) hipFree(e->stage_in);
  if (e->stage_out) hipFree(e->stage_out);
  for (int k = 0; k < 2; ++k) {
    if (e->pipe_in[k]) hipFree(e->pipe_in[k]);
    if (e->pipe_out[k]) hipFree(e->pipe_out[k]);
    if (e->in_ready[k]) hipEventDestroy(e->in_ready[k]);
    if (e->k_done[k]) hipEventDestroy(e->k_done[k]);
    if (e->out_done[k]) hipEventDestroy(e->out_done[k]);
  }
  if (e->s_in) hipStreamDestroy(e->s_in);
  if (e->s_out) hipStreamDestroy(e->s_out);
  if (e->state_stage) hipFree(e->state_stage);
  for (int i = 0; i < zab_engine::kTimingSlots; ++i) {
    if (e->ev0[i]) hipEventDestroy(e->ev0[i]);
    if (e->ev1[i]) hipEventDestroy(e->ev1[i]);
  }
  if (e->stream) hipStreamDestroy(e->stream);
  // the module stays loaded: its code object is registered with the HIP runtime for the process lifetime
  delete e;
  return ZAB_OK;
}

int zab_get_info(zab_engine* e, zab_info* o) {
  if (!e || !o) return fail(ZAB_E_ARG, "zab_get_info: null argument");
  memset(o, 0, sizeof *o);
  snprintf(o->name, sizeof o->name, "%s", e->mod->name);
  o->nvars = e->mod->nvars; o->n_channels = e->mod->nch; o->n_inputs = e->mod->n_in; o->n_outputs = e->mod->n_out;
  o->has_init = e->mod->has_init; o->has_slider = e->mod->has_slider; o->has_block = e->mod->has_block;
  o->has_sample = e->mod->has_sample; o->has_fast_path = e->mod->launch_fast != nullptr;
  o->mem_cap = e->b.mem_cap; o->n_instances = e->b.n_inst; o->layout_instance_major = e->b.instance_major;
  return ZAB_OK;
}

int zab_var_count(zab_engine* e) { return e ? e->mod->nvars : 0; }
const char* zab_var_name(zab_engine* e, int i) {
  if (!e || i < 0 || i >= e->mod->nvars || !e->mod->var_names) return "";
  return e->mod->var_names[i];
}
int zab_var_index(zab_engine* e, const char* name) {
  if (!e || !name || !e->mod->var_names) return -1;
  for (int i = 0; i < e->mod->nvars; ++i) if (!strcmp(e->mod->var_names[i], name)) return i;
  return -1;
}

static int range_ok(zab_engine* e, int32_t first, int32_t count) {
  return first >= 0 && count >= 0 && (int64_t)first + count <= e->b.n_inst;
}

int zab_set_sliders(zab_engine* e, int32_t first, int32_t count, const double* values) {
  if (!e || !values) return fail(ZAB_E_ARG, "zab_set_sliders: null argument");
  const bool bcast = (count == 0 && first == 0);
  if (!bcast && !range_ok(e, first, count)) return fail(ZAB_E_ARG, "zab_set_sliders: range [%d,+%d) outside batch", first, count);
  const ZabBatch& b = e->b;
  const int lo = bcast ? 0 : first, n = bcast ? b.n_inst : count;
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->last_pushed.empty()) { e->last_pushed.assign((size_t)b.n_inst * 64, 0.0); e->pushed_valid.assign((size_t)b.n_inst, 0); }
  // pushParamsToStateSliders (:9286-9357): the host's values always land in st.sliders[]; "changed" compares them with what
  // the host pushed last time, not with what a script may have written there since
  std::vector<double> stage((size_t)n * 64);
  std::vector<uint32_t> flags((size_t)n);
  HIP_TRY(hipMemcpy(flags.data(), b.flags + lo, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    const double* src = values + (bcast ? 0 : (size_t)i * 64);
    double* last = &e->last_pushed[(size_t)(lo + i) * 64];
    if (!e->pushed_valid[(size_t)(lo + i)] || memcmp(last, src, 64 * sizeof(double))) {
      memcpy(last, src, 64 * sizeof(double));
      e->pushed_valid[(size_t)(lo + i)] = 1;
      flags[(size_t)i] |= ZAB_FLAG_SLIDER_DIRTY;
    }
    if (b.instance_major) memcpy(&stage[(size_t)i * 64], src, 64 * sizeof(double));
    else for (int k = 0; k < 64; ++k) stage[(size_t)k * n + i] = src[k];
  }
  if (b.instance_major)
    HIP_TRY(hipMemcpy(b.sliders + (size_t)lo * 64, stage.data(), sizeof(double) * 64 * n, hipMemcpyHostToDevice));
  else
    HIP_TRY(hipMemcpy2D(b.sliders + lo, sizeof(double) * b.n_pad, stage.data(), sizeof(double) * n, sizeof(double) * n, 64, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.flags + lo, flags.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
  for (uint32_t f : flags) if (f & ZAB_FLAG_SLIDER_DIRTY) e->sliders_dirty = true;
  e->b.epoch++;
  return ZAB_OK;
}

static int read_strided(zab_engine* e, const double* base, int64_t se, int64_t si, int32_t first, int32_t count,
                        int64_t e0, int64_t ne, double* dst) {
  // dst[i][k] = base[(e0+k)*se + (first+i)*si]
  if (ne <= 0 || count <= 0) return ZAB_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (se == 1) {
    HIP_TRY(hipMemcpy2D(dst, sizeof(double) * ne, base + (int64_t)first * si + e0, sizeof(double) * si, sizeof(double) * ne,
                        count, hipMemcpyDeviceToHost));
    return ZAB_OK;
  }
  std::vector<double> tmp((size_t)ne * count);
  HIP_TRY(hipMemcpy2D(tmp.data(), sizeof(double) * count, base + e0 * se + first, sizeof(double) * se, sizeof(double) * count,
                      ne, hipMemcpyDeviceToHost));
  for (int64_t k = 0; k < ne; ++k)
    for (int i = 0; i < count; ++i) dst[(size_t)i * ne + k] = tmp[(size_t)k * count + i];
  return ZAB_OK;
}

int zab_get_sliders(zab_engine* e, int32_t first, int32_t count, double* values) {
  if (!e || !values || !range_ok(e, first, count)) return fail(ZAB_E_ARG, "zab_get_sliders: bad argument");
  return read_strided(e, e->b.sliders, e->b.sl_se, e->b.sl_si, first, count, 0, 64, values);
}

int zab_consume_slider_changes(zab_engine* e, int32_t first, int32_t count, uint64_t* masks, double* values) {
  if (!e || !range_ok(e, first, count)) return fail(ZAB_E_ARG, "zab_consume_slider_changes: bad argument");
  if (count == 0) return ZAB_OK;
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const ZabBatch& b = e->b;
  std::vector<uint64_t> m((size_t)count);
  uint64_t* sticky = b.pend + 3 * (int64_t)b.n_pad + first;
  HIP_TRY(hipMemcpy(m.data(), sticky, sizeof(uint64_t) * count, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(sticky, 0, sizeof(uint64_t) * count));
  bool any = false;
  for (uint64_t v : m) any = any || v != 0;
  std::vector<double> rows;
  double* dst = values;
  if (!dst && any) { rows.resize((size_t)count * 64); dst = rows.data(); }
  if (dst) { const int rc = read_strided(e, b.sliders, b.sl_se, b.sl_si, first, count, 0, 64, dst); if (rc) return rc; }
  if (any && !e->last_pushed.empty())
    for (int i = 0; i < count; ++i)
      for (int k = 0; k < 64; ++k)
        if (m[(size_t)i] >> k & 1ull) e->last_pushed[(size_t)(first + i) * 64 + k] = dst[(size_t)i * 64 + k];
  if (masks) memcpy(masks, m.data(), sizeof(uint64_t) * count);
  return ZAB_OK;
}

static int check_device_errors(zab_engine* e, const char* where) {
  const ZabBatch& b = e->b;
  std::vector<uint32_t> err((size_t)b.n_inst);
  HIP_TRY(hipMemcpy(err.data(), b.err, sizeof(uint32_t) * b.n_inst, hipMemcpyDeviceToHost));
  uint32_t any = 0; int who = -1;
  for (int i = 0; i < b.n_inst; ++i) if (err[i] & 7u) { any |= err[i]; if (who < 0) who = i; }
  if (!any) return ZAB_OK;
  HIP_TRY(hipMemset(b.err, 0, sizeof(uint32_t) * b.n_pad));
  if (any & 1u) {
    std::vector<int64_t> need((size_t)b.n_inst);
    HIP_TRY(hipMemcpy(need.data(), b.mem_need, sizeof(int64_t) * b.n_inst, hipMemcpyDeviceToHost));
    int64_t mx = 0;
    for (int64_t v : need) mx = v > mx ? v : mx;
    return fail(ZAB_E_MEM_OVERFLOW, "%s: %s stored to mem[] past the arena (instance %d first): mem_cap=%lld, needed >= %lld; "
                "state is invalid, recreate with a larger zab_config.mem_cap", where, e->mod->name, who, (long long)b.mem_cap, (long long)mx);
  }
  if (any & 4u) return fail(ZAB_E_UNSUPPORTED, "%s: %s reached a host-only builtin (MIDI/file/msg) on the device (instance %d)", where, e->mod->name, who);
  return fail(ZAB_E_LOOP_CAP, "%s: %s exceeded the loop safety cap (instance %d)", where, e->mod->name, who);
}

// LDS window of the lane-per-instance process kernel (zart.h "LDS WINDOW"): when the arena footprint of every instance
// fits, mem[0, K) lives in LDS for the length of a launch. K * ipw doubles per single-wave workgroup, next to the audio
// tile; thinner waves (smaller ipw) buy a larger window per instance. Taken only when every wave of the batch is resident
// at once (256 CUs x what 160 KB of LDS hold) -- a window that serialises the batch into rounds loses what it gains.
static int choose_lmem(zab_engine* e) {
  e->lmem_stale = false;
  e->b.lmem_words = 0;
  e->b.ipw = e->ipw0;
  const char* off = getenv("ZAB_LMEM");
  if (!e->mod->lmem_ok || (off && !strcmp(off, "0")) || e->b.n_inst <= 0) return ZAB_OK;
  std::vector<int64_t> hi((size_t)e->b.n_inst);
  HIP_TRY(hipMemcpyAsync(hi.data(), e->b.mem_high, hi.size() * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  int64_t top = 0;
  for (int64_t h : hi) top = h > top ? h : top;
  const int64_t K = (top + 7) & ~(int64_t)7;
  if (top <= 0 || K > e->b.mem_cap) return ZAB_OK;
  const int nch = e->mod->nch, tt = nch <= 4 ? 32 : nch <= 8 ? 16 : nch <= 16 ? 8 : nch <= 32 ? 4 : 2;    // ZA_TT
  const int64_t cu_lds = 160 * 1024, wg_max = 128 * 1024;
  const bool replicas = e->mod->prefer_instance_major == 2;       // (their tile has ipw rows, not 64: zab_generic.hip.h)
  for (int ipw = e->ipw0; ipw >= 1; ipw >>= 1) {
    const int64_t tile = (int64_t)nch * (replicas ? ipw : 64) * (tt + 1) * 4;
    const int64_t lds = K * ipw * 8 + tile;
    if (lds > wg_max) continue;
    const int64_t waves = (e->b.n_inst + ipw - 1) / ipw;
    if (waves > 256 * (cu_lds / lds)) break;             // thinner waves only add more of them
    e->b.ipw = ipw;
    e->b.lmem_words = (int32_t)K;
    break;
  }
  return ZAB_OK;
}

int zab_prepare(zab_engine* e) {
  if (!e) return fail(ZAB_E_ARG, "zab_prepare: null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->mod->uses_msg) { int rb = reset_bus(e); if (rb) return rb; }     // DspJsfxRuntime::reset + fresh registrations
  hipError_t he = e->mod->launch_prepare(&e->b, e->stream);
  if (he != hipSuccess) return fail(ZAB_E_HIP, "prepare launch failed: %s", hipGetErrorString(he));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->prepared = true;
  e->sliders_dirty = false;
  e->lmem_stale = true;
  e->b.epoch++;
  return check_device_errors(e, "zab_prepare");
}

// One span of frames through the leaf's kernels on the engine's stream: the message leaves host block by host block with a
// bus flush in between, everything else in one launch (hand-written kernel when it applies).
static hipError_t launch_span(zab_engine* e, const ZabAudio& a, bool fast) {
  hipError_t he = hipSuccess;
  if (e->mod->uses_msg && e->mod->launch_msg_flush) {
    // block k's messages are block k + 1's inbox (csrc/zart_msg.h)
    for (int64_t pos = 0; pos < a.frames && he == hipSuccess; pos += a.block) {
      ZabAudio ab = a;
      ab.in = a.in ? a.in + pos : nullptr;
      ab.out = a.out ? a.out + pos : nullptr;
      ab.frames = (a.frames - pos < a.block) ? (a.frames - pos) : a.block;
      he = fast ? e->mod->launch_fast(&e->b, &ab, e->stream) : e->mod->launch_process(&e->b, &ab, e->stream);
      if (he == hipSuccess) he = e->mod->launch_msg_flush(&e->b, e->stream);
      e->launches += 2;
    }
    return he;
  }
  e->launches += 1;
  if (fast) return e->mod->launch_fast(&e->b, &a, e->stream);
  he = e->mod->launch_process(&e->b, &a, e->stream);
  if (he != hipSuccess && e->b.lmem_words) {          // the device refused that much LDS: run from the arena instead
    (void)hipGetLastError();
    e->b.lmem_words = 0; e->b.ipw = e->ipw0;
    he = e->mod->launch_process(&e->b, &a, e->stream);
  }
  return he;
}

// Host buffers, chunked along time: H2D of chunk k + 1 (stream s_in), kernels of chunk k (engine stream) and D2H of chunk
// k - 1 (stream s_out) run together; two device buffers per direction. Instance state simply carries over from launch to
// launch, as between any two zab_process calls. Pinned host memory (zab_host_alloc) makes the copies truly asynchronous.
static int process_host_pipelined(zab_engine* e, const float* in, float* out, const ZabAudio& a, bool fast, int64_t chunk) {
  const int64_t rows = (int64_t)e->b.n_inst * e->mod->nch;
  const int64_t need = rows * chunk * (int64_t)sizeof(float);
  if (!e->s_in) {
    HIP_TRY(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(hipEventCreateWithFlags(&e->in_ready[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->k_done[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->out_done[k], hipEventDisableTiming));
    }
  }
  if (need > e->pipe_bytes) {
    for (int k = 0; k < 2; ++k) {
      if (e->pipe_in[k]) hipFree(e->pipe_in[k]);
      if (e->pipe_out[k]) hipFree(e->pipe_out[k]);
      e->pipe_in[k] = e->pipe_out[k] = nullptr;
    }
    e->pipe_bytes = 0;
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(hipMalloc((void**)&e->pipe_in[k], need));
      HIP_TRY(hipMalloc((void**)&e->pipe_out[k], need));
    }
    e->pipe_bytes = need;
  }
  // whatever the engine stream has queued so far (slider prologue, earlier calls) precedes the first chunk
  int64_t k = 0;
  for (int64_t pos = 0; pos < a.frames; pos += chunk, ++k) {
    const int buf = (int)(k & 1);
    const int64_t n = a.frames - pos < chunk ? a.frames - pos : chunk;
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(e->s_in, e->k_done[buf], 0));            // kernels of chunk k - 2 have read this buffer
    HIP_TRY(hipMemcpy2DAsync(e->pipe_in[buf], (size_t)chunk * 4, in + pos, (size_t)a.frame_stride * 4, (size_t)n * 4, (size_t)rows,
                             hipMemcpyHostToDevice, e->s_in));
    HIP_TRY(hipEventRecord(e->in_ready[buf], e->s_in));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->in_ready[buf], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(e->stream, e->out_done[buf], 0));        // chunk k - 2's output has left this buffer
    ZabAudio ac = a;
    ac.in = e->pipe_in[buf]; ac.out = e->pipe_out[buf]; ac.frames = n; ac.frame_stride = chunk;
    const hipError_t he = launch_span(e, ac, fast);
    if (he != hipSuccess) return fail(ZAB_E_HIP, "process launch failed: %s", hipGetErrorString(he));
    HIP_TRY(hipEventRecord(e->k_done[buf], e->stream));
    HIP_TRY(hipStreamWaitEvent(e->s_out, e->k_done[buf], 0));
    HIP_TRY(hipMemcpy2DAsync(out + pos, (size_t)a.frame_stride * 4, e->pipe_out[buf], (size_t)chunk * 4, (size_t)n * 4, (size_t)rows,
                             hipMemcpyDeviceToHost, e->s_out));
    HIP_TRY(hipEventRecord(e->out_done[buf], e->s_out));
  }
  return ZAB_OK;
}

int zab_process(zab_engine* e, const void* in, void* out, int64_t frames, int64_t frame_stride, int32_t block, int32_t placement) {
  if (!e) return fail(ZAB_E_ARG, "zab_process: null engine");
  if (!e->prepared) return fail(ZAB_E_STATE, "zab_process before zab_prepare");
  const int nch = e->mod->nch;
  if (frames < 0 || block <= 0 || frame_stride < frames) return fail(ZAB_E_ARG, "zab_process: bad frames/block/stride");
  if (block > e->cfg.max_block) return fail(ZAB_E_ARG, "zab_process: block %d > max_block %d", block, e->cfg.max_block);
  if (nch > 0 && e->mod->has_sample && (!in || !out)) return fail(ZAB_E_ARG, "zab_process: null audio buffer");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->b.resume) HIP_TRY(hipMemsetAsync(e->b.resume + e->b.n_pad, 0, 2 * sizeof(int64_t), e->stream));     // hand-back counters of this call
  ZabAudio a{};
  a.frames = frames; a.frame_stride = frame_stride; a.block = block;
  const int64_t bytes = (int64_t)e->b.n_inst * nch * frame_stride * (int64_t)sizeof(float);
  // Host buffers longer than two chunks go through the three-stream pipeline (chunks are whole host blocks, so the
  // script sees the same block sequence); short ones are staged whole.
  const int64_t rows = (int64_t)e->b.n_inst * nch;
  int64_t chunk = 0;
  bool pipelined = false;
  if (placement == ZAB_BUF_HOST && rows > 0 && e->mod->has_sample) {
    const char* env = getenv("ZAB_PIPE_CHUNK_KB");                             // 0 disables the pipeline
    const int64_t target = (env ? atoll(env) : 32768) << 10;                   // bytes per chunk and direction
    chunk = target > 0 ? (target / (rows * 4) / block) * block : 0;
    if (chunk < block) chunk = block;
    pipelined = target > 0 && frames >= 3 * chunk;
  }
  if (placement == ZAB_BUF_HOST && pipelined) {
    a.in = nullptr; a.out = nullptr;          // per chunk, see process_host_pipelined
  } else if (placement == ZAB_BUF_HOST) {
    if (bytes > e->stage_bytes) {
      if (e->stage_in) hipFree(e->stage_in);
      if (e->stage_out) hipFree(e->stage_out);
      e->stage_in = e->stage_out = nullptr; e->stage_bytes = 0;
      HIP_TRY(hipMalloc((void**)&e->stage_in, bytes));
      HIP_TRY(hipMalloc((void**)&e->stage_out, bytes));
      e->stage_bytes = bytes;
    }
    if (bytes) HIP_TRY(hipMemcpyAsync(e->stage_in, in, bytes, hipMemcpyHostToDevice, e->stream));
    a.in = e->stage_in; a.out = e->stage_out;
  } else if (placement == ZAB_BUF_DEVICE) {
    a.in = (const float*)in; a.out = (float*)out;
  } else {
    return fail(ZAB_E_ARG, "zab_process: placement %d", placement);
  }
  if (e->sliders_dirty && e->cfg.path != ZAB_PATH_GENERIC && e->mod->launch_fast) {
    // hand-written kernels expect @slider to have run already; the generic process kernel does it itself
    hipError_t hs = e->mod->launch_slider(&e->b, e->stream);
    if (hs != hipSuccess) return fail(ZAB_E_HIP, "slider launch failed: %s", hipGetErrorString(hs));
  }
  e->sliders_dirty = false;
  bool fast = false;
  if (e->cfg.path != ZAB_PATH_GENERIC && e->mod->launch_fast && e->mod->fast_applies) fast = e->mod->fast_applies(&e->b, &a) != 0;
  if (e->cfg.path == ZAB_PATH_FAST && !fast)
    return fail(ZAB_E_ARG, "ZAB_PATH_FAST requested but %s's hand-written kernel does not apply to this configuration", e->mod->name);
  if (!fast && e->lmem_stale) { const int rc = choose_lmem(e); if (rc) return rc; }
  const int slot = (int)(e->n_process % zab_engine::kTimingSlots);
  HIP_TRY(hipEventRecord(e->ev0[slot], e->stream));
  e->launches = 0;
  hipError_t he = hipSuccess;
  if (pipelined) {
    const int rc = process_host_pipelined(e, (const float*)in, (float*)out, a, fast, chunk);
    if (rc) {                                            // drain what was queued: the caller owns the host buffers again
      if (e->s_in) hipStreamSynchronize(e->s_in);
      hipStreamSynchronize(e->stream);
      if (e->s_out) hipStreamSynchronize(e->s_out);
      return rc;
    }
  } else {
    he = launch_span(e, a, fast);
  }
  if (he != hipSuccess) return fail(ZAB_E_HIP, "process launch failed: %s", hipGetErrorString(he));
  HIP_TRY(hipEventRecord(e->ev1[slot], e->stream));
  e->n_process++;
  e->timing_valid = true; e->used_fast = fast;
  { const char* kn = fast ? e->mod->fast_kernel_name : e->mod->generic_kernel_name; e->kernel_name = kn ? kn : ""; }
  if (placement == ZAB_BUF_HOST) {
    if (pipelined) {
      HIP_TRY(hipStreamSynchronize(e->s_out));
      HIP_TRY(hipStreamSynchronize(e->s_in));
    } else if (bytes) {
      HIP_TRY(hipMemcpyAsync(out, e->stage_out, bytes, hipMemcpyDeviceToHost, e->stream));
    }
    return zab_sync(e);
  }
  return ZAB_OK;
}

int zab_sync(zab_engine* e) {
  if (!e) return fail(ZAB_E_ARG, "zab_sync: null engine");
  HIP_TRY(hipStreamSynchronize(e->stream));
  return check_device_errors(e, "zab_process");
}

int zab_read_vars(zab_engine* e, int32_t first, int32_t count, double* dst) {
  if (!e || !dst || !range_ok(e, first, count)) return fail(ZAB_E_ARG, "zab_read_vars: bad argument");
  return read_strided(e, e->b.vars, e->b.var_se, e->b.var_si, first, count, 0, e->mod->nvars, dst);
}

int zab_read_mem(zab_engine* e, int32_t first, int32_t count, int64_t start, int64_t n, double* dst) {
  if (!e || !dst || !range_ok(e, first, count) || start < 0 || n < 0) return fail(ZAB_E_ARG, "zab_read_mem: bad argument");
  const int64_t avail = start < e->b.mem_cap ? (e->b.mem_cap - start < n ? e->b.mem_cap - start : n) : 0;
  if (avail < n) memset(dst, 0, sizeof(double) * (size_t)count * n);   // beyond the arena reads 0, like ungrown mem
  if (avail == n) return read_strided(e, e->b.mem, e->b.mem_se, e->b.mem_si, first, count, start, n, dst);
  if (avail > 0) {
    std::vector<double> tmp((size_t)count * avail);
    int rc = read_strided(e, e->b.mem, e->b.mem_se, e->b.mem_si, first, count, start, avail, tmp.data());
    if (rc) return rc;
    for (int i = 0; i < count; ++i) memcpy(dst + (size_t)i * n, &tmp[(size_t)i * avail], sizeof(double) * avail);
  }
  return ZAB_OK;
}

int zab_write_mem(zab_engine* e, int32_t first, int32_t count, int64_t start, int64_t n, const double* src) {
  if (!e || !src || !range_ok(e, first, count) || start < 0 || n < 0) return fail(ZAB_E_ARG, "zab_write_mem: bad argument");
  if (start + n > e->b.mem_cap) return fail(ZAB_E_MEM_OVERFLOW, "zab_write_mem: [%lld,+%lld) beyond mem_cap %lld", (long long)start, (long long)n, (long long)e->b.mem_cap);
  if (!n || !count) return ZAB_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  const ZabBatch& b = e->b;
  if (b.mem_se == 1) {
    HIP_TRY(hipMemcpy2D(b.mem + (int64_t)first * b.mem_si + start, sizeof(double) * b.mem_si, src, sizeof(double) * n, sizeof(double) * n, count, hipMemcpyHostToDevice));
  } else {
    std::vector<double> tmp((size_t)n * count);
    for (int64_t k = 0; k < n; ++k) for (int i = 0; i < count; ++i) tmp[(size_t)k * count + i] = src[(size_t)i * n + k];
    HIP_TRY(hipMemcpy2D(b.mem + start * b.mem_se + first, sizeof(double) * b.mem_se, tmp.data(), sizeof(double) * count, sizeof(double) * count, n, hipMemcpyHostToDevice));
  }
  std::vector<int64_t> hi((size_t)count);
  HIP_TRY(hipMemcpy(hi.data(), b.mem_high + first, sizeof(int64_t) * count, hipMemcpyDeviceToHost));
  for (auto& h : hi) if (h < start + n) h = start + n;
  HIP_TRY(hipMemcpy(b.mem_high + first, hi.data(), sizeof(int64_t) * count, hipMemcpyHostToDevice));
  e->b.epoch++;
  return ZAB_OK;
}

int zab_read_mem_high(zab_engine* e, int32_t first, int32_t count, int64_t* dst) {
  if (!e || !dst || !range_ok(e, first, count)) return fail(ZAB_E_ARG, "zab_read_mem_high: bad argument");
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(dst, e->b.mem_high + first, sizeof(int64_t) * count, hipMemcpyDeviceToHost));
  return ZAB_OK;
}

int zab_gmem_read(zab_engine* e, int64_t start, int64_t n, double* dst) {
  if (!e || !dst || start < 0 || n < 0) return fail(ZAB_E_ARG, "zab_gmem_read: bad argument");
  if (!e->gmem_cells) return fail(ZAB_E_STATE, "zab_gmem_read: leaf %s has no gmem", e->mod->name);
  if ((uint64_t)(start + n) > e->gmem_cell_count) return fail(ZAB_E_ARG, "zab_gmem_read: range beyond %llu cells", (unsigned long long)e->gmem_cell_count);
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(dst, e->gmem_cells + start, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return ZAB_OK;
}
int zab_gmem_write(zab_engine* e, int64_t start, int64_t n, const double* src) {
  if (!e || !src || start < 0 || n < 0) return fail(ZAB_E_ARG, "zab_gmem_write: bad argument");
  if (!e->gmem_cells) return fail(ZAB_E_STATE, "zab_gmem_write: leaf %s has no gmem", e->mod->name);
  if ((uint64_t)(start + n) > e->gmem_cell_count) return fail(ZAB_E_ARG, "zab_gmem_write: range beyond %llu cells", (unsigned long long)e->gmem_cell_count);
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(e->gmem_cells + start, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  return ZAB_OK;
}
int zab_gmem_seq(zab_engine* e, int64_t page, uint64_t* out) {
  if (!e || !out) return fail(ZAB_E_ARG, "zab_gmem_seq: bad argument");
  if (!e->gmem_cells) return fail(ZAB_E_STATE, "zab_gmem_seq: leaf %s has no gmem", e->mod->name);
  HIP_TRY(hipStreamSynchronize(e->stream));
  const unsigned long long* src = page < 0 ? e->gmem_global_seq : e->gmem_page_seq + page;
  if (page >= (int64_t)(e->gmem_cell_count / 1024)) return fail(ZAB_E_ARG, "zab_gmem_seq: page out of range");
  unsigned long long v = 0;
  HIP_TRY(hipMemcpy(&v, src, sizeof v, hipMemcpyDeviceToHost));
  *out = v;
  return ZAB_OK;
}

// sample pool: entries + packed float32 arena, uploaded once, read-only on the device (src/DspJsfxSamplePool.h:55-79)
struct ZabPoolEntryDev { uint64_t offset_items; uint32_t frames, sample_rate, channels; float peak, rms; uint32_t pad; };
struct ZabPoolViewDev { const float* audio; uint64_t audio_items; const ZabPoolEntryDev* entries; uint32_t n_entries, generation; };
int zab_pool_upload(zab_engine* e, int32_t n_entries, const zab_pool_entry* entries, const float* audio, int64_t audio_items) {
  if (!e || n_entries < 0 || audio_items < 0 || (n_entries && !entries) || (audio_items && !audio))
    return fail(ZAB_E_ARG, "zab_pool_upload: bad argument");
  if (!e->mod->uses_pool) return fail(ZAB_E_STATE, "zab_pool_upload: leaf %s does not read the sample pool", e->mod->name);
  std::vector<ZabPoolEntryDev> dev((size_t)n_entries);
  for (int i = 0; i < n_entries; ++i) {
    const zab_pool_entry& s = entries[i];
    if (s.channels <= 0 || s.frames < 0 || s.offset_items < 0 || s.offset_items + (int64_t)s.frames * s.channels > audio_items)
      return fail(ZAB_E_ARG, "zab_pool_upload: entry %d lies outside the arena", i);
    dev[(size_t)i] = ZabPoolEntryDev{(uint64_t)s.offset_items, (uint32_t)s.frames, (uint32_t)s.sample_rate, (uint32_t)s.channels, s.peak, s.rms, 0};
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  float* d_audio = nullptr; ZabPoolEntryDev* d_ent = nullptr; ZabPoolViewDev* d_view = nullptr;
  int rc;
  if ((rc = e->alloc(&d_audio, (size_t)(audio_items ? audio_items : 1))) || (rc = e->alloc(&d_ent, (size_t)(n_entries ? n_entries : 1))) ||
      (rc = e->alloc(&d_view, 1)))
    return rc;
  if (audio_items) HIP_TRY(hipMemcpyAsync(d_audio, audio, sizeof(float) * (size_t)audio_items, hipMemcpyHostToDevice, e->stream));
  if (n_entries) HIP_TRY(hipMemcpyAsync(d_ent, dev.data(), sizeof(ZabPoolEntryDev) * dev.size(), hipMemcpyHostToDevice, e->stream));
  ZabPoolViewDev v{d_audio, (uint64_t)audio_items, d_ent, (uint32_t)n_entries, ++e->pool_generation};
  HIP_TRY(hipMemcpyAsync(d_view, &v, sizeof v, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->b.pool = d_view;          // earlier generations stay allocated until zab_destroy (readers may still hold them)
  e->b.epoch++;
  return ZAB_OK;
}

int zab_file_slot_set(zab_engine* e, int32_t slot, int32_t channels, double sample_rate, const double* items, int64_t n_items) {
  if (!e || slot < 0 || slot >= 16 || n_items < 0 || (n_items && !items) || channels < 0)
    return fail(ZAB_E_ARG, "zab_file_slot_set: bad argument");
  if (!e->mod->uses_files) return fail(ZAB_E_STATE, "zab_file_slot_set: leaf %s has no file_*() calls", e->mod->name);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  ZabFileSlotDev& f = e->h_files.slot[slot];
  if (!items) {                            // unassign (file_open fails again); data of open handles stays allocated
    f = ZabFileSlotDev{};
  } else {
    double* d = nullptr;
    int rc = e->alloc(&d, (size_t)(n_items ? n_items : 1));
    if (rc) return rc;
    if (n_items) HIP_TRY(hipMemcpyAsync(d, items, sizeof(double) * (size_t)n_items, hipMemcpyHostToDevice, e->stream));
    f = ZabFileSlotDev{d, n_items, channels, 1, sample_rate};
  }
  HIP_TRY(hipMemcpyAsync(e->d_files, &e->h_files, sizeof e->h_files, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->b.epoch++;
  return ZAB_OK;
}

// ---- RIFF/WAVE ingestion: host work, no device code -------------------------------------------------------------
namespace {
uint32_t rd_u32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t rd_u16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
}  // namespace

int zab_wav_read(const char* path, zab_wav_info* info, float** out) {
  if (!path || !info || !out) return fail(ZAB_E_ARG, "zab_wav_read: bad argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(ZAB_E_ARG, "zab_wav_read: cannot open %s", path);
  std::vector<unsigned char> buf;
  {
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
  }
  if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) != 0 || memcmp(buf.data() + 8, "WAVE", 4) != 0)
    return fail(ZAB_E_ARG, "zab_wav_read: %s is not a RIFF/WAVE file", path);
  uint32_t fmt_tag = 0, channels = 0, rate = 0, bits = 0, block_align = 0;
  const unsigned char* data = nullptr;
  size_t data_len = 0;
  for (size_t pos = 12; pos + 8 <= buf.size();) {            // chunks: id, size, payload, pad byte to an even length
    const unsigned char* ck = buf.data() + pos;
    size_t len = rd_u32(ck + 4);
    const size_t body = pos + 8;
    if (body + len > buf.size()) len = buf.size() - body;     // (a truncated last chunk: take what is there)
    if (memcmp(ck, "fmt ", 4) == 0 && len >= 16) {
      fmt_tag = rd_u16(ck + 8); channels = rd_u16(ck + 10); rate = rd_u32(ck + 12); block_align = rd_u16(ck + 20); bits = rd_u16(ck + 22);
      if (fmt_tag == 0xFFFE && len >= 40) fmt_tag = rd_u16(ck + 8 + 24);      // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first word
    } else if (memcmp(ck, "data", 4) == 0 && !data) {
      data = buf.data() + body; data_len = len;
    }
    pos = body + len + (len & 1);
  }
  const bool is_float = fmt_tag == 3;
  if (!data || channels == 0 || (fmt_tag != 1 && fmt_tag != 3) ||
      !((!is_float && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (is_float && (bits == 32 || bits == 64))))
    return fail(ZAB_E_ARG, "zab_wav_read: %s: unsupported format (tag %u, %u bits, %u channels)", path, fmt_tag, bits, channels);
  const size_t bps = bits / 8, frame_bytes = block_align >= bps * channels ? block_align : bps * channels;
  const size_t frames = data_len / frame_bytes;
  float* o = (float*)malloc(sizeof(float) * (frames * channels ? frames * channels : 1));
  if (!o) return fail(ZAB_E_ARG, "zab_wav_read: out of memory");
  for (size_t t = 0; t < frames; ++t) {
    for (uint32_t c = 0; c < channels; ++c) {
      const unsigned char* p = data + t * frame_bytes + c * bps;
      float v;
      if (is_float && bits == 32) { uint32_t u = rd_u32(p); memcpy(&v, &u, 4); }
      else if (is_float) { uint64_t u = (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); double d; memcpy(&d, &u, 8); v = (float)d; }
      else if (bits == 8) v = (float)((int)p[0] - 128) * (1.0f / 128.0f);
      else if (bits == 16) v = (float)(int16_t)rd_u16(p) * (1.0f / 32768.0f);
      else if (bits == 24) v = (float)((int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8) * (1.0f / 8388608.0f);
      else v = (float)((double)(int32_t)rd_u32(p) * (1.0 / 2147483648.0));
      o[t * channels + c] = v;
    }
  }
  info->channels = (int32_t)channels; info->sample_rate = (int32_t)rate; info->bits = (int32_t)bits; info->is_float = is_float ? 1 : 0;
  info->frames = (int64_t)frames;
  *out = o;
  return ZAB_OK;
}

void zab_wav_free(float* p) { free(p); }

int zab_file_slot_load_wav(zab_engine* e, int32_t slot, const char* path, zab_wav_info* info) {
  zab_wav_info wi{};
  float* a = nullptr;
  int rc = zab_wav_read(path, &wi, &a);
  if (rc) return rc;
  std::vector<double> items((size_t)wi.frames * wi.channels);
  for (size_t i = 0; i < items.size(); ++i) items[i] = (double)a[i];
  zab_wav_free(a);
  if (info) *info = wi;
  return zab_file_slot_set(e, slot, wi.channels, (double)wi.sample_rate, items.data(), (int64_t)items.size());
}

int zab_pool_upload_wav(zab_engine* e, int32_t n_files, const char* const* paths) {
  if (!e || n_files < 0 || (n_files && !paths)) return fail(ZAB_E_ARG, "zab_pool_upload_wav: bad argument");
  std::vector<zab_pool_entry> ents;
  std::vector<float> audio;
  for (int i = 0; i < n_files; ++i) {
    zab_wav_info wi{};
    float* a = nullptr;
    int rc = zab_wav_read(paths[i], &wi, &a);
    if (rc) return rc;
    const size_t n = (size_t)wi.frames * wi.channels;
    double sq = 0.0; float peak = 0.0f;
    for (size_t k = 0; k < n; ++k) { const float v = a[k] < 0 ? -a[k] : a[k]; peak = v > peak ? v : peak; sq += (double)a[k] * a[k]; }
    ents.push_back(zab_pool_entry{(int64_t)audio.size(), (int32_t)wi.frames, wi.sample_rate, wi.channels, peak, n ? (float)sqrt(sq / (double)n) : 0.0f});
    audio.insert(audio.end(), a, a + n);
    zab_wav_free(a);
  }
  return zab_pool_upload(e, n_files, ents.data(), audio.data(), (int64_t)audio.size());
}

int zab_device_alloc(zab_engine* e, int64_t bytes, void** out) {
  if (!e || !out || bytes < 0) return fail(ZAB_E_ARG, "zab_device_alloc: bad argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipMalloc(out, (size_t)(bytes ? bytes : 1)));
  return ZAB_OK;
}
int zab_device_free(zab_engine* e, void* p) {
  if (!e) return fail(ZAB_E_ARG, "zab_device_free: null engine");
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipFree(p));
  return ZAB_OK;
}
int zab_device_upload(zab_engine* e, void* dst, const void* src, int64_t bytes) {
  if (!e) return fail(ZAB_E_ARG, "zab_device_upload: null engine");
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return ZAB_OK;
}
int zab_device_download(zab_engine* e, void* dst, const void* src, int64_t bytes) {
  if (!e) return fail(ZAB_E_ARG, "zab_device_download: null engine");
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return ZAB_OK;
}
int zab_device_noise(zab_engine* e, void* dst, int64_t frames, int64_t frame_stride, uint64_t id_offset) {
  if (!e || !dst || frames < 0 || frame_stride < frames) return fail(ZAB_E_ARG, "zab_device_noise: bad argument");
  const int n = e->b.n_inst, nch = e->mod->nch;
  hipLaunchKernelGGL(zab_k_noise, dim3((n + 63) / 64), dim3(64), 0, e->stream, (float*)dst, n, nch, frames, frame_stride, id_offset);
  HIP_TRY(hipGetLastError());
  return ZAB_OK;
}

int zab_last_timing(zab_engine* e, double* kernel_ms, int32_t* launches) {
  if (!e || !e->timing_valid) return fail(ZAB_E_STATE, "zab_last_timing: no zab_process yet");
  const int slot = (int)((e->n_process - 1) % zab_engine::kTimingSlots);
  HIP_TRY(hipEventSynchronize(e->ev1[slot]));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e->ev0[slot], e->ev1[slot]));
  if (kernel_ms) *kernel_ms = ms;
  if (launches) *launches = e->launches;
  return ZAB_OK;
}

int zab_timing_history(zab_engine* e, double* kernel_ms, int32_t max_entries) {
  if (!e || !kernel_ms || max_entries < 0) return fail(ZAB_E_ARG, "zab_timing_history: bad argument");
  HIP_TRY(hipStreamSynchronize(e->stream));
  uint64_t have = e->n_process < (uint64_t)zab_engine::kTimingSlots ? e->n_process : (uint64_t)zab_engine::kTimingSlots;
  if (have > (uint64_t)max_entries) have = (uint64_t)max_entries;
  for (uint64_t k = 0; k < have; ++k) {            // oldest of the returned window first
    const uint64_t idx = e->n_process - have + k;
    const int slot = (int)(idx % zab_engine::kTimingSlots);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0[slot], e->ev1[slot]));
    kernel_ms[k] = ms;
  }
  return (int)have;
}
void* zab_stream(zab_engine* e) { return e ? (void*)e->stream : nullptr; }

// ---- single-instance state exchange + raw section calls (the jsfx_* compatibility shim sits on these) -----------------
extern "C++" {
#define HIP_TRYW(w, expr)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return fail(ZAB_E_HIP, "[%s] %s failed: %s", w, #expr, hipGetErrorString(e_));    \
  } while (0)
static int ensure_state_stage(zab_engine* e, int64_t bytes) {
  const char* what = "staging";
  if (bytes <= e->state_stage_bytes) return ZAB_OK;
  if (e->state_stage) (void)hipFree(e->state_stage);
  e->state_stage = nullptr; e->state_stage_bytes = 0;
  HIP_TRYW(what, hipMalloc(&e->state_stage, (size_t)bytes));
  e->state_stage_bytes = bytes;
  return ZAB_OK;
}
template <class T>
static int put_strided(zab_engine* e, const char* what, T* base, int64_t se, int64_t n, const T* src) {
  if (!src || n <= 0) return ZAB_OK;
  if (se == 1) { HIP_TRYW(what, hipMemcpyAsync(base, src, sizeof(T) * n, hipMemcpyHostToDevice, e->stream)); return ZAB_OK; }
  int rc = ensure_state_stage(e, (int64_t)sizeof(T) * n);
  if (rc) return rc;
  HIP_TRYW(what, hipMemcpyAsync(e->state_stage, src, sizeof(T) * n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(zab_k_scatter<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, base, se, n, (const T*)e->state_stage);
  HIP_TRYW(what, hipGetLastError());
  HIP_TRYW(what, hipStreamSynchronize(e->stream));      // the staging buffer is reused by the next field
  return ZAB_OK;
}
template <class T>
static int get_strided(zab_engine* e, const char* what, const T* base, int64_t se, int64_t n, T* dst) {
  if (!dst || n <= 0) return ZAB_OK;
  if (se == 1) { HIP_TRYW(what, hipMemcpyAsync(dst, base, sizeof(T) * n, hipMemcpyDeviceToHost, e->stream)); HIP_TRYW(what, hipStreamSynchronize(e->stream)); return ZAB_OK; }
  int rc = ensure_state_stage(e, (int64_t)sizeof(T) * n);
  if (rc) return rc;
  hipLaunchKernelGGL(zab_k_gather<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, base, se, n, (T*)e->state_stage);
  HIP_TRYW(what, hipGetLastError());
  HIP_TRYW(what, hipMemcpyAsync(dst, e->state_stage, sizeof(T) * n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRYW(what, hipStreamSynchronize(e->stream));
  return ZAB_OK;
}
}  // extern "C++"

int zab_state_upload(zab_engine* e, int32_t inst, const zab_host_state* h) {
  if (!e || !h || inst < 0 || inst >= e->b.n_inst) return fail(ZAB_E_ARG, "zab_state_upload: bad argument");
  if (!host_state_ok(h)) return fail(ZAB_E_ARG, "zab_state_upload: zab_host_state.struct_size %llu is not a size this library knows (host ABI %d)", (unsigned long long)h->struct_size, ZAB_HOST_ABI);
  HIP_TRY(hipSetDevice(e->cfg.device));
  ZabBatch& b = e->b;
  int rc;
  if ((rc = put_strided(e, "spl", b.spl + inst * b.sl_si, b.sl_se, 64, h->spl))) return rc;
  if ((rc = put_strided(e, "sliders", b.sliders + inst * b.sl_si, b.sl_se, 64, h->sliders))) return rc;
  if ((rc = put_strided(e, "vars", b.vars + inst * b.var_si, b.var_se, e->mod->nvars, h->vars))) return rc;
  if (h->sliders) {                 // the host's state object is what it pushed last
    if (e->last_pushed.empty()) { e->last_pushed.assign((size_t)b.n_inst * 64, 0.0); e->pushed_valid.assign((size_t)b.n_inst, 0); }
    memcpy(&e->last_pushed[(size_t)inst * 64], h->sliders, 64 * sizeof(double));
    e->pushed_valid[(size_t)inst] = 1;
  }
  if (h->mem) {
    const int64_t n = h->mem_n < b.mem_cap ? (h->mem_n > 0 ? h->mem_n : 0) : b.mem_cap;
    if ((rc = put_strided(e, "mem", b.mem + inst * b.mem_si, b.mem_se, n, h->mem))) return rc;
    // The host image ends at n: cells above it that an earlier run of this instance stored to must read as zeros again
    // (cells at or above the instance's high-water mark never left their zero state).
    int64_t old_high = 0;
    HIP_TRY(hipMemcpyAsync(&old_high, b.mem_high + inst, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (old_high > b.mem_cap) old_high = b.mem_cap;
    if (old_high > n) {
      const int64_t cnt = old_high - n;
      hipLaunchKernelGGL(zab_k_fill<double>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream,
                         b.mem + inst * b.mem_si + n * b.mem_se, b.mem_se, cnt, 0.0);
      HIP_TRY(hipGetLastError());
    }
    int64_t high = h->mem_high ? *h->mem_high : n;
    if (high < 0) high = 0;
    HIP_TRY(hipMemcpyAsync(b.mem_high + inst, &high, 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));       // `high` leaves scope
  }
  if (h->flags) {
    const uint32_t f = *h->flags;
    HIP_TRY(hipMemcpyAsync(b.flags + inst, &f, 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (f & ZAB_FLAG_SLIDER_DIRTY) e->sliders_dirty = true;
  }
  if (ZAB_HS_HAS(h, slider_changes) && h->slider_changes) HIP_TRY(hipMemcpyAsync(b.pend + 3 * (int64_t)b.n_pad + inst, h->slider_changes, 8, hipMemcpyHostToDevice, e->stream));
  if (h->pending_masks) {
    for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpyAsync(b.pend + (int64_t)k * b.n_pad + inst, h->pending_masks + k, 8, hipMemcpyHostToDevice, e->stream));
  }
  if ((rc = put_strided(e, "randMT", b.mt + inst * b.mt_si, b.mt_se, 624, h->rand_mt))) return rc;
  if (h->rand_index) HIP_TRY(hipMemcpyAsync(b.mti + inst, h->rand_index, 4, hipMemcpyHostToDevice, e->stream));
  if (h->slider_visible_mask) HIP_TRY(hipMemcpyAsync(b.vis_mask + inst, h->slider_visible_mask, 8, hipMemcpyHostToDevice, e->stream));
  if (h->slider_visibility_init) HIP_TRY(hipMemcpyAsync(b.vis_init + inst, h->slider_visibility_init, 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->b.epoch++;
  e->lmem_stale = true;
  e->prepared = true;              // the host's state object is authoritative: it has run (or will run) the sections itself
  return ZAB_OK;
}

int zab_state_download(zab_engine* e, int32_t inst, zab_host_state* h) {
  if (!e || !h || inst < 0 || inst >= e->b.n_inst) return fail(ZAB_E_ARG, "zab_state_download: bad argument");
  if (!host_state_ok(h)) return fail(ZAB_E_ARG, "zab_state_download: zab_host_state.struct_size %llu is not a size this library knows (host ABI %d)", (unsigned long long)h->struct_size, ZAB_HOST_ABI);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const ZabBatch& b = e->b;
  int rc;
  if ((rc = get_strided(e, "spl", b.spl + inst * b.sl_si, b.sl_se, 64, h->spl))) return rc;
  if ((rc = get_strided(e, "sliders", b.sliders + inst * b.sl_si, b.sl_se, 64, h->sliders))) return rc;
  if ((rc = get_strided(e, "vars", b.vars + inst * b.var_si, b.var_se, e->mod->nvars, h->vars))) return rc;
  if (h->mem) {
    const int64_t n = h->mem_n < b.mem_cap ? h->mem_n : b.mem_cap;
    if ((rc = get_strided(e, "mem", b.mem + inst * b.mem_si, b.mem_se, n, h->mem))) return rc;
  }
  if (ZAB_HS_HAS(h, slider_changes) && h->slider_changes) HIP_TRY(hipMemcpy(h->slider_changes, b.pend + 3 * (int64_t)b.n_pad + inst, 8, hipMemcpyDeviceToHost));
  if (h->pending_masks) {
    for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpy(h->pending_masks + k, b.pend + (int64_t)k * b.n_pad + inst, 8, hipMemcpyDeviceToHost));
  }
  if ((rc = get_strided(e, "randMT", b.mt + inst * b.mt_si, b.mt_se, 624, h->rand_mt))) return rc;
  if (h->rand_index) HIP_TRY(hipMemcpy(h->rand_index, b.mti + inst, 4, hipMemcpyDeviceToHost));
  if (h->slider_visible_mask) HIP_TRY(hipMemcpy(h->slider_visible_mask, b.vis_mask + inst, 8, hipMemcpyDeviceToHost));
  if (h->slider_visibility_init) HIP_TRY(hipMemcpy(h->slider_visibility_init, b.vis_init + inst, 4, hipMemcpyDeviceToHost));
  if (h->mem_high) HIP_TRY(hipMemcpy(h->mem_high, b.mem_high + inst, 8, hipMemcpyDeviceToHost));
  if (h->flags) HIP_TRY(hipMemcpy(h->flags, b.flags + inst, 4, hipMemcpyDeviceToHost));
  return ZAB_OK;
}

int zab_run_section(zab_engine* e, int32_t section, int32_t samplesblock) {
  if (!e || section < ZAB_SECTION_INIT || section > ZAB_SECTION_SAMPLE) return fail(ZAB_E_ARG, "zab_run_section: bad argument");
  if (!e->mod->launch_section) return fail(ZAB_E_UNSUPPORTED, "zab_run_section: %s has no JSFX sections", e->mod->name);
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipError_t he = e->mod->launch_section(&e->b, section, (double)samplesblock, e->stream);
  if (he != hipSuccess) return fail(ZAB_E_HIP, "section launch failed: %s", hipGetErrorString(he));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->prepared = true;              // the caller drives the @init/@slider sequence itself
  e->b.epoch++;
  return check_device_errors(e, "zab_run_section");
}

int zab_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(ZAB_E_ARG, "zab_host_alloc: null out");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return ZAB_OK;
}
int zab_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return ZAB_OK;
}
int zab_used_fast_path(zab_engine* e) { return e && e->used_fast ? 1 : 0; }
int zab_launch_shape(zab_engine* e, int32_t* instances_per_wave, int32_t* lds_mem_words) {
  if (!e) return fail(ZAB_E_ARG, "zab_launch_shape: null engine");
  if (instances_per_wave) *instances_per_wave = e->b.ipw;
  if (lds_mem_words) *lds_mem_words = e->b.lmem_words;
  return ZAB_OK;
}
int zab_handback_stats(zab_engine* e, uint64_t* instances, uint64_t* frames) {
  if (!e) return fail(ZAB_E_ARG, "zab_handback_stats: null engine");
  uint64_t h[2] = {0, 0};
  if (e->b.resume) {
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpyAsync(h, e->b.resume + e->b.n_pad, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (instances) *instances = h[0];
  if (frames) *frames = h[1];
  return ZAB_OK;
}
const char* zab_last_kernel_name(zab_engine* e) {
  return e ? e->kernel_name.c_str() : "";
}

}  // extern "C"

// ---- one job over several GPUs: instance shards, one host thread + stream per GPU, RCCL only for the end-of-run reduce -------
struct zab_group {
  std::vector<zab_engine*> eng;
  std::vector<int32_t> first, count, device;
  int32_t total = 0;
  int64_t last_frames = 0;
  // RCCL (loaded on demand)
  void* rccl = nullptr;
  std::vector<ncclComm_t> comm;
  std::vector<double*> d_stat;          // [4] per shard: {kernel ms, caller value | kernel ms, units}
  bool rccl_tried = false, rccl_ok = false;
  decltype(&ncclCommInitAll) p_init = nullptr;
  decltype(&ncclAllReduce) p_allreduce = nullptr;
  decltype(&ncclGroupStart) p_gstart = nullptr;
  decltype(&ncclGroupEnd) p_gend = nullptr;
  decltype(&ncclCommDestroy) p_destroy = nullptr;
};

namespace {
// fn(k) for every shard on its own host thread; the first failure (code + text) is reported in the calling thread
int each_shard(zab_group* g, const std::function<int(int)>& fn) {
  const int n = (int)g->eng.size();
  std::vector<int> rc((size_t)n, ZAB_OK);
  std::vector<std::string> msg((size_t)n);
  std::vector<std::thread> th;
  for (int k = 0; k < n; ++k)
    th.emplace_back([&, k] { rc[(size_t)k] = fn(k); if (rc[(size_t)k]) msg[(size_t)k] = zab_last_error(); });
  for (auto& t : th) t.join();
  for (int k = 0; k < n; ++k)
    if (rc[(size_t)k]) return fail(rc[(size_t)k], "shard %d (device %d, instances [%d,+%d)): %s", k, g->device[(size_t)k], g->first[(size_t)k],
                                   g->count[(size_t)k], msg[(size_t)k].c_str());
  return ZAB_OK;
}

bool group_rccl(zab_group* g) {
  if (g->rccl_tried) return g->rccl_ok;
  g->rccl_tried = true;
  for (size_t a = 0; a < g->device.size(); ++a)
    for (size_t b = a + 1; b < g->device.size(); ++b)
      if (g->device[a] == g->device[b]) return false;            // ranks of one communicator must sit on distinct devices
  g->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!g->rccl) g->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!g->rccl) return false;
  g->p_init = (decltype(g->p_init))dlsym(g->rccl, "ncclCommInitAll");
  g->p_allreduce = (decltype(g->p_allreduce))dlsym(g->rccl, "ncclAllReduce");
  g->p_gstart = (decltype(g->p_gstart))dlsym(g->rccl, "ncclGroupStart");
  g->p_gend = (decltype(g->p_gend))dlsym(g->rccl, "ncclGroupEnd");
  g->p_destroy = (decltype(g->p_destroy))dlsym(g->rccl, "ncclCommDestroy");
  if (!g->p_init || !g->p_allreduce || !g->p_gstart || !g->p_gend || !g->p_destroy) return false;
  g->comm.assign(g->device.size(), nullptr);
  if (g->p_init(g->comm.data(), (int)g->device.size(), g->device.data()) != ncclSuccess) { g->comm.clear(); return false; }
  g->d_stat.assign(g->device.size(), nullptr);
  for (size_t k = 0; k < g->device.size(); ++k) {
    if (hipSetDevice(g->device[k]) != hipSuccess || hipMalloc((void**)&g->d_stat[k], 4 * sizeof(double)) != hipSuccess) return false;
  }
  g->rccl_ok = true;
  return true;
}
}  // namespace

extern "C" {

int zab_group_create(const char* module, const zab_config* cfg, const int32_t* devices, int32_t n_devices, zab_group** out) {
  if (!module || !cfg || !devices || !out || n_devices <= 0) return fail(ZAB_E_ARG, "zab_group_create: bad argument");
  if (cfg->n_instances < n_devices) return fail(ZAB_E_ARG, "zab_group_create: %d instances over %d shards leaves a shard empty", cfg->n_instances, n_devices);
  *out = nullptr;
  zab_group* g = new zab_group();
  g->total = cfg->n_instances;
  const int32_t base = cfg->n_instances / n_devices, extra = cfg->n_instances % n_devices;
  for (int32_t k = 0; k < n_devices; ++k) {          // contiguous ranges, sizes differ by at most one (sharding.instance_range)
    const int32_t lo = k * base + (k < extra ? k : extra), cnt = base + (k < extra ? 1 : 0);
    zab_config c = *cfg;
    c.n_instances = cnt;
    c.device = devices[k];
    c.first_instance_id = (cfg->first_instance_id ? cfg->first_instance_id : 1) + (uint64_t)lo;
    zab_engine* e = nullptr;
    const int rc = zab_create(module, &c, &e);
    if (rc) {
      const std::string why = zab_last_error();
      zab_group_destroy(g);
      return fail(rc, "zab_group_create: shard %d on device %d: %s", k, devices[k], why.c_str());
    }
    g->eng.push_back(e); g->first.push_back(lo); g->count.push_back(cnt); g->device.push_back(devices[k]);
  }
  *out = g;
  return ZAB_OK;
}

int zab_group_destroy(zab_group* g) {
  if (!g) return ZAB_OK;
  for (size_t k = 0; k < g->d_stat.size(); ++k) if (g->d_stat[k]) { hipSetDevice(g->device[k]); hipFree(g->d_stat[k]); }
  if (g->p_destroy) for (ncclComm_t c : g->comm) if (c) g->p_destroy(c);
  for (zab_engine* e : g->eng) zab_destroy(e);
  delete g;                               // (librccl stays loaded, like the plugin modules)
  return ZAB_OK;
}

int zab_group_size(zab_group* g) { return g ? (int)g->eng.size() : 0; }

int zab_group_shard(zab_group* g, int32_t k, zab_engine** engine, int32_t* first, int32_t* count) {
  if (!g || k < 0 || k >= (int32_t)g->eng.size()) return fail(ZAB_E_ARG, "zab_group_shard: bad argument");
  if (engine) *engine = g->eng[(size_t)k];
  if (first) *first = g->first[(size_t)k];
  if (count) *count = g->count[(size_t)k];
  return ZAB_OK;
}

int zab_group_set_sliders(zab_group* g, int32_t first, int32_t count, const double* values) {
  if (!g || !values) return fail(ZAB_E_ARG, "zab_group_set_sliders: null argument");
  const bool bcast = first == 0 && count == 0;
  if (!bcast && (first < 0 || count < 0 || (int64_t)first + count > g->total)) return fail(ZAB_E_ARG, "zab_group_set_sliders: range outside the job");
  for (size_t k = 0; k < g->eng.size(); ++k) {
    if (bcast) { const int rc = zab_set_sliders(g->eng[k], 0, 0, values); if (rc) return rc; continue; }
    const int32_t lo = std::max(first, g->first[k]), hi = std::min(first + count, g->first[k] + g->count[k]);
    if (lo >= hi) continue;
    const int rc = zab_set_sliders(g->eng[k], lo - g->first[k], hi - lo, values + (size_t)(lo - first) * 64);
    if (rc) return rc;
  }
  return ZAB_OK;
}

int zab_group_prepare(zab_group* g) {
  if (!g) return fail(ZAB_E_ARG, "zab_group_prepare: null group");
  return each_shard(g, [g](int k) { return zab_prepare(g->eng[(size_t)k]); });
}

int zab_group_process(zab_group* g, const void* const* in, void* const* out, int64_t frames, int64_t frame_stride, int32_t block, int32_t placement) {
  if (!g || !in || !out) return fail(ZAB_E_ARG, "zab_group_process: null argument");
  g->last_frames = frames;
  return each_shard(g, [=](int k) { return zab_process(g->eng[(size_t)k], in[k], out[k], frames, frame_stride, block, placement); });
}

int zab_group_sync(zab_group* g) {
  if (!g) return fail(ZAB_E_ARG, "zab_group_sync: null group");
  return each_shard(g, [g](int k) { return zab_sync(g->eng[(size_t)k]); });
}

int zab_group_reduce(zab_group* g, const double* shard_values, zab_group_stats* out) {
  if (!g || !out) return fail(ZAB_E_ARG, "zab_group_reduce: null argument");
  const size_t n = g->eng.size();
  std::vector<double> ms(n, 0.0), units(n, 0.0);
  for (size_t k = 0; k < n; ++k) {
    int32_t launches = 0;
    if (g->eng[k]->timing_valid) { const int rc = zab_last_timing(g->eng[k], &ms[k], &launches); if (rc) return rc; }
    units[k] = (double)g->count[k] * g->eng[k]->mod->nch * (double)g->last_frames;
  }
  memset(out, 0, sizeof *out);
  out->n_shards = (int32_t)n;
  if (group_rccl(g)) {
    // {max: kernel ms, caller value} and {sum: kernel ms, units}: two all-reduces per shard on its own stream, grouped
    std::vector<double> hs(4 * n);            // (outlives the asynchronous copies: they are joined below, before it goes)
    for (size_t k = 0; k < n; ++k) {
      double* h = &hs[4 * k];
      h[0] = ms[k]; h[1] = shard_values ? shard_values[k] : 0.0; h[2] = ms[k]; h[3] = units[k];
      HIP_TRY(hipSetDevice(g->device[k]));
      HIP_TRY(hipMemcpyAsync(g->d_stat[k], h, 4 * sizeof(double), hipMemcpyHostToDevice, g->eng[k]->stream));
    }
    if (g->p_gstart() != ncclSuccess) return fail(ZAB_E_HIP, "ncclGroupStart failed");
    for (size_t k = 0; k < n; ++k) {
      if (g->p_allreduce(g->d_stat[k], g->d_stat[k], 2, ncclDouble, ncclMax, g->comm[k], g->eng[k]->stream) != ncclSuccess ||
          g->p_allreduce(g->d_stat[k] + 2, g->d_stat[k] + 2, 2, ncclDouble, ncclSum, g->comm[k], g->eng[k]->stream) != ncclSuccess)
        return fail(ZAB_E_HIP, "ncclAllReduce failed on shard %zu", k);
    }
    if (g->p_gend() != ncclSuccess) return fail(ZAB_E_HIP, "ncclGroupEnd failed");
    double h[4];
    HIP_TRY(hipSetDevice(g->device[0]));
    HIP_TRY(hipMemcpyAsync(h, g->d_stat[0], sizeof h, hipMemcpyDeviceToHost, g->eng[0]->stream));
    HIP_TRY(hipStreamSynchronize(g->eng[0]->stream));
    for (size_t k = 1; k < n; ++k) { HIP_TRY(hipSetDevice(g->device[k])); HIP_TRY(hipStreamSynchronize(g->eng[k]->stream)); }
    out->max_kernel_ms = h[0]; out->max_value = h[1]; out->sum_kernel_ms = h[2]; out->units = h[3];
    out->used_rccl = 1;
    return ZAB_OK;
  }
  for (size_t k = 0; k < n; ++k) {        // shards share a device (or no RCCL on this host): nothing to move between GPUs
    out->max_kernel_ms = std::max(out->max_kernel_ms, ms[k]);
    out->sum_kernel_ms += ms[k];
    out->units += units[k];
    if (shard_values) out->max_value = std::max(out->max_value, shard_values[k]);
  }
  return ZAB_OK;
}

}  // extern "C"
