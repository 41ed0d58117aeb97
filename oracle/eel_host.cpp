// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// JUCE-free host for the reference's own WDL/EEL2 portable VM (compiled from the sources where they
// lie under /root/reference/src/WDL by oracle/Makefile; nothing from the reference is copied here).
// It replays the reference's shadow-runtime harness so that tests can obtain reference outputs:
//
//   * VM flavour, builtin set, compile flags ........ src/YSFXGfxInterpreter.h:23-57,507-529
//                                                      src/WDL/eel2/eelscript.h:362-395 (COMMONFUNCS)
//   * section split ................................... src/YSFXGfxInterpreter.h:153-215
//   * slider(i)=v / spl(i)=v rewrite .................. src/YSFXGfxInterpreter.h:237-432
//   * spl( -> dsp_spl( rename ......................... src/JSFXCorrectnessCheck.h:355-462
//   * hooks dsp_spl/slider/spl/freembuf/memset/
//     sliderchange/slider_automate/slider_show ........ src/JSFXCorrectnessCheck.h:127-139,508-529,663-699
//                                                      src/YSFXGfxInterpreter.h:995-1014,1721-1855
//   * prime sequence (sliders -> @init -> alias -> @slider)  src/JSFXCorrectnessCheck.h:732-750
//   * block loop (@block, pending masks -> @slider, per-sample lock step)
//                                                      src/JSFXJuceProcessor.cpp:3589-3657
//   * write-trace high-water mark ..................... src/JSFXCorrectnessCheck.h:497-506
//
// Exposed as a plain C ABI (libeel_oracle.so) so Python tests / bench.py's cpu_baseline leg can drive it.

#define EEL_TARGET_PORTABLE 1
#define EELSCRIPT_NO_FILE 1
#define EELSCRIPT_NO_NET 1
#define EELSCRIPT_NO_MDCT 1
#define EELSCRIPT_NO_PREPROC 1
#define EELSCRIPT_NO_LICE 1

#include "WDL/eel2/ns-eel.h"
#include "WDL/eel2/eelscript.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static std::mutex g_eel_mutex;
extern "C" void NSEEL_HOSTSTUB_EnterMutex() { g_eel_mutex.lock(); }
extern "C" void NSEEL_HOSTSTUB_LeaveMutex() { g_eel_mutex.unlock(); }

namespace {

inline int64_t trunc_like_aot(double v) { return (int64_t)(v + 1.0e-5); }
inline bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

struct Sections { std::string init, slider, block, sample; };

bool starts_ci(const std::string& s, const char* sec) {
  size_t n = std::strlen(sec);
  if (s.size() < n + 1 || s[0] != '@') return false;
  for (size_t i = 0; i < n; ++i)
    if (std::tolower((unsigned char)s[i + 1]) != sec[i]) return false;
  return true;
}

Sections split_sections(const std::string& text) {
  Sections out;
  std::string* cur = nullptr;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t e = text.find('\n', pos);
    std::string line = text.substr(pos, (e == std::string::npos ? text.size() : e) - pos);
    pos = (e == std::string::npos) ? text.size() : e + 1;
    size_t f = line.find_first_not_of(" \t\r");
    if (f != std::string::npos && line[f] == '@') {
      std::string lt = line.substr(f);
      if (starts_ci(lt, "init")) cur = &out.init;
      else if (starts_ci(lt, "slider")) cur = &out.slider;
      else if (starts_ci(lt, "block")) cur = &out.block;
      else if (starts_ci(lt, "sample")) cur = &out.sample;
      else cur = nullptr;  // @gfx, @serialize, unknown: not run by this host
      continue;
    }
    if (cur) { cur->append(line); cur->push_back('\n'); }
  }
  return out;
}

// Scanner state shared by the two text transforms: skip comments and string literals verbatim.
struct Skipper {
  const std::string& in; std::string& out; size_t& i;
  bool skip() {
    char c = in[i], n = (i + 1 < in.size()) ? in[i + 1] : 0;
    if (c == '/' && n == '/') { while (i < in.size() && in[i] != '\n') out.push_back(in[i++]); return true; }
    if (c == '/' && n == '*') {
      out.push_back(in[i++]); out.push_back(in[i++]);
      while (i < in.size()) {
        if (in[i] == '*' && i + 1 < in.size() && in[i + 1] == '/') { out.push_back(in[i++]); out.push_back(in[i++]); break; }
        out.push_back(in[i++]);
      }
      return true;
    }
    if (c == '"' || c == '\'') {
      char q = c; out.push_back(in[i++]);
      while (i < in.size()) {
        if (in[i] == '\\' && i + 1 < in.size()) { out.push_back(in[i++]); out.push_back(in[i++]); continue; }
        char d = in[i]; out.push_back(in[i++]);
        if (d == q) break;
      }
      return true;
    }
    return false;
  }
};

// name(args) = rhs;   ->   name(args, rhs);      (plain '=' only)
bool rewrite_call_assign(const std::string& in, size_t& i, std::string& out, const char* name) {
  size_t nl = std::strlen(name);
  if (i + nl + 1 >= in.size() || in.compare(i, nl, name) != 0) return false;
  if (i > 0 && ident_char(in[i - 1])) return false;
  if (in[i + nl] != '(') return false;
  size_t p = i + nl + 1; int depth = 1; bool s = false; char q = 0;
  while (p < in.size() && depth > 0) {
    char c = in[p];
    if (s) { if (c == '\\' && p + 1 < in.size()) { p += 2; continue; } if (c == q) s = false; ++p; continue; }
    if (c == '"' || c == '\'') { s = true; q = c; }
    else if (c == '(') ++depth;
    else if (c == ')') --depth;
    ++p;
  }
  if (depth != 0) return false;
  size_t close = p - 1, a = p;
  while (a < in.size() && std::isspace((unsigned char)in[a])) ++a;
  if (a >= in.size() || in[a] != '=' || (a + 1 < in.size() && in[a + 1] == '=')) return false;
  size_t rs = a + 1;
  while (rs < in.size() && std::isspace((unsigned char)in[rs])) ++rs;
  size_t r = rs; int par = 0, br = 0, cr = 0; s = false;
  while (r < in.size()) {
    char c = in[r];
    if (s) { if (c == '\\' && r + 1 < in.size()) { r += 2; continue; } if (c == q) s = false; ++r; continue; }
    if (c == ';' && !par && !br && !cr) break;
    if (c == '"' || c == '\'') { s = true; q = c; }
    else if (c == '(') ++par; else if (c == ')' && par > 0) --par;
    else if (c == '[') ++br; else if (c == ']' && br > 0) --br;
    else if (c == '{') ++cr; else if (c == '}' && cr > 0) --cr;
    ++r;
  }
  out.append(name); out.push_back('(');
  out.append(in, i + nl + 1, close - (i + nl + 1));
  out.append(", ");
  out.append(in, rs, r - rs);
  out.push_back(')');
  if (r < in.size() && in[r] == ';') { out.push_back(';'); ++r; }
  i = r;
  return true;
}

std::string portable_rewrite(const std::string& in) {
  std::string out; out.reserve(in.size() + 64);
  for (size_t i = 0; i < in.size();) {
    Skipper sk{in, out, i};
    if (sk.skip()) continue;
    if (in[i] == 's' && (rewrite_call_assign(in, i, out, "slider") || rewrite_call_assign(in, i, out, "spl"))) continue;
    out.push_back(in[i++]);
  }
  return out;
}

std::string rename_calls(const std::string& in, const char* from, const char* to) {
  std::string out; out.reserve(in.size() + 64);
  size_t fl = std::strlen(from);
  for (size_t i = 0; i < in.size();) {
    Skipper sk{in, out, i};
    if (sk.skip()) continue;
    if (i + fl < in.size() && in.compare(i, fl, from) == 0 && !(i > 0 && ident_char(in[i - 1])) && in[i + fl] == '(') {
      out.append(to); i += fl; continue;
    }
    out.push_back(in[i++]);
  }
  return out;
}

}  // namespace

struct eelo : public eelScriptInst {
  Sections sec;
  NSEEL_CODEHANDLE c_init = nullptr, c_slider = nullptr, c_block = nullptr, c_sample = nullptr;
  bool ready = false;
  std::string error;
  EEL_F* splp[64] = {};
  EEL_F* sliderp[64] = {};
  EEL_F* aliasp[64] = {};
  EEL_F *p_srate = nullptr, *p_samplesblock = nullptr;
  uint64_t m_change = 0, m_automate = 0, m_automate_end = 0, m_visible = ~0ull;
  int64_t high = 0;
  int64_t ram_size = 0;

  EEL_F* var(const char* n) { return m_vm ? NSEEL_VM_regvar(m_vm, n) : nullptr; }

  void ensure_ram(int64_t need) {
    if (!m_vm || need <= ram_size) return;
    need = std::min<int64_t>(need, 0x7fffffff);
    NSEEL_VM_setramsize(m_vm, (unsigned)need);
    ram_size = need;
  }

  uint64_t mask_from_arg(EEL_F* p, double v) {
    for (int i = 0; i < 64; ++i) if (sliderp[i] == p) return 1ull << i;
    if (v <= 0.0) return 0;
    long long m = std::llround(v);
    return m <= 0 ? 0 : (uint64_t)m;
  }

  static EEL_F NSEEL_CGEN_CALL f_zero(void*, INT_PTR, EEL_F**) { return 0.0; }
  static EEL_F NSEEL_CGEN_CALL f_one(void*, INT_PTR, EEL_F**) { return 1.0; }
  static EEL_F NSEEL_CGEN_CALL f_neg(void*, INT_PTR, EEL_F**) { return -1.0; }

  static EEL_F NSEEL_CGEN_CALL f_dsp_spl(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 1) return 0.0;
    int64_t idx = trunc_like_aot(*a[0]);
    if (idx < 0 || idx >= 64 || !s->splp[idx]) return np >= 2 ? *a[1] : 0.0;
    if (np >= 2) *s->splp[idx] = *a[1];
    return *s->splp[idx];
  }
  static EEL_F NSEEL_CGEN_CALL f_slider(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 1) return 0.0;
    int64_t idx = trunc_like_aot(*a[0]);
    if (idx < 1 || idx > 64 || !s->sliderp[idx - 1]) return np >= 2 ? *a[1] : 0.0;
    if (np >= 2) *s->sliderp[idx - 1] = *a[1];
    return *s->sliderp[idx - 1];
  }
  static EEL_F NSEEL_CGEN_CALL f_spl_inert(void*, INT_PTR np, EEL_F** a) { return np >= 2 ? *a[1] : 0.0; }
  static EEL_F NSEEL_CGEN_CALL f_sliderchange(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 1) return 0.0;
    s->m_change |= s->mask_from_arg(a[0], *a[0]);
    return 0.0;
  }
  static EEL_F NSEEL_CGEN_CALL f_slider_automate(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 1) return 0.0;
    uint64_t m = s->mask_from_arg(a[0], *a[0]);
    if (!m) return 0.0;
    if (np >= 2 && *a[1] != 0.0) s->m_automate_end |= m; else s->m_automate |= m;
    return 0.0;
  }
  static EEL_F NSEEL_CGEN_CALL f_slider_show(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 1) return 0.0;
    uint64_t m = s->mask_from_arg(a[0], *a[0]);
    if (!m) return 0.0;
    if (np >= 2) {
      double v = *a[1];
      if (v == -1.0) s->m_visible ^= m; else if (v <= 0.0) s->m_visible &= ~m; else s->m_visible |= m;
    }
    return (EEL_F)(double)(s->m_visible & m);
  }
  static EEL_F NSEEL_CGEN_CALL f_memset(void* o, INT_PTR np, EEL_F** a) {
    auto* s = (eelo*)o; if (!s || np < 3) return 0.0;
    int64_t dst = (int64_t)(*a[0] + 1.0e-5), len = (int64_t)(*a[2] + 1.0e-5);
    if (dst < 0) dst = 0;
    if (len <= 0) return (EEL_F)dst;
    s->ensure_ram(dst + len);
    EEL_F v = *a[1];
    int64_t pos = dst, rem = len;
    while (rem > 0 && pos <= 0xffffffffLL) {
      int valid = 0;
      EEL_F* p = NSEEL_VM_getramptr(s->m_vm, (unsigned)pos, &valid);
      if (!p || valid <= 0) break;
      int n = (int)std::min<int64_t>(valid, rem);
      for (int i = 0; i < n; ++i) p[i] = v;
      pos += n; rem -= n;
    }
    s->high = std::max(s->high, dst + len);
    return (EEL_F)dst;
  }
  static void write_trace(void* ctx, EEL_F* addr, unsigned int count) {
    auto* s = (eelo*)ctx;
    if (!s || !s->m_vm || !addr || !count) return;
    unsigned idx = 0;
    if (NSEEL_VM_GetRAMIndexForPtr(s->m_vm, addr, &idx, nullptr))
      s->high = std::max<int64_t>(s->high, (int64_t)idx + count);
  }

  static void register_globals() {
    NSEEL_init();
    eelScriptInst::init();
    // Drawing / UI builtins: the oracle records nothing, it only needs the names to resolve
    // (same names the reference registers, src/YSFXGfxInterpreter.h:974-991, YSFXGfxCommCompat.h:79).
    static const char* gfx[] = {"gfx_set", "gfx_rect", "gfx_rectto", "gfx_circle", "gfx_roundrect", "gfx_arc",
      "gfx_triangle", "gfx_line", "gfx_lineto", "gfx_drawstr", "gfx_printf", "gfx_setfont", "gfx_measurestr",
      "gfx_getchar", "gfx_showmenu", "gfx_showmenu_nb_open", "gfx_showmenu_nb_poll", "gfx_showmenu_nb_cancel",
      "gfx_drawnumber", nullptr};
    static const int gfx_min[] = {1, 4, 2, 3, 5, 5, 6, 4, 2, 1, 1, 1, 1, 0, 1, 1, 0, 0, 1};
    for (int i = 0; gfx[i]; ++i) NSEEL_addfunc_varparm_ex(gfx[i], gfx_min[i], 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    // Inert comm / sample-pool / track-name builtins (src/YSFXGfxCommCompat.h:82-151,
    // src/YSFXGfxInterpreter.h:1002-1007): single VM, nothing attached.
    static const char* inert[] = {"instance_uid", "instance_set_name", "instance_get_name", "comm_join",
      "gmem_attach", "gmem_attach_size", "gmem_size", "gmem_get", "gmem_put", "gmem_fill", "gmem_zero", "gmem_copy",
      "gmem_seq", "gmem_page", "msg_subscribe", "msg_unsubscribe", "msg_advertise", "msg_send", "msg_sendto",
      "msg_avail", "msg_kind", "msg_recv", "msg_send_buf", "msg_sendto_buf", "msg_recv_buf", "msg_length",
      "msg_dropped", "msg_clear", "msg_peer_count", "msg_peer_id", "msg_peer_name", "msg_peer_uid", "msg_peer_caps",
      "msg_peer_alive", "sample_pool_from_slot", "sample_pool_set_mode", "sample_pool_set_budget_mb",
      "sample_pool_commit", "sample_pool_state", "sample_pool_selected", "sample_pool_loaded", "sample_pool_failed",
      "sample_pool_ram_mb", "sample_pool_generation", "sample_get", "sample_len", "sample_channels", "sample_srate",
      "sample_peak", "sample_rms", "sample_name", "sample_read", "sample_read_interp", "sample_read2",
      "sample_read2_interp", "sample_preview_bins", "sample_preview_read", "sample_export_mem", "sample_export_mem2",
      "track_name_available", "host_track_name_available", "track_name_seq", "host_track_name_seq", nullptr};
    for (int i = 0; inert[i]; ++i) NSEEL_addfunc_varparm_ex(inert[i], 0, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("instance_id", 0, 0, NSEEL_PProc_THIS, &f_one, nullptr);
    NSEEL_addfunc_varparm_ex("track_name", 1, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("host_track_name", 1, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    // No MIDI / files in this harness: empty queue, missing files.
    NSEEL_addfunc_varparm_ex("midirecv", 4, 1, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("midisend", 4, 1, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_open", 1, 0, NSEEL_PProc_THIS, &f_neg, nullptr);
    NSEEL_addfunc_varparm_ex("file_open_multi", 1, 0, NSEEL_PProc_THIS, &f_neg, nullptr);
    static const char* files[] = {"file_close", "file_rewind", "file_avail", "file_text", "file_multi_count", nullptr};
    for (int i = 0; files[i]; ++i) NSEEL_addfunc_varparm_ex(files[i], 1, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_seek", 2, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_var", 2, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_multi_select", 2, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_riff", 3, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("file_mem", 3, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    // DSP hooks
    NSEEL_addfunc_varparm_ex("sliderchange", 1, 0, NSEEL_PProc_THIS, &f_sliderchange, nullptr);
    NSEEL_addfunc_varparm_ex("slider_automate", 1, 0, NSEEL_PProc_THIS, &f_slider_automate, nullptr);
    NSEEL_addfunc_varparm_ex("slider_show", 1, 0, NSEEL_PProc_THIS, &f_slider_show, nullptr);
    NSEEL_addfunc_varparm_ex("slider", 1, 0, NSEEL_PProc_THIS, &f_slider, nullptr);
    NSEEL_addfunc_varparm_ex("spl", 1, 0, NSEEL_PProc_THIS, &f_spl_inert, nullptr);
    NSEEL_addfunc_varparm_ex("freembuf", 1, 0, NSEEL_PProc_THIS, &f_zero, nullptr);
    NSEEL_addfunc_varparm_ex("dsp_spl", 1, 0, NSEEL_PProc_THIS, &f_dsp_spl, nullptr);
    NSEEL_addfunc_varparm_ex("memset", 3, 1, NSEEL_PProc_THIS, &f_memset, nullptr);
  }

  NSEEL_CODEHANDLE compile_section(const std::string& code, const char* label) {
    std::string pre = rename_calls(portable_rewrite(code.empty() ? std::string("0;") : code), "spl", "dsp_spl");
    const char* err = nullptr;
    NSEEL_CODEHANDLE h = compile_code(pre.empty() ? "0;" : pre.c_str(), &err);
    if (!h && error.empty()) error = std::string(label) + ": " + (err ? err : "unknown EEL compile error");
    return h;
  }

  explicit eelo(const char* text) {
    static std::once_flag once;
    std::call_once(once, &eelo::register_globals);
    for (int i = 0; i < 64; ++i) {
      sliderp[i] = var(("slider" + std::to_string(i + 1)).c_str());
      splp[i] = var(("spl" + std::to_string(i)).c_str());
      if (splp[i]) *splp[i] = 0.0;
    }
    p_srate = var("srate"); p_samplesblock = var("samplesblock");
    if (p_srate) *p_srate = 44100.0;
    if (p_samplesblock) *p_samplesblock = 0.0;
    if (m_vm) NSEEL_VM_SetWriteTrace(m_vm, &eelo::write_trace, this);
    sec = split_sections(text ? text : "");
    c_init = compile_section(sec.init, "@init");
    if (error.empty()) c_slider = compile_section(sec.slider, "@slider");
    if (error.empty()) c_block = compile_section(sec.block, "@block");
    if (error.empty()) c_sample = compile_section(sec.sample, "@sample");
    ready = error.empty();
  }

  void run(NSEEL_CODEHANDLE h) { if (h) NSEEL_code_execute(h); }
  void sync_alias() { for (int i = 0; i < 64; ++i) if (aliasp[i] && sliderp[i]) *aliasp[i] = *sliderp[i]; }
};

extern "C" {

eelo* eelo_create(const char* jsfx_text) { return new eelo(jsfx_text); }
void eelo_destroy(eelo* e) { delete e; }
const char* eelo_error(eelo* e) { return e->ready ? "" : e->error.c_str(); }

void eelo_bind_alias(eelo* e, int idx0, const char* name) {
  if (idx0 >= 0 && idx0 < 64 && name && *name) e->aliasp[idx0] = e->var(name);
}
void eelo_set_sliders(eelo* e, const double* v, int count) {
  for (int i = 0; i < std::min(count, 64); ++i) if (e->sliderp[i]) *e->sliderp[i] = v[i];
  e->sync_alias();
}
void eelo_get_sliders(eelo* e, double* v, int count) {
  for (int i = 0; i < std::min(count, 64); ++i) v[i] = e->sliderp[i] ? *e->sliderp[i] : 0.0;
}

// prepareToPlay analogue: sliders must have been set already (valid inside @init).
void eelo_prepare(eelo* e, double srate, int64_t mem_hint) {
  // The reference sizes the shadow VM's RAM to max(65536, memN of the compiled side) at every block start
  // (src/JSFXCorrectnessCheck.h:259), i.e. to whatever the compiled code has grown its heap to. A standalone oracle has
  // no compiled side to follow, so without a hint it keeps EEL2's own default ceiling (8 Mi items, ns-eel.h:230).
  e->ensure_ram(mem_hint > 0 ? std::max<int64_t>(65536, mem_hint) : (int64_t)NSEEL_RAM_BLOCKS * NSEEL_RAM_ITEMSPERBLOCK);
  e->sync_alias();
  if (e->p_srate) *e->p_srate = srate;
  if (e->p_samplesblock) *e->p_samplesblock = 0.0;
  e->run(e->c_init);
  e->sync_alias();
  e->run(e->c_slider);
}
void eelo_run_slider(eelo* e) { e->sync_alias(); e->run(e->c_slider); }
void eelo_run_block(eelo* e) { e->run(e->c_block); }
void eelo_run_sample(eelo* e) { e->run(e->c_sample); }

// One host block: in/out are planar [nCh][ch_stride] float, n frames used. out may alias in.
void eelo_process_block(eelo* e, const float* in, float* out, int nCh, int n, int64_t ch_stride, double srate) {
  nCh = std::max(0, std::min(nCh, 64));
  if (e->p_srate) *e->p_srate = srate;
  if (e->p_samplesblock) *e->p_samplesblock = (double)n;
  e->m_change = e->m_automate = e->m_automate_end = 0;
  e->ensure_ram(65536);
  e->run(e->c_block);
  if (e->m_change | e->m_automate | e->m_automate_end) e->run(e->c_slider);
  for (int i = 0; i < n; ++i) {
    for (int ch = 0; ch < nCh; ++ch) if (e->splp[ch]) *e->splp[ch] = in ? (EEL_F)in[ch * ch_stride + i] : 0.0;
    e->run(e->c_sample);
    if (out) for (int ch = 0; ch < nCh; ++ch) out[ch * ch_stride + i] = (float)(e->splp[ch] ? *e->splp[ch] : 0.0);
  }
}

// Whole run split into host blocks of `block` frames (last one short).
void eelo_process(eelo* e, const float* in, float* out, int nCh, int64_t frames, int block, double srate) {
  for (int64_t pos = 0; pos < frames; pos += block) {
    int n = (int)std::min<int64_t>(block, frames - pos);
    eelo_process_block(e, in ? in + pos : nullptr, out ? out + pos : nullptr, nCh, n, frames, srate);
  }
}

int eelo_get_var(eelo* e, const char* name, double* out) {
  EEL_F* p = NSEEL_VM_getvar(e->m_vm, name);
  if (!p) { *out = 0.0; return 0; }
  *out = *p; return 1;
}
void eelo_set_var(eelo* e, const char* name, double v) { EEL_F* p = e->var(name); if (p) *p = v; }
double eelo_get_spl(eelo* e, int ch) { return (ch >= 0 && ch < 64 && e->splp[ch]) ? *e->splp[ch] : 0.0; }

int64_t eelo_mem_read(eelo* e, int64_t start, int64_t count, double* dst) {
  int64_t done = 0;
  while (done < count) {
    const int64_t off = start + done;
    const int64_t in_block = NSEEL_RAM_ITEMSPERBLOCK - (off % NSEEL_RAM_ITEMSPERBLOCK);
    const int64_t n = std::min<int64_t>(in_block, count - done);
    int valid = 0;
    EEL_F* p = NSEEL_VM_getramptr_noalloc(e->m_vm, (unsigned)off, &valid);   // never-touched blocks read as zeros
    if (p && valid >= n) std::memcpy(dst + done, p, (size_t)n * sizeof(double));
    else std::memset(dst + done, 0, (size_t)n * sizeof(double));
    done += n;
  }
  return done;
}
int64_t eelo_mem_write(eelo* e, int64_t start, int64_t count, const double* src) {
  e->ensure_ram(start + count);
  int64_t done = 0;
  while (done < count) {
    int valid = 0;
    EEL_F* p = NSEEL_VM_getramptr(e->m_vm, (unsigned)(start + done), &valid);
    if (!p || valid <= 0) break;
    int64_t n = std::min<int64_t>(valid, count - done);
    std::memcpy(p, src + done, (size_t)n * sizeof(double));
    done += n;
  }
  e->high = std::max(e->high, start + done);
  return done;
}
int64_t eelo_mem_high(eelo* e) { return e->high; }
// The write trace is the shadow runtime's instrumentation (src/JSFXCorrectnessCheck.h registers it to follow the touched
// pages); it costs a RAM-block search per store. Timing runs switch it off to see the VM itself; fixtures keep it on.
void eelo_set_write_trace(eelo* e, int on) {
  if (e && e->m_vm) NSEEL_VM_SetWriteTrace(e->m_vm, on ? &eelo::write_trace : nullptr, on ? e : nullptr);
}
void eelo_pending_masks(eelo* e, uint64_t* m3) { m3[0] = e->m_change; m3[1] = e->m_automate; m3[2] = e->m_automate_end; }

// Direct access to the reference FFT (src/WDL/fft.c) for builtin known-answer tests.
void eelo_wdl_fft(double* buf, int len, int isInverse) { WDL_fft_init(); WDL_fft((WDL_FFT_COMPLEX*)buf, len, isInverse); }
void eelo_wdl_real_fft(double* buf, int len, int isInverse) { WDL_fft_init(); WDL_real_fft((WDL_FFT_REAL*)buf, len, isInverse); }
int eelo_wdl_fft_permute(int fftsize, int idx) { WDL_fft_init(); return WDL_fft_permute(fftsize, idx); }

}  // extern "C"
