// TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference's compiled hot path.
//
// A generated translation unit defines ZA_NV / ZA_NCH / ZA_HAS_SAMPLE, includes csrc/zart.h and the zajit
// section code, then this header, giving a scalar f64 CPU build of the same lowering the reference's LLVM AOT
// object has (dsp_jsfx_aot.py:3310-5905) -- the stand-in for that object, which cannot be produced here
// (llvmlite absent). Call sequence restated from:
//   prepareToPlay ............ src/JSFXJuceProcessor.cpp:3297-3318 (sliders -> @init -> alias re-apply -> @slider)
//   jsfx_process_block ....... dsp_jsfx_aot.py:5713-5905 (@block; pending masks -> @slider; per-sample f32<->f64)
//   consumeDspSliderChanges .. src/JSFXJuceProcessor.cpp:3745 (pending masks cleared after each host block)
// Never linked into the product.
#pragma once

#include <stdlib.h>
#include <string.h>
#include <vector>

struct ZaPort {
  ZaState<ZA_NV> s;
  std::vector<double> mem;
  std::vector<uint32_t> mt;
  std::vector<double> fft;
#ifdef ZA_GMEM_PAGE_CELLS
  std::vector<unsigned long long> gcells, gpseq, gpwr;
  unsigned long long gseq = 0;
  ZaGmemView gview;
#endif
#ifdef ZA_POOL_H_INCLUDED
  std::vector<float> paudio;
  std::vector<ZaPoolEntry> pent;
  ZaPoolView pview;
#endif
#ifdef ZA_MSG_H_INCLUDED
  ZaBusView bview;                       // a bus with this one instance on it
  std::vector<ZaMsg> mring, moutbox, minbox;
  uint64_t mseq, mdomain, mlast, mch_hash[ZA_MSG_CHANNELS], mch_caps[ZA_MSG_CHANNELS], mch_dropped[ZA_MSG_CHANNELS];
  uint32_t mch_flags[ZA_MSG_CHANNELS], mout_count, min_count, mout_cells, mlast_len;
  std::vector<double> mout_pay, mring_pay, min_pay;
#endif
#ifdef ZA_FILE_H_INCLUDED
  ZaFileView fview;
  std::vector<double> fitems[ZA_FILE_SLOTS];
  int64_t fhw[ZA_FH_WORDS];
#endif
  int alias[64];
};

extern "C" {

ZaPort* port_create(double srate, int64_t mem_cap) {
  ZaPort* p = new ZaPort();
  memset(&p->s, 0, sizeof(p->s));
  p->mem.assign((size_t)mem_cap, 0.0);
  p->mt.assign(624, 0u);
  p->s.mem = p->mem.data();
  p->s.mem_stride = 1;
  p->s.mem_cap = mem_cap;
  p->s.mt = p->mt.data();
  p->s.mt_stride = 1;
  p->s.srate = srate;
  p->s.instance_id = 1;
#ifdef ZA_MSG_H_INCLUDED
  p->mring.assign(ZA_MSG_RING, ZaMsg{}); p->moutbox.assign(ZA_MSG_OUTBOX, ZaMsg{}); p->minbox.assign(ZA_MSG_INBOX, ZaMsg{});
  p->mseq = 0; p->mdomain = ZA_MSG_DEFAULT_DOMAIN; p->mlast = 0; p->mout_count = 0; p->min_count = 0;
  memset(p->mch_hash, 0, sizeof p->mch_hash); memset(p->mch_caps, 0, sizeof p->mch_caps);
  memset(p->mch_dropped, 0, sizeof p->mch_dropped); memset(p->mch_flags, 0, sizeof p->mch_flags);
  p->bview = ZaBusView{p->mring.data(), &p->mseq, &p->mdomain, p->mch_hash, p->mch_flags, p->mch_caps, p->mch_dropped, &p->mlast,
                       p->moutbox.data(), &p->mout_count, p->minbox.data(), &p->min_count, 1u, 0u, 1ull, nullptr, nullptr, nullptr,
                       nullptr, nullptr};
  p->mout_pay.assign((size_t)ZA_MSG_OUTBOX * ZA_MSG_PAY, 0.0); p->mring_pay.assign((size_t)ZA_MSG_RING * ZA_MSG_PAY, 0.0);
  p->min_pay.assign((size_t)ZA_MSG_INBOX * ZA_MSG_PAY, 0.0); p->mout_cells = 0; p->mlast_len = 0;
  p->bview.out_pay = p->mout_pay.data(); p->bview.ring_pay = p->mring_pay.data(); p->bview.in_pay = p->min_pay.data();
  p->bview.out_cells = &p->mout_cells; p->bview.last_len = &p->mlast_len;
  p->s.bus = &p->bview;
  p->s.inst_index = 0;
#endif
#ifdef ZA_FILE_H_INCLUDED
  memset(&p->fview, 0, sizeof p->fview);
  memset(p->fhw, 0, sizeof p->fhw);
  p->s.files = &p->fview;
  p->s.fh = p->fhw;
  p->s.fh_stride = 1;
#endif
#ifdef ZA_FFT_MAX
  za_fft_table_init();
  p->fft.assign(2 * ZA_FFT_MAX, 0.0);
  p->s.fft = p->fft.data();
  p->s.fft_stride = 1;
  p->s.fft_cap = 2 * ZA_FFT_MAX;
#endif
#ifdef ZA_GMEM_PAGE_CELLS
  p->gcells.assign(ZA_GMEM_DEFAULT_CELLS, 0ull);
  p->gpseq.assign(ZA_GMEM_DEFAULT_CELLS / ZA_GMEM_PAGE_CELLS, 0ull);
  p->gpwr.assign(ZA_GMEM_DEFAULT_CELLS / ZA_GMEM_PAGE_CELLS, 0ull);
  p->gview = ZaGmemView{p->gcells.data(), p->gpseq.data(), p->gpwr.data(), &p->gseq, ZA_GMEM_DEFAULT_CELLS,
                        ZA_GMEM_DEFAULT_CELLS / ZA_GMEM_PAGE_CELLS};
  p->s.gmem = &p->gview;
  p->s.gmem_attached = ZA_GMEM_AUTOATTACH;
#endif
  for (int i = 0; i < 64; ++i) p->alias[i] = -1;
  return p;
}
void port_destroy(ZaPort* p) { delete p; }
void port_bind_alias(ZaPort* p, int idx0, int var_index) { if (idx0 >= 0 && idx0 < 64) p->alias[idx0] = var_index; }
static void port_sync_alias(ZaPort* p) {
  for (int i = 0; i < 64; ++i) if (p->alias[i] >= 0 && p->alias[i] < ZA_NV) p->s.v[p->alias[i]] = p->s.sl[i];
}
void port_set_sliders(ZaPort* p, const double* v, int n) {
  for (int i = 0; i < n && i < 64; ++i) p->s.sl[i] = v[i];
  port_sync_alias(p);
}
void port_prepare(ZaPort* p) {
  port_sync_alias(p);
  za_section_init(p->s);
  port_sync_alias(p);
  za_section_slider(p->s);
}
void port_run_slider(ZaPort* p) { port_sync_alias(p); za_section_slider(p->s); }

// planar float [nCh][ch_stride]; processes `frames` frames in host blocks of `block`.
void port_process(ZaPort* p, const float* in, float* out, int nCh, int64_t frames, int block, int64_t ch_stride) {
  ZaState<ZA_NV>& s = p->s;
  if (nCh < 0) nCh = 0;
  if (nCh > 64) nCh = 64;
  for (int64_t pos = 0; pos < frames; pos += block) {
    int n = (int)((frames - pos < block) ? frames - pos : block);
    s.samplesblock = (double)n;
    s.block_size = n;
#ifdef ZA_MSG_H_INCLUDED
    za_msg_begin_block(s);
#endif
    za_section_block(s);
    if (s.pend_change | s.pend_automate | s.pend_automate_end) za_section_slider(s);
#if ZA_HAS_SAMPLE
    for (int i = 0; i < n; ++i) {
      for (int ch = 0; ch < nCh; ++ch) s.spl[ch] = (double)in[ch * ch_stride + pos + i];
      za_section_sample(s);
      for (int ch = 0; ch < nCh; ++ch) out[ch * ch_stride + pos + i] = (float)s.spl[ch];
    }
#endif
    s.pend_change = s.pend_automate = s.pend_automate_end = 0;
#ifdef ZA_MSG_H_INCLUDED
    za_msg_flush_all(&p->bview);
#endif
  }
}

int port_nvars(void) { return ZA_NV; }
void port_get_vars(ZaPort* p, double* dst) { memcpy(dst, p->s.v, sizeof(double) * ZA_NV); }
void port_set_var(ZaPort* p, int i, double v) { if (i >= 0 && i < ZA_NV) p->s.v[i] = v; }
void port_get_sliders(ZaPort* p, double* dst) { memcpy(dst, p->s.sl, sizeof(double) * 64); }
void port_get_spl(ZaPort* p, double* dst) { memcpy(dst, p->s.spl, sizeof(double) * 64); }
int64_t port_mem_read(ZaPort* p, int64_t start, int64_t n, double* dst) {
  int64_t k = 0;
  for (; k < n && start + k < p->s.mem_cap; ++k) dst[k] = p->mem[(size_t)(start + k)];
  for (int64_t j = k; j < n; ++j) dst[j] = 0.0;
  return k;
}
int64_t port_mem_write(ZaPort* p, int64_t start, int64_t n, const double* src) {
  int64_t k = 0;
  for (; k < n && start + k < p->s.mem_cap; ++k) p->mem[(size_t)(start + k)] = src[k];
  if (start + k > p->s.mem_high) p->s.mem_high = start + k;
  return k;
}
#ifdef ZA_GMEM_PAGE_CELLS
void port_gmem_read(ZaPort* p, int64_t start, int64_t n, double* dst) { memcpy(dst, p->gcells.data() + start, sizeof(double) * (size_t)n); }
void port_gmem_write(ZaPort* p, int64_t start, int64_t n, const double* src) { memcpy(p->gcells.data() + start, src, sizeof(double) * (size_t)n); }
uint64_t port_gmem_seq(ZaPort* p, int64_t page) { return page < 0 ? p->gseq : p->gpseq[(size_t)page]; }
#endif
#ifdef ZA_POOL_H_INCLUDED
// entries: n x {offset_items, frames, sample_rate, channels} as int64, peaks/rms as float pairs
void port_pool_upload(ZaPort* p, int n, const int64_t* ent4, const float* peak_rms, const float* audio, int64_t items) {
  p->paudio.assign(audio, audio + items);
  p->pent.resize((size_t)n);
  for (int i = 0; i < n; ++i)
    p->pent[(size_t)i] = ZaPoolEntry{(uint64_t)ent4[4 * i], (uint32_t)ent4[4 * i + 1], (uint32_t)ent4[4 * i + 2], (uint32_t)ent4[4 * i + 3],
                                     peak_rms[2 * i], peak_rms[2 * i + 1], 0};
  p->pview = ZaPoolView{p->paudio.data(), (uint64_t)items, p->pent.data(), (uint32_t)n, 1};
  p->s.pool = &p->pview;
}
#endif
#ifdef ZA_FILE_H_INCLUDED
void port_file_slot_set(ZaPort* p, int slot, int channels, double srate, const double* items, int64_t n) {
  if (slot < 0 || slot >= ZA_FILE_SLOTS) return;
  if (!items) { p->fview.slot[slot] = ZaFileSlot{nullptr, 0, 0, 0, 0.0}; return; }
  p->fitems[slot].assign(items, items + n);
  p->fview.slot[slot] = ZaFileSlot{p->fitems[slot].data(), n, channels, 1, srate};
}
#endif
int64_t port_mem_high(ZaPort* p) { return p->s.mem_high; }
int64_t port_mem_need(ZaPort* p) { return p->s.mem_need; }
uint32_t port_err(ZaPort* p) { return p->s.err; }

}  // extern "C"
