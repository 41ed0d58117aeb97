// zart_pool.h -- sample-pool READ builtins for generated section code.
//
// Restates the reference's immutable-generation read path over a float32 arena that the host uploaded to HBM:
//   entry lookup (1-based id) ............................. src/DspJsfxSamplePool.cpp:311-327
//   read: frame = llround(frame), out of range -> 0, channel clamped to [0, ch-1], index =
//         offsetItems + frame*channels + channel (bit-exact integer path) ............ :377-399
//   readInterp: linear between floor(phase) and +1 ....................................... :401-410
//   read2: false (and zeros) unless 0 <= phase <= frames-1; mono entries duplicate L ..... :412-441
//   metadata / export_mem wrappers (llround coercion) ....... src/JSFXJuceProcessor.cpp:5304-5476
// Decode / resample / preview workers and file slots are host work and out of scope; the pool content is whatever
// zab_pool_upload() was given. One pool per engine: its handle is 1.
#pragma once
#define ZA_POOL_H_INCLUDED 1

#include "zart.h"

struct ZaPoolEntry {            // mirrors the fields of DspJsfxSamplePoolEntry the read path uses
  uint64_t offset_items;
  uint32_t frames;
  uint32_t sample_rate;
  uint32_t channels;
  float peak;
  float rms;
  uint32_t pad;
};
struct ZaPoolView {
  const float* audio;           // packed float32, interleaved per entry
  uint64_t audio_items;
  const ZaPoolEntry* entries;
  uint32_t n_entries;
  uint32_t generation;
};

ZA_FN int64_t za_llround(double v) {
  if (!(v == v) || v > 9.2e18 || v < -9.2e18) return 0;
  return v < 0.0 ? -(int64_t)(-v + 0.5) : (int64_t)(v + 0.5);
}
template <class S> ZA_FN const ZaPoolView* za_pool(S& s, double handle) {
  return (s.pool && za_llround(handle) == 1) ? s.pool : nullptr;
}
template <class S> ZA_FN const ZaPoolEntry* za_pool_entry(S& s, double handle, double sampleId) {
  const ZaPoolView* p = za_pool(s, handle);
  if (!p) return nullptr;
  const uint64_t id = (uint64_t)za_llround(sampleId);
  if (id == 0 || id > p->n_entries) return nullptr;
  return &p->entries[id - 1];
}
ZA_FN double za_pool_read_raw(const ZaPoolView* p, const ZaPoolEntry* e, int channel, double frame) {
  if (!e || e->frames == 0 || e->channels == 0) return 0.0;
  if (!(frame == frame) || frame > 1.0e300 || frame < -1.0e300) frame = 0.0;
  const int64_t f = za_llround(frame);
  if (f < 0 || f >= (int64_t)e->frames) return 0.0;
  if (channel < 0) channel = 0;
  if (channel >= (int)e->channels) channel = (int)e->channels - 1;
  const uint64_t idx = e->offset_items + (uint64_t)f * e->channels + (uint64_t)channel;
  return idx < p->audio_items ? (double)p->audio[idx] : 0.0;
}
ZA_FN double za_pool_read_lin(const ZaPoolView* p, const ZaPoolEntry* e, int channel, double phase) {
  if (!(phase == phase) || phase > 1.0e300 || phase < -1.0e300) phase = 0.0;
  const double base = floor(phase), frac = phase - base;
  const double x0 = za_pool_read_raw(p, e, channel, base), x1 = za_pool_read_raw(p, e, channel, base + 1.0);
  return x0 + (x1 - x0) * frac;
}

template <class S> ZA_FN double za_sample_read(S& s, double pool, double id, double ch, double frame) {
  const ZaPoolView* p = za_pool(s, pool);
  return p ? za_pool_read_raw(p, za_pool_entry(s, pool, id), (int)za_llround(ch), frame) : 0.0;
}
template <class S> ZA_FN double za_sample_read_interp(S& s, double pool, double id, double ch, double phase) {
  const ZaPoolView* p = za_pool(s, pool);
  return p ? za_pool_read_lin(p, za_pool_entry(s, pool, id), (int)za_llround(ch), phase) : 0.0;
}
template <class S> ZA_FN double za_pool_read2(S& s, double pool, double id, double phase, double* oL, double* oR, bool interp) {
  const ZaPoolView* p = za_pool(s, pool);
  const ZaPoolEntry* e = p ? za_pool_entry(s, pool, id) : nullptr;
  if (!e || e->frames == 0 || e->channels == 0 || !(phase == phase) || phase < 0.0 || phase > (double)(e->frames - 1)) {
    *oL = 0.0; *oR = 0.0;
    return 0.0;
  }
  const double l = interp ? za_pool_read_lin(p, e, 0, phase) : za_pool_read_raw(p, e, 0, phase);
  const double r = e->channels >= 2 ? (interp ? za_pool_read_lin(p, e, 1, phase) : za_pool_read_raw(p, e, 1, phase)) : l;
  *oL = l; *oR = r;
  return 1.0;
}
template <class S> ZA_FN double za_sample_read2(S& s, double pool, double id, double phase, double* oL, double* oR) {
  return za_pool_read2(s, pool, id, phase, oL, oR, false);
}
template <class S> ZA_FN double za_sample_read2_interp(S& s, double pool, double id, double phase, double* oL, double* oR) {
  return za_pool_read2(s, pool, id, phase, oL, oR, true);
}

template <class S> ZA_FN double za_sample_len(S& s, double pool, double id) { const ZaPoolEntry* e = za_pool_entry(s, pool, id); return e ? (double)e->frames : 0.0; }
template <class S> ZA_FN double za_sample_channels(S& s, double pool, double id) { const ZaPoolEntry* e = za_pool_entry(s, pool, id); return e ? (double)e->channels : 0.0; }
template <class S> ZA_FN double za_sample_srate(S& s, double pool, double id) { const ZaPoolEntry* e = za_pool_entry(s, pool, id); return e ? (double)e->sample_rate : 0.0; }
template <class S> ZA_FN double za_sample_peak(S& s, double pool, double id) { const ZaPoolEntry* e = za_pool_entry(s, pool, id); return e ? (double)e->peak : 0.0; }
template <class S> ZA_FN double za_sample_rms(S& s, double pool, double id) { const ZaPoolEntry* e = za_pool_entry(s, pool, id); return e ? (double)e->rms : 0.0; }
template <class S> ZA_FN double za_sample_get(S& s, double pool, double index) {       // sampleIdAt: ids are 1..n in order
  const ZaPoolView* p = za_pool(s, pool);
  const int64_t i = za_llround(index);
  return (p && i >= 0 && i < (int64_t)p->n_entries) ? (double)(i + 1) : 0.0;
}
template <class S> ZA_FN double za_sample_pool_loaded(S& s, double pool) { const ZaPoolView* p = za_pool(s, pool); return p ? (double)p->n_entries : 0.0; }
template <class S> ZA_FN double za_sample_pool_selected(S& s, double pool) { return za_sample_pool_loaded(s, pool); }
template <class S> ZA_FN double za_sample_pool_failed(S& s, double pool) { (void)s; (void)pool; return 0.0; }
template <class S> ZA_FN double za_sample_pool_generation(S& s, double pool) { const ZaPoolView* p = za_pool(s, pool); return p ? (double)p->generation : 0.0; }
template <class S> ZA_FN double za_sample_pool_state(S& s, double pool) { return za_pool(s, pool) ? 3.0 : 0.0; }   // kSamplePoolReady
template <class S> ZA_FN double za_sample_pool_ram_mb(S& s, double pool) {
  const ZaPoolView* p = za_pool(s, pool);
  return p ? (double)p->audio_items * 4.0 / (1024.0 * 1024.0) : 0.0;
}

// sample_export_mem(pool, id, dst, srcFrame, count): mem[dst+i] = read(id, 0, srcFrame+i); the 2 variant writes L,R pairs.
template <class S> ZA_NOINLINE double za_pool_export_o(S& s, double pool, double id, double dstD, double srcD, double cntD, bool stereo) {
  const ZaPoolView* p = za_pool(s, pool);
  if (!p) return 0.0;
  const int64_t dst = (int32_t)za_llround(dstD), start = (int32_t)za_llround(srcD), count = (int32_t)za_llround(cntD);
  const int stride = stereo ? 2 : 1;
  if (dst < 0 || start < 0 || count <= 0) return 0.0;
  const int64_t needed = dst + count * stride;
  if (needed > s.mem_cap) {                               // the reference grows mem here
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if (needed > s.mem_need) s.mem_need = needed;
    return 0.0;
  }
  za_note_store(s, needed);
  const ZaPoolEntry* e = za_pool_entry(s, pool, id);
  for (int64_t i = 0; i < count; ++i) {
    if (stereo) {
      double l = 0.0, r = 0.0;
      za_pool_read2(s, pool, id, (double)(start + i), &l, &r, false);
      s.mem[(dst + 2 * i) * s.mem_stride] = l;
      s.mem[(dst + 2 * i + 1) * s.mem_stride] = r;
    } else {
      s.mem[(dst + i) * s.mem_stride] = za_pool_read_raw(p, e, 0, (double)(start + i));
    }
  }
  return (double)count;
}
template <class S> ZA_FN double za_pool_export(S& s, double pool, double id, double dstD, double srcD, double cntD, bool stereo) { ZA_OUTCALL(za_pool_export_o(e, pool, id, dstD, srcD, cntD, stereo)); }
template <class S> ZA_FN double za_sample_export_mem(S& s, double pool, double id, double dst, double src, double cnt) { return za_pool_export(s, pool, id, dst, src, cnt, false); }
template <class S> ZA_FN double za_sample_export_mem2(S& s, double pool, double id, double dst, double src, double cnt) { return za_pool_export(s, pool, id, dst, src, cnt, true); }
// host-side setup calls reached from a section: the engine's single pool is always "slot 0 / committed"
template <class S> ZA_FN double za_sample_pool_from_slot(S& s, double, double) { return s.pool ? 1.0 : 0.0; }
template <class S> ZA_FN double za_sample_pool_commit(S& s, double pool) { return za_pool(s, pool) ? 1.0 : 0.0; }
template <class S> ZA_FN double za_sample_pool_set_mode(S& s, double pool, double) { return za_pool(s, pool) ? 1.0 : 0.0; }
template <class S> ZA_FN double za_sample_pool_set_budget_mb(S& s, double pool, double) { return za_pool(s, pool) ? 1.0 : 0.0; }
