// zart_file.h -- file_*() builtins over host-provided file slots, for generated section code.
//
// Restates the reference's runtime file handles (src/JSFXJuceProcessor.cpp:4893-5215): a plugin declares file slots, the
// host decodes whatever the user assigned to a slot into a flat array of doubles ("items": interleaved audio samples),
// and the script reads it through handles:
//   file_open(slot)        -1 if the slot is unassigned, else a 1-based handle from a LIFO free list, else a NEW one: the
//                          reference's handle table grows without bound (:4976-4981), and leaves lean on that -- TextureXY
//                          opens slot 0 in every @block and never closes it. Handle numbers therefore grow the same way
//                          here; the state of the ZA_FILE_HANDLES most recent ones is kept (a window: handle k lives in cell
//                          k % 8 under its tag), and USING a handle that has left the window raises ZA_ERR_UNSUPPORTED
//                          instead of answering differently from the reference ................................ :4948-4990
//   file_close / file_rewind / file_seek(offset: trunc(x + 1e-5), clamped to [0, items]) .................... :5002-5054
//   file_avail             remaining items ......................................................... :5056-5072
//   file_riff(h, nch, sr)  1 and (channels, sample rate) for audio data, else 0 and zeros ............. :5087-5105
//   file_var(h, v)         next item (0 past the end) .............................................. :5107-5132
//   file_mem(h, dst, len)  copy min(len, avail) items to mem[dst..], advance; returns the count .... :5134-5172
//   file_text 0, file_multi_count 1, file_multi_select(h, 0) (one file per slot here)
// The host keeps decoding (zab_file_slot_set uploads a slot's items; all instances of an engine see the same slots, each
// with its own handles and cursors). Text files and multi-file sets are host features that are not mirrored.
#pragma once
#define ZA_FILE_H_INCLUDED 1

#include "zart.h"

#define ZA_FILE_SLOTS 16
#define ZA_FILE_HANDLES 8
// per-instance handle state, int64 words: cell c = k % 8 of handle k: [c] = slot + 1 (0: closed), [8 + c] = cursor,
// [26 + c] = k (the tag); [16] = handles ever created, [17] = free count, [18 + j] = free stack (handle indices)
#define ZA_FH_WORDS 34

struct ZaFileSlot {
  const double* items;
  int64_t n_items;
  int32_t channels;
  int32_t assigned;
  double srate;
};
struct ZaFileView { ZaFileSlot slot[ZA_FILE_SLOTS]; };

#define ZA_FH(k) s.fh[(int64_t)(k) * s.fh_stride]

template <class S> ZA_FN int za_file_h(S& s, double handle) {             // getRuntimeFileHandle: index or -1
  if (!s.fh) return -1;
  const int64_t hid = za_f2i64(handle + 1.0e-5);
  if (hid <= 0 || hid > ZA_FH(16)) return -1;
  const int k = (int)((hid - 1) % ZA_FILE_HANDLES);
  if (ZA_FH(26 + k) != hid - 1) { s.err |= ZA_ERR_UNSUPPORTED; return -1; }   // left the window (still valid in the reference)
  return ZA_FH(k) > 0 ? k : -1;
}
template <class S> ZA_FN const ZaFileSlot* za_file_data(S& s, int k) {     // the handle's slot if it holds data
  const int64_t sl = ZA_FH(k) - 1;
  if (!s.files || sl < 0 || sl >= ZA_FILE_SLOTS) return nullptr;
  const ZaFileSlot* f = &s.files->slot[sl];
  return (f->assigned && f->items) ? f : nullptr;
}
template <class S> ZA_NOINLINE double za_file_open_o(S& s, double indexOrSlot, double mode) {
  (void)mode;
  if (!s.files || !s.fh) return -1.0;
  const int64_t sl = za_f2i64(indexOrSlot + 1.0e-5);
  if (sl < 0 || sl >= ZA_FILE_SLOTS || !s.files->slot[sl].assigned) return -1.0;
  int64_t hx;                                 // 0-based handle index
  if (ZA_FH(17) > 0) {                       // LIFO free list (a closed handle was inside the window when it closed, and new
    const int64_t nf = ZA_FH(17) - 1;        //  indices are only made while this list is empty: its cell is still its own)
    hx = ZA_FH(18 + nf);
    ZA_FH(17) = nf;
  } else {
    hx = ZA_FH(16);
    if (hx >= (int64_t)1 << 52) { s.err |= ZA_ERR_UNSUPPORTED; return -1.0; }
    ZA_FH(16) = hx + 1;
  }
  const int k = (int)(hx % ZA_FILE_HANDLES);
  ZA_FH(k) = sl + 1;
  ZA_FH(8 + k) = 0;
  ZA_FH(26 + k) = hx;
  return (double)(hx + 1);
}
template <class S> ZA_FN double za_file_open(S& s, double indexOrSlot, double mode) { ZA_OUTCALL(za_file_open_o(e, indexOrSlot, mode)); }
template <class S> ZA_FN double za_file_open_multi(S& s, double a, double b) { return za_file_open(s, a, b); }
template <class S> ZA_NOINLINE double za_file_close_o(S& s, double handle) {
  const int k = za_file_h(s, handle);
  if (k < 0) return 0.0;
  ZA_FH(k) = 0; ZA_FH(8 + k) = 0;
  const int64_t nf = ZA_FH(17);              // (at most ZA_FILE_HANDLES handles are inside the window, so are closable)
  ZA_FH(18 + nf) = ZA_FH(26 + k);
  ZA_FH(17) = nf + 1;
  return 0.0;
}
template <class S> ZA_FN double za_file_close(S& s, double handle) { ZA_OUTCALL(za_file_close_o(e, handle)); }
template <class S> ZA_FN double za_file_rewind(S& s, double handle) {
  const int k = za_file_h(s, handle);
  if (k >= 0) ZA_FH(8 + k) = 0;
  return 0.0;
}
template <class S> ZA_FN double za_file_seek(S& s, double handle, double offset) {
  const int k = za_file_h(s, handle);
  if (k < 0) return 0.0;
  const ZaFileSlot* f = za_file_data(s, k);
  if (!f) { ZA_FH(8 + k) = 0; return 0.0; }
  int64_t off = za_f2i64(offset + 1.0e-5);
  if (off < 0) off = 0;
  if (off > f->n_items) off = f->n_items;
  ZA_FH(8 + k) = off;
  return (double)off;
}
template <class S> ZA_FN double za_file_avail(S& s, double handle) {
  const int k = za_file_h(s, handle);
  if (k < 0) return 0.0;
  const ZaFileSlot* f = za_file_data(s, k);
  if (!f) return 0.0;
  const int64_t rem = f->n_items - ZA_FH(8 + k);
  return rem > 0 ? (double)rem : 0.0;
}
template <class S> ZA_FN double za_file_text(S& s, double handle) { (void)s; (void)handle; return 0.0; }
template <class S> ZA_FN double za_file_riff(S& s, double handle, double* nch, double* sr) {
  const int k = za_file_h(s, handle);
  if (k < 0) return 0.0;                                   // (outputs untouched, as in the reference's early return)
  const ZaFileSlot* f = za_file_data(s, k);
  if (!f || f->channels <= 0) { *nch = 0.0; *sr = 0.0; return 0.0; }
  *nch = (double)f->channels; *sr = f->srate;
  return 1.0;
}
template <class S> ZA_FN double za_file_var(S& s, double handle, double* out) {
  const int k = za_file_h(s, handle);
  if (k < 0) { *out = 0.0; return 0.0; }
  const ZaFileSlot* f = za_file_data(s, k);
  const int64_t c = ZA_FH(8 + k);
  if (!f || c < 0 || c >= f->n_items) { *out = 0.0; return 0.0; }
  const double v = f->items[c];
  ZA_FH(8 + k) = c + 1;
  *out = v;
  return v;
}
template <class S> ZA_NOINLINE double za_file_mem_o(S& s, double handle, double destIndex, double length) {
  const int k = za_file_h(s, handle);
  if (k < 0) return 0.0;
  const ZaFileSlot* f = za_file_data(s, k);
  if (!f) return 0.0;
  int64_t dst = za_f2i64(destIndex + 1.0e-5), len = za_f2i64(length + 1.0e-5);
  if (dst < 0) dst = 0;
  if (len <= 0) return 0.0;
  const int64_t cur = ZA_FH(8 + k), avail = f->n_items - cur;
  if (avail <= 0) return 0.0;
  const int64_t n = len < avail ? len : avail;
  if (dst + n > s.mem_cap) {                               // the reference grows mem here (jsfx_ensure_mem)
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if (dst + n > s.mem_need) s.mem_need = dst + n;
    return 0.0;
  }
  for (int64_t i = 0; i < n; ++i) s.mem[(dst + i) * s.mem_stride] = f->items[cur + i];
  za_note_store(s, dst + n);
  ZA_FH(8 + k) = cur + n;
  return (double)n;
}
template <class S> ZA_FN double za_file_mem(S& s, double handle, double destIndex, double length) { ZA_OUTCALL(za_file_mem_o(e, handle, destIndex, length)); }
template <class S> ZA_FN double za_file_multi_count(S& s, double handle) {
  const int k = za_file_h(s, handle);
  return (k >= 0 && za_file_data(s, k)) ? 1.0 : 0.0;
}
template <class S> ZA_FN double za_file_multi_select(S& s, double handle, double index) {
  const int k = za_file_h(s, handle);
  if (k < 0 || !za_file_data(s, k)) return 0.0;
  if (floor(index + 1.0e-5) != 0.0) return 0.0;
  ZA_FH(8 + k) = 0;
  return 1.0;
}
#undef ZA_FH
