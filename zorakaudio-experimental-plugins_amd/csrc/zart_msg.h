// zart_msg.h -- block-resolved message bus between the instances of one engine, for generated section code.
//
// Restates the observable behaviour of the reference's per-instance runtime + shared-memory bus for scalar messages:
//   comm_join(domain) ............ leaves the old domain (inbox and read cursor reset), joins the new one; 0 selects the default
//                                  domain .................................................. src/DspJsfxRuntime.cpp:234-247
//   msg_subscribe / unsubscribe / advertise(channel, caps): per-instance channel table (24 entries) ... :291-313,
//                                  src/DspJsfxMessageBus.cpp:19-21
//   msg_send / msg_sendto ........ appended to the instance's outbox (1024 per block, else dropped[channel]++) ... :315-372
//   end of block ................. outbox flushed to the domain's ring (4096 slots) in order; a broadcast with no OTHER active
//                                  instance of the domain subscribed to its channel (a direct message with no active target)
//                                  is dropped and counted ............................ DspJsfxMessageBus.cpp:529-607
//   start of block ............... ring entries newer than the instance's read cursor that target it (direct: target == id;
//                                  broadcast: source != id and channel subscribed) are moved to its ready inbox; a receiver
//                                  that lagged past the ring counts one drop per subscription ............. :609-677
//   msg_recv / avail / kind / clear / dropped / length ........................ DspJsfxRuntime.cpp:409-478
//   msg_peer_count / peer_id / peer_caps / peer_alive over the domain's instances (self included) ... MessageBus.cpp:713-790
// The reference delivers what a sender flushed at the end of ITS block to whoever starts a block afterwards; instances run
// on unrelated threads, so the order inside one host block is unspecified there. Here every instance of the engine runs
// host block k, then all outboxes are flushed in instance order, then block k + 1 starts: a message sent in block k is
// received in block k + 1.
//   msg_send_buf / msg_sendto_buf(…, tag, srcBase, len): the payload is copied out of mem[] when the message is queued; an
//                                  outbox holds at most 65536 payload cells per block (else dropped[channel]++); a payload of
//                                  more than 64 cells does not fit a ring slot and is dropped (and counted) at the flush
//                                  ............. src/DspJsfxRuntimeBuiltins.cpp:289-338, DspJsfxRuntime.cpp:318-392,442-464,
//                                  DspJsfxMessageBus.cpp:22,205-222,558-562
//   msg_recv_buf(chan, src, tag, dstBase, maxLen): only if the channel's oldest message is a buffer; copies min(maxLen, n) cells,
//                                  returns n, or -n when maxLen < n; msg_length() = n of the last buffer received (0 after a
//                                  scalar); msg_recv only takes a scalar at the front; msg_kind: 1 scalar, 2 buffer
// The string-valued peer queries (msg_peer_name / uid) stay host-only.
#pragma once
#define ZA_MSG_H_INCLUDED 1

#include "zart.h"

#define ZA_MSG_RING 4096
#define ZA_MSG_CHANNELS 24
#define ZA_MSG_OUTBOX 1024
#define ZA_MSG_INBOX 1024
#define ZA_MSG_MAX_INSTANCES 256
#define ZA_MSG_PAY 64                  /* payload cells a ring slot carries (kIpcMaxPayloadCells) */
#define ZA_MSG_OUT_CELLS 65536         /* payload cells an outbox may hold per block (kMaxOutboxBufferCells) */
#define ZA_MSG_DEFAULT_DOMAIN 0x9ae16a3b2f90404full

struct ZaMsg {
  uint64_t seq, chan, src, target;
  double tag, a, b, c, d;
  uint32_t kind;       // 0 none / consumed, 1 scalar, 2 buffer
  uint32_t pad;        // outbox: 1 = direct
  uint32_t blen;       // buffer: payload cells as sent (what msg_length reports)
  uint32_t pad2;
};
struct ZaBusView {
  ZaMsg* ring;                 // [ZA_MSG_RING]
  uint64_t* global_seq;        // [1]
  uint64_t* domain;            // [n]
  uint64_t* ch_hash;           // [n][ZA_MSG_CHANNELS]
  uint32_t* ch_flags;          // [n][ZA_MSG_CHANNELS] bit 0 subscribed, bit 1 advertised, bit 2 slot in use
  uint64_t* ch_caps;           // [n][ZA_MSG_CHANNELS]
  uint64_t* ch_dropped;        // [n][ZA_MSG_CHANNELS]
  uint64_t* last_read;         // [n]
  ZaMsg* outbox;               // [n][ZA_MSG_OUTBOX]
  uint32_t* out_count;         // [n]
  ZaMsg* inbox;                // [n][ZA_MSG_INBOX]  (kind 0 = consumed)
  uint32_t* in_count;          // [n]
  uint32_t n_inst;
  uint32_t pad;
  uint64_t first_id;
  // buffer messages (null unless the leaf uses msg_*_buf): payload rows parallel to outbox / ring / inbox entries
  double* out_pay;             // [n][ZA_MSG_OUTBOX][ZA_MSG_PAY]
  double* ring_pay;            // [ZA_MSG_RING][ZA_MSG_PAY]
  double* in_pay;              // [n][ZA_MSG_INBOX][ZA_MSG_PAY]
  uint32_t* out_cells;         // [n] payload cells queued in this block
  uint32_t* last_len;          // [n] msg_length()
};

// channel / domain identity of a string handle: the handle value itself (one script per engine: equal names are equal handles)
ZA_FN uint64_t za_msg_key(double h) { const int64_t v = za_f2i64(h + (h < 0 ? -0.5 : 0.5)); return (uint64_t)v; }

template <class S> ZA_FN int za_msg_slot(S& s, uint64_t chan, bool create) {
  const ZaBusView* B = s.bus;
  const int64_t base = (int64_t)s.inst_index * ZA_MSG_CHANNELS;
  int free_k = -1;
  for (int k = 0; k < ZA_MSG_CHANNELS; ++k) {
    const uint32_t f = B->ch_flags[base + k];
    if (!(f & 4u)) { free_k = k; break; }               // slots are handed out first-free and never released: a prefix
    if (B->ch_hash[base + k] == chan) return k;
  }
  if (!create || free_k < 0) return -1;
  B->ch_hash[base + free_k] = chan;
  B->ch_flags[base + free_k] = 4u;
  B->ch_caps[base + free_k] = 0;
  B->ch_dropped[base + free_k] = 0;
  return free_k;
}
template <class S> ZA_FN void za_msg_drop(S& s, uint64_t chan) {
  const int k = za_msg_slot(s, chan, true);
  if (k >= 0) s.bus->ch_dropped[(int64_t)s.inst_index * ZA_MSG_CHANNELS + k] += 1u;
}

template <class S> ZA_NOINLINE double za_comm_join_o(S& s, double domainH) {
  if (!s.bus) return 0.0;
  uint64_t d = za_msg_key(domainH);
  if (d == 0) d = ZA_MSG_DEFAULT_DOMAIN;
  if (s.bus->domain[s.inst_index] != d) {            // joinDomain: inboxes cleared, read cursor reset
    s.bus->in_count[s.inst_index] = 0;
    s.bus->last_read[s.inst_index] = 0;
  }
  s.bus->domain[s.inst_index] = d;
  return 1.0;
}
template <class S> ZA_FN double za_comm_join(S& s, double domainH) { ZA_OUTCALL(za_comm_join_o(e, domainH)); }
template <class S> ZA_NOINLINE double za_msg_subscribe_o(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const int k = za_msg_slot(s, za_msg_key(chanH), true);
  if (k >= 0) s.bus->ch_flags[(int64_t)s.inst_index * ZA_MSG_CHANNELS + k] |= 1u;
  return 1.0;
}
template <class S> ZA_FN double za_msg_subscribe(S& s, double chanH) { ZA_OUTCALL(za_msg_subscribe_o(e, chanH)); }
template <class S> ZA_NOINLINE double za_msg_unsubscribe_o(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const int k = za_msg_slot(s, za_msg_key(chanH), false);
  if (k >= 0) s.bus->ch_flags[(int64_t)s.inst_index * ZA_MSG_CHANNELS + k] &= ~1u;
  return 1.0;
}
template <class S> ZA_FN double za_msg_unsubscribe(S& s, double chanH) { ZA_OUTCALL(za_msg_unsubscribe_o(e, chanH)); }
template <class S> ZA_NOINLINE double za_msg_advertise_o(S& s, double chanH, double capsD) {
  if (!s.bus) return 0.0;
  const int64_t ci = za_f2i64(capsD + (capsD < 0 ? -0.5 : 0.5));
  const uint64_t caps = ci < 0 ? 0 : (uint64_t)ci;
  const int k = za_msg_slot(s, za_msg_key(chanH), caps != 0);
  if (k >= 0) {
    const int64_t at = (int64_t)s.inst_index * ZA_MSG_CHANNELS + k;
    s.bus->ch_caps[at] = caps;
    if (caps) s.bus->ch_flags[at] |= 2u; else s.bus->ch_flags[at] &= ~2u;
  }
  return 1.0;
}
template <class S> ZA_FN double za_msg_advertise(S& s, double chanH, double capsD) { ZA_OUTCALL(za_msg_advertise_o(e, chanH, capsD)); }
template <class S> ZA_NOINLINE double za_msg_queue_o(S& s, uint64_t target, bool direct, double chanH, double tag, double a, double b,
                                                  double c, double d, int64_t buf_base = -1, int64_t buf_len = 0) {
  if (!s.bus) return 0.0;
  const uint64_t chan = za_msg_key(chanH);
  const uint32_t n = s.bus->out_count[s.inst_index];
  const bool is_buf = buf_base >= 0;
  if (n >= ZA_MSG_OUTBOX || (is_buf && (int64_t)s.bus->out_cells[s.inst_index] + buf_len > ZA_MSG_OUT_CELLS)) { za_msg_drop(s, chan); return 0.0; }
  ZaMsg& m = s.bus->outbox[(int64_t)s.inst_index * ZA_MSG_OUTBOX + n];
  m.seq = 0; m.chan = chan; m.src = s.instance_id; m.target = direct ? target : 0;
  m.tag = tag; m.a = a; m.b = b; m.c = c; m.d = d; m.kind = is_buf ? 2u : 1u; m.pad = direct ? 1u : 0u;
  m.blen = is_buf ? (uint32_t)buf_len : 0u; m.pad2 = 0;
  if (is_buf) {
    double* row = s.bus->out_pay + ((int64_t)s.inst_index * ZA_MSG_OUTBOX + n) * ZA_MSG_PAY;
    for (int64_t i = 0; i < buf_len && i < ZA_MSG_PAY; ++i) row[i] = s.mem[(buf_base + i) * s.mem_stride];   // (longer ones die at the flush)
    s.bus->out_cells[s.inst_index] += (uint32_t)buf_len;
  }
  s.bus->out_count[s.inst_index] = n + 1;
  return 1.0;
}
template <class S> ZA_FN double za_msg_queue(S& s, uint64_t target, bool direct, double chanH, double tag, double a, double b, double c, double d, int64_t buf_base = -1, int64_t buf_len = 0) { ZA_OUTCALL(za_msg_queue_o(e, target, direct, chanH, tag, a, b, c, d, buf_base, buf_len)); }
// std::llround into int with saturation, non-finite -> 0 (toInt, src/DspJsfxRuntimeBuiltins.cpp:35-44)
ZA_FN int64_t za_msg_toint(double v) {
  if (!(v == v) || v - v != 0.0) return 0;
  if (v <= -2147483648.0) return -2147483648ll;
  if (v >= 2147483647.0) return 2147483647ll;
  return za_f2i64(v + (v < 0 ? -0.5 : 0.5));
}
template <class S> ZA_FN double za_msg_send_buf(S& s, double chanH, double tag, double srcBase, double len) {
  if (!s.bus || !s.bus->out_pay) return 0.0;
  const int64_t base = za_msg_toint(srcBase), n = za_msg_toint(len);
  if (base < 0 || n <= 0 || base + n > s.mem_cap) return 0.0;
  return za_msg_queue(s, 0, false, chanH, tag, 0.0, 0.0, 0.0, 0.0, base, n);
}
template <class S> ZA_FN double za_msg_sendto_buf(S& s, double targetD, double chanH, double tag, double srcBase, double len) {
  if (!s.bus || !s.bus->out_pay) return 0.0;
  const int64_t base = za_msg_toint(srcBase), n = za_msg_toint(len);
  if (base < 0 || n <= 0 || base + n > s.mem_cap) return 0.0;
  const int64_t t = za_f2i64(targetD + 0.5);
  return za_msg_queue(s, t < 0 ? 0 : (uint64_t)t, true, chanH, tag, 0.0, 0.0, 0.0, 0.0, base, n);
}
template <class S> ZA_FN double za_msg_send(S& s, double chanH, double tag, double a, double b, double c, double d) {
  return za_msg_queue(s, 0, false, chanH, tag, a, b, c, d);
}
template <class S> ZA_FN double za_msg_sendto(S& s, double targetD, double chanH, double tag, double a, double b, double c, double d) {
  const int64_t t = za_f2i64(targetD + 0.5);
  return za_msg_queue(s, t < 0 ? 0 : (uint64_t)t, true, chanH, tag, a, b, c, d);
}
// first unconsumed inbox entry of a channel, or -1
template <class S> ZA_FN int64_t za_msg_front(S& s, uint64_t chan) {
  const int64_t base = (int64_t)s.inst_index * ZA_MSG_INBOX;
  const uint32_t n = s.bus->in_count[s.inst_index];
  for (uint32_t i = 0; i < n; ++i)
    if (s.bus->inbox[base + i].kind != 0 && s.bus->inbox[base + i].chan == chan) return base + i;
  return -1;
}
template <class S> ZA_NOINLINE double za_msg_recv_o(S& s, double chanH, double* src, double* tag, double* a, double* b, double* c, double* d) {
  if (!s.bus) return 0.0;
  const int64_t at = za_msg_front(s, za_msg_key(chanH));
  if (at < 0 || s.bus->inbox[at].kind != 1) return 0.0;      // (a buffer at the front waits for msg_recv_buf)
  ZaMsg& m = s.bus->inbox[at];
  *src = (double)m.src; *tag = m.tag; *a = m.a; *b = m.b; *c = m.c; *d = m.d;
  m.kind = 0;
  if (s.bus->last_len) s.bus->last_len[s.inst_index] = 0;
  return 1.0;
}
template <class S> ZA_FN double za_msg_recv(S& s, double chanH, double* src, double* tag, double* a, double* b, double* c, double* d) { typename S::Env e = s; double o1_ = *src; double o2_ = *tag; double o3_ = *a; double o4_ = *b; double o5_ = *c; double o6_ = *d; const double r_ = za_msg_recv_o(e, chanH, &o1_, &o2_, &o3_, &o4_, &o5_, &o6_); *src = o1_; *tag = o2_; *a = o3_; *b = o4_; *c = o5_; *d = o6_; static_cast<typename S::Env&>(s) = e; return r_; }
template <class S> ZA_NOINLINE double za_msg_recv_buf_o(S& s, double chanH, double* src, double* tag, double dstBase, double maxLen) {
  if (!s.bus || !s.bus->in_pay) return 0.0;
  int64_t cap = za_msg_toint(maxLen);
  if (cap <= 0) return 0.0;
  const int64_t dst = za_msg_toint(dstBase);
  if (dst < 0) return 0.0;
  if (dst + cap > s.mem_cap) {            // the reference grows mem here; a fixed arena reports the overflow
    s.err |= ZA_ERR_MEM_OVERFLOW;
    if (dst + cap > s.mem_need) s.mem_need = dst + cap;
    return 0.0;
  }
  const int64_t at = za_msg_front(s, za_msg_key(chanH));
  if (at < 0 || s.bus->inbox[at].kind != 2) return 0.0;
  ZaMsg& m = s.bus->inbox[at];
  *src = (double)m.src; *tag = m.tag;
  const int64_t n = m.blen, cp = cap < n ? cap : n;
  const double* row = s.bus->in_pay + at * ZA_MSG_PAY;
  for (int64_t i = 0; i < cp; ++i) s.mem[(dst + i) * s.mem_stride] = row[i];
  if (cp > 0) za_note_store(s, dst + cp);
  s.bus->last_len[s.inst_index] = (uint32_t)n;
  m.kind = 0;
  return cap >= n ? (double)n : -(double)n;
}
template <class S> ZA_FN double za_msg_recv_buf(S& s, double chanH, double* src, double* tag, double dstBase, double maxLen) { typename S::Env e = s; double o1_ = *src; double o2_ = *tag; const double r_ = za_msg_recv_buf_o(e, chanH, &o1_, &o2_, dstBase, maxLen); *src = o1_; *tag = o2_; static_cast<typename S::Env&>(s) = e; return r_; }
template <class S> ZA_NOINLINE double za_msg_avail_o(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const uint64_t chan = za_msg_key(chanH);
  const int64_t base = (int64_t)s.inst_index * ZA_MSG_INBOX;
  int cnt = 0;
  for (uint32_t i = 0; i < s.bus->in_count[s.inst_index]; ++i) cnt += (s.bus->inbox[base + i].kind != 0 && s.bus->inbox[base + i].chan == chan);
  return (double)cnt;
}
template <class S> ZA_FN double za_msg_avail(S& s, double chanH) { ZA_OUTCALL(za_msg_avail_o(e, chanH)); }
template <class S> ZA_FN double za_msg_kind(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const int64_t at = za_msg_front(s, za_msg_key(chanH));
  return at >= 0 ? (double)s.bus->inbox[at].kind : 0.0;
}
template <class S> ZA_NOINLINE double za_msg_clear_o(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const uint64_t chan = za_msg_key(chanH);
  const int64_t base = (int64_t)s.inst_index * ZA_MSG_INBOX;
  int cnt = 0;
  for (uint32_t i = 0; i < s.bus->in_count[s.inst_index]; ++i)
    if (s.bus->inbox[base + i].kind != 0 && s.bus->inbox[base + i].chan == chan) { s.bus->inbox[base + i].kind = 0; ++cnt; }
  return (double)cnt;
}
template <class S> ZA_FN double za_msg_clear(S& s, double chanH) { ZA_OUTCALL(za_msg_clear_o(e, chanH)); }
template <class S> ZA_FN double za_msg_dropped(S& s, double chanH) {
  if (!s.bus) return 0.0;
  const int k = za_msg_slot(s, za_msg_key(chanH), false);
  return k >= 0 ? (double)s.bus->ch_dropped[(int64_t)s.inst_index * ZA_MSG_CHANNELS + k] : 0.0;
}
template <class S> ZA_FN double za_msg_length(S& s) { return (s.bus && s.bus->last_len) ? (double)s.bus->last_len[s.inst_index] : 0.0; }

ZA_FN bool za_msg_matches(const ZaBusView* B, uint32_t j, uint64_t chan, int role) {   // channelMatches
  const bool wantSub = role == 1 || role == 3 || role <= 0, wantPub = role == 2 || role == 3 || role <= 0;
  const int64_t base = (int64_t)j * ZA_MSG_CHANNELS;
  for (int k = 0; k < ZA_MSG_CHANNELS; ++k) {
    const uint32_t f = B->ch_flags[base + k];
    if (!(f & 4u)) break;                                // used slots are a prefix (za_msg_slot)
    if (B->ch_hash[base + k] == chan) return (wantSub && (f & 1u)) || (wantPub && (f & 2u));
  }
  return false;
}
template <class S> ZA_NOINLINE double za_msg_peer_count_o(S& s, double chanH, double roleD) {
  if (!s.bus) return 0.0;
  const uint64_t chan = za_msg_key(chanH), dom = s.bus->domain[s.inst_index];
  const int role = (int)za_f2i64(roleD + (roleD < 0 ? -0.5 : 0.5));
  int cnt = 0;
  for (uint32_t j = 0; j < s.bus->n_inst; ++j) cnt += (s.bus->domain[j] == dom && za_msg_matches(s.bus, j, chan, role));
  return (double)cnt;
}
template <class S> ZA_FN double za_msg_peer_count(S& s, double chanH, double roleD) { ZA_OUTCALL(za_msg_peer_count_o(e, chanH, roleD)); }
template <class S> ZA_NOINLINE double za_msg_peer_id_o(S& s, double chanH, double roleD, double indexD) {
  if (!s.bus) return 0.0;
  const uint64_t chan = za_msg_key(chanH), dom = s.bus->domain[s.inst_index];
  const int role = (int)za_f2i64(roleD + (roleD < 0 ? -0.5 : 0.5));
  const int64_t want = za_f2i64(indexD + (indexD < 0 ? -0.5 : 0.5));
  if (want < 0) return 0.0;
  int64_t cnt = 0;
  for (uint32_t j = 0; j < s.bus->n_inst; ++j)
    if (s.bus->domain[j] == dom && za_msg_matches(s.bus, j, chan, role)) { if (cnt == want) return (double)(s.bus->first_id + j); ++cnt; }
  return 0.0;
}
template <class S> ZA_FN double za_msg_peer_id(S& s, double chanH, double roleD, double indexD) { ZA_OUTCALL(za_msg_peer_id_o(e, chanH, roleD, indexD)); }
template <class S> ZA_FN int64_t za_msg_peer_index(S& s, double idD) {
  const int64_t id = za_f2i64(idD + 0.5) - (int64_t)s.bus->first_id;
  return (id >= 0 && id < (int64_t)s.bus->n_inst) ? id : -1;
}
template <class S> ZA_NOINLINE double za_msg_peer_caps_o(S& s, double idD) {     // merged caps of the peer's advertisements
  if (!s.bus) return 0.0;
  const int64_t j = za_msg_peer_index(s, idD);
  if (j < 0) return 0.0;
  uint64_t caps = 0;
  for (int k = 0; k < ZA_MSG_CHANNELS; ++k)
    if (s.bus->ch_flags[j * ZA_MSG_CHANNELS + k] & 2u) caps |= s.bus->ch_caps[j * ZA_MSG_CHANNELS + k];
  return (double)caps;
}
template <class S> ZA_FN double za_msg_peer_caps(S& s, double idD) { ZA_OUTCALL(za_msg_peer_caps_o(e, idD)); }
template <class S> ZA_FN double za_msg_peer_alive(S& s, double idD) { return (s.bus && za_msg_peer_index(s, idD) >= 0) ? 1.0 : 0.0; }

// beginBlock: collect what the ring holds for this instance (DspJsfxMessageBus::collectInbox)
template <class S> ZA_NOINLINE void za_msg_begin_block_o(S& s) {
  const ZaBusView* B = s.bus;
  if (!B) return;
  const uint32_t me = s.inst_index;
  const int64_t ibase = (int64_t)me * ZA_MSG_INBOX;
  uint32_t n = 0;                                        // compact the ready inbox (drop consumed entries, keep order)
  for (uint32_t i = 0; i < B->in_count[me]; ++i)
    if (B->inbox[ibase + i].kind != 0) {
      if (n != i) {
        B->inbox[ibase + n] = B->inbox[ibase + i];
        if (B->in_pay && B->inbox[ibase + i].kind == 2)
          for (int c = 0; c < ZA_MSG_PAY; ++c) B->in_pay[(ibase + n) * ZA_MSG_PAY + c] = B->in_pay[(ibase + i) * ZA_MSG_PAY + c];
      }
      ++n;
    }
  const uint64_t newest = *B->global_seq, last = B->last_read[me];
  if (newest > last) {
    uint64_t first = last + 1;
    if (newest > ZA_MSG_RING && first + ZA_MSG_RING <= newest) {     // lagged past the ring window
      for (int k = 0; k < ZA_MSG_CHANNELS; ++k)
        if ((B->ch_flags[(int64_t)me * ZA_MSG_CHANNELS + k] & 5u) == 5u) B->ch_dropped[(int64_t)me * ZA_MSG_CHANNELS + k] += 1u;
      first = newest - ZA_MSG_RING + 1;
    }
    for (uint64_t q = first; q <= newest; ++q) {
      const ZaMsg& m = B->ring[q % ZA_MSG_RING];
      if (m.seq != q) continue;
      bool mine;
      if (m.target != 0) mine = m.target == s.instance_id;
      else mine = m.src != s.instance_id && za_msg_matches(B, me, m.chan, 1);
      if (!mine) continue;
      if (n < ZA_MSG_INBOX) {
        B->inbox[ibase + n] = m;
        if (B->in_pay && m.kind == 2)
          for (int c = 0; c < ZA_MSG_PAY; ++c) B->in_pay[(ibase + n) * ZA_MSG_PAY + c] = B->ring_pay[(int64_t)(q % ZA_MSG_RING) * ZA_MSG_PAY + c];
        ++n;
      } else za_msg_drop(s, m.chan);
    }
    B->last_read[me] = newest;
  }
  B->in_count[me] = n;
}
template <class S> ZA_FN void za_msg_begin_block(S& s) { ZA_OUTCALL_VOID(za_msg_begin_block_o(e)); }

// endBlock of every instance, in instance order: DspJsfxMessageBus::flushOutbox. One thread on the CPU; one wavefront on
// the device, where the lanes share the search for a subscriber (the only part that grows with the instance count) and
// lane 0 does the bookkeeping, so the ring order stays the serial one.
#if defined(__HIPCC__)
#define ZA_MSG_LANE() ((uint32_t)threadIdx.x)
#define ZA_MSG_LANES 64u
#define ZA_MSG_ANY(p) (__ballot(p) != 0ull)
#define ZA_MSG_CONVERGE() __builtin_amdgcn_wave_barrier()
#else
#define ZA_MSG_CONVERGE() ((void)0)
#define ZA_MSG_LANE() 0u
#define ZA_MSG_LANES 1u
#define ZA_MSG_ANY(p) (p)
#endif
ZA_FN void za_msg_flush_all(const ZaBusView* B) {
  const uint32_t lane = ZA_MSG_LANE();
  for (uint32_t i = 0; i < B->n_inst; ++i) {
    const uint32_t cnt = B->out_count[i];
    const uint64_t dom = B->domain[i], me = B->first_id + i;
    for (uint32_t k = 0; k < cnt; ++k) {
      const ZaMsg& in = B->outbox[(int64_t)i * ZA_MSG_OUTBOX + k];
      bool has_target = false;
      const bool oversize = in.kind == 2 && in.blen > ZA_MSG_PAY;      // does not fit a ring slot: dropped, whoever listens
      if (oversize) {
      } else if (in.pad) {                               // direct
        const int64_t j = (int64_t)in.target - (int64_t)B->first_id;
        has_target = j >= 0 && j < (int64_t)B->n_inst && B->domain[j] == dom;
      } else {
        for (uint32_t j0 = 0; j0 < B->n_inst && !has_target; j0 += ZA_MSG_LANES) {
          const uint32_t j = j0 + lane;
          const bool mine = j < B->n_inst && j != i && B->domain[j] == dom && za_msg_matches(B, j, in.chan, 1);
          has_target = ZA_MSG_ANY(mine);
        }
      }
      // Lane 0 alone does the bookkeeping, inside an `if` (no `continue`): every lane reaches the next trip's ballot together,
      // and the wave barrier keeps the compiler from scheduling a later ballot across the divergent region.
      if (lane == 0) {
        if (!has_target) {                               // dropped[channel]++ on the sender
          const int64_t base = (int64_t)i * ZA_MSG_CHANNELS;
          int slot = -1, free_k = -1;
          for (int c = 0; c < ZA_MSG_CHANNELS; ++c) {
            const uint32_t f = B->ch_flags[base + c];
            if (!(f & 4u)) { free_k = c; break; }
            if (B->ch_hash[base + c] == in.chan) { slot = c; break; }
          }
          if (slot < 0 && free_k >= 0) { slot = free_k; B->ch_hash[base + slot] = in.chan; B->ch_flags[base + slot] = 4u; B->ch_caps[base + slot] = 0; B->ch_dropped[base + slot] = 0; }
          if (slot >= 0) B->ch_dropped[base + slot] += 1u;
        } else {
          const uint64_t seq = ++*B->global_seq;
          ZaMsg out = in;
          out.seq = seq; out.src = me; out.pad = 0;
          if (!in.pad) out.target = 0;
          B->ring[seq % ZA_MSG_RING] = out;
          if (in.kind == 2)
            for (int c = 0; c < ZA_MSG_PAY; ++c)
              B->ring_pay[(int64_t)(seq % ZA_MSG_RING) * ZA_MSG_PAY + c] = c < (int)in.blen ? B->out_pay[((int64_t)i * ZA_MSG_OUTBOX + k) * ZA_MSG_PAY + c] : 0.0;
        }
      }
      ZA_MSG_CONVERGE();
    }
    if (lane == 0) { B->out_count[i] = 0; if (B->out_cells) B->out_cells[i] = 0; }
  }
}
