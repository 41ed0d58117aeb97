"""CPU model of the reference's scalar message bus for the instances of one engine. TEST INFRASTRUCTURE ONLY.

Follows src/DspJsfxRuntime.cpp:161-193,234-247,291-372,409-478 (per-instance runtime: outbox, ready inbox, subscriptions,
advertisements, dropped counters) and src/DspJsfxMessageBus.cpp:112-180,529-677,713-790 (ring of 4096 slots, target rules,
peer queries), with the engine's block order: every instance runs block k (beginBlock = collect, script, endBlock deferred),
then all outboxes are flushed in instance order. No reference test pins the bus (parity unpinned).
"""
from __future__ import annotations

RING, OUTBOX, INBOX, CHANNELS = 4096, 1024, 1024, 24
PAY, OUT_CELLS = 64, 65536          # payload cells per ring slot / per outbox and block
DEFAULT_DOMAIN = 0x9ae16a3b2f90404f


class Inst:
    def __init__(self, iid):
        self.id = iid
        self.domain = DEFAULT_DOMAIN
        self.chan = {}            # key -> [subscribed, advertised, caps, dropped]   (insertion ordered, <= 24)
        self.outbox, self.inbox, self.last_read = [], [], 0

    def slot(self, key, create):
        if key not in self.chan and create and len(self.chan) < CHANNELS:
            self.chan[key] = [False, False, 0, 0]
        return self.chan.get(key)


class BusRef:
    def __init__(self, n, first_id=1):
        self.inst = [Inst(first_id + i) for i in range(n)]
        self.first_id = first_id
        self.ring = {}            # seq -> message dict (seq % RING collisions overwrite)
        self.seq = 0

    @staticmethod
    def key(x):
        return int(x + (0.5 if x >= 0 else -0.5))

    # ---- per-instance operations (return the builtin's value) ----
    def comm_join(self, i, d):
        me, k = self.inst[i], self.key(d) or DEFAULT_DOMAIN
        if me.domain != k:
            me.inbox, me.last_read = [], 0
        me.domain = k
        return 1.0

    def subscribe(self, i, c):
        s = self.inst[i].slot(self.key(c), True)
        if s: s[0] = True
        return 1.0

    def unsubscribe(self, i, c):
        s = self.inst[i].slot(self.key(c), False)
        if s: s[0] = False
        return 1.0

    def advertise(self, i, c, caps):
        caps = max(0, self.key(caps))
        s = self.inst[i].slot(self.key(c), caps != 0)
        if s:
            s[2] = caps; s[1] = caps != 0
        return 1.0

    def _drop(self, me, key):
        s = me.slot(key, True)
        if s: s[3] += 1

    def send(self, i, c, tag, a, b, cc, d, target=None, buf=None):
        """buf: list of payload cells for msg_send_buf / msg_sendto_buf (copied when queued)."""
        me, key = self.inst[i], self.key(c)
        cells = sum(len(m["buf"]) for m in me.outbox if m["buf"] is not None)
        if len(me.outbox) >= OUTBOX or (buf is not None and cells + len(buf) > OUT_CELLS):
            self._drop(me, key)
            return 0.0
        me.outbox.append({"chan": key, "src": me.id, "target": (max(0, int(target + 0.5)) if target is not None else 0),
                          "direct": target is not None, "vals": (tag, a, b, cc, d), "buf": None if buf is None else list(buf)})
        return 1.0

    def _front(self, me, key):
        for m in me.inbox:
            if not m.get("used") and m["chan"] == key:
                return m
        return None

    def recv(self, i, c):
        me = self.inst[i]
        m = self._front(me, self.key(c))
        if m is None or m["buf"] is not None:        # a buffer at the front waits for recv_buf
            return 0.0, None
        m["used"] = True
        me.last_len = 0
        return 1.0, (float(m["src"]),) + m["vals"]

    def recv_buf(self, i, c, capacity):
        """-> (return value, (src, tag, copied cells) or None)"""
        me = self.inst[i]
        if capacity <= 0:
            return 0.0, None
        m = self._front(me, self.key(c))
        if m is None or m["buf"] is None:
            return 0.0, None
        m["used"] = True
        n = len(m["buf"])
        me.last_len = n
        return (float(n) if capacity >= n else -float(n)), (float(m["src"]), m["vals"][0], m["buf"][:int(min(capacity, n))])

    def length(self, i):
        return float(getattr(self.inst[i], "last_len", 0))

    def avail(self, i, c):
        key = self.key(c)
        return float(sum(1 for m in self.inst[i].inbox if not m.get("used") and m["chan"] == key))

    def kind(self, i, c):
        m = self._front(self.inst[i], self.key(c))
        return 0.0 if m is None else (2.0 if m["buf"] is not None else 1.0)

    def clear(self, i, c):
        key, n = self.key(c), 0
        for m in self.inst[i].inbox:
            if not m.get("used") and m["chan"] == key:
                m["used"] = True; n += 1
        return float(n)

    def dropped(self, i, c):
        s = self.inst[i].slot(self.key(c), False)
        return float(s[3]) if s else 0.0

    def _matches(self, inst, key, role):
        s = inst.chan.get(key)
        if not s:
            return False
        want_sub, want_pub = role in (1, 3) or role <= 0, role in (2, 3) or role <= 0
        return (want_sub and s[0]) or (want_pub and s[1])

    def peers(self, i, c, role):
        me, key, role = self.inst[i], self.key(c), self.key(role)
        return [p for p in self.inst if p.domain == me.domain and self._matches(p, key, role)]

    def peer_count(self, i, c, role):
        return float(len(self.peers(i, c, role)))

    def peer_id(self, i, c, role, index):
        ps, k = self.peers(i, c, role), self.key(index)
        return float(ps[k].id) if 0 <= k < len(ps) else 0.0

    def _peer(self, pid):
        j = int(pid + 0.5) - self.first_id
        return self.inst[j] if 0 <= j < len(self.inst) else None

    def peer_caps(self, i, pid):
        p = self._peer(pid)
        caps = 0
        if p:
            for s in p.chan.values():
                if s[1]: caps |= s[2]
        return float(caps)

    def peer_alive(self, i, pid):
        return 1.0 if self._peer(pid) else 0.0

    # ---- block boundaries ----
    def begin_block(self, i):
        me = self.inst[i]
        me.inbox = [m for m in me.inbox if not m.get("used")]
        newest = self.seq
        if newest > me.last_read:
            first = me.last_read + 1
            if newest > RING and first + RING <= newest:
                for s in me.chan.values():
                    if s[0]: s[3] += 1
                first = newest - RING + 1
            for q in range(first, newest + 1):
                m = self.ring.get(q % RING)
                if not m or m["seq"] != q:
                    continue
                mine = (m["target"] == me.id) if m["target"] else (m["src"] != me.id and self._matches(me, m["chan"], 1))
                if not mine:
                    continue
                if len(me.inbox) < INBOX:
                    me.inbox.append(dict(m, used=False))
                else:
                    self._drop(me, m["chan"])
            me.last_read = newest

    def flush_all(self):
        for me in self.inst:
            for m in me.outbox:
                if m["buf"] is not None and len(m["buf"]) > PAY:       # does not fit a ring slot
                    self._drop(me, m["chan"])
                    continue
                if m["direct"]:
                    t = self._peer(m["target"]) if m["target"] >= self.first_id else None
                    ok = t is not None and t.domain == me.domain
                else:
                    ok = any(p is not me and p.domain == me.domain and self._matches(p, m["chan"], 1) for p in self.inst)
                if not ok:
                    self._drop(me, m["chan"])
                    continue
                self.seq += 1
                self.ring[self.seq % RING] = {"seq": self.seq, "chan": m["chan"], "src": me.id,
                                              "target": m["target"] if m["direct"] else 0, "vals": m["vals"], "buf": m["buf"]}
            me.outbox = []
