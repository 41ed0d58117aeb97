"""Message bus (scalar and buffer messages) between the instances of one engine against oracle/msg_ref.py, the model of the
reference's per-instance runtime + ring (src/DspJsfxRuntime.cpp, src/DspJsfxMessageBus.cpp). No reference test pins the bus
(parity unpinned); every returned value, received payload and counter is compared exactly, block by block, and the
reference's own two-instance probe leaf (Control/IPCProbeA) is run as sender + receiver."""
import numpy as np
import pytest

N = 5
CH, CH2, DOM = 101.0, 202.0, 9001.0
# one row of (op, a, b, c) per instance per block; op 0 = idle
SCRIPT = [
    [(2, CH, 0, 0), (2, CH, 0, 0), (4, CH, 3, 0), (2, CH2, 0, 0), (1, DOM, 0, 0)],          # subscribe / advertise / join other domain
    [(12, CH, 1, 0), (12, CH, 2, 0), (12, CH, 3, 0), (12, CH2, 0, 0), (12, CH, 3, 0)],     # peer counts by role
    [(5, CH, 11, 1.5), (0, 0, 0, 0), (5, CH, 12, 2.5), (5, CH, 13, 0), (5, CH, 14, 0)],    # broadcasts (inst 4: other domain -> dropped)
    [(7, CH, 0, 0), (8, CH, 0, 0), (7, CH, 0, 0), (7, CH, 0, 0), (11, CH, 0, 0)],          # receive: own messages excluded
    [(7, CH, 0, 0), (7, CH, 0, 0), (11, CH, 0, 0), (8, CH2, 0, 0), (0, 0, 0, 0)],
    [(6, CH2, 4, 77), (6, CH, 99, 78), (13, CH, 1, 0), (13, CH, 1, 1), (13, CH, 1, 5)],    # direct messages (one to a non-instance), peer ids
    [(0, 0, 0, 0), (11, CH, 0, 0), (14, 3, 0, 0), (7, CH2, 0, 0), (15, 2, 0, 0)],          # direct one arrives without subscription to its sender
    [(17, CH, 30, 0), (0, 0, 0, 0), (3, CH, 0, 0), (9, CH, 0, 0), (16, 0, 0, 0)],          # burst of 30, unsubscribe
    [(0, 0, 0, 0), (8, CH, 0, 0), (8, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0)],
    [(0, 0, 0, 0), (10, CH, 0, 0), (7, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0)],
    [(0, 0, 0, 0), (8, CH, 0, 0), (12, CH, 1, 0), (0, 0, 0, 0), (1, 0, 0, 0)],             # rejoin the default domain
    [(5, CH, 21, 4), (3, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (2, CH, 0, 0)],            # inst 1 unsubscribes, inst 4 subscribes
    [(5, CH, 22, 5), (7, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (7, CH, 0, 0)],
    [(0, 0, 0, 0), (7, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (7, CH, 0, 0)],
    # buffer messages (msg_send_buf / msg_sendto_buf / msg_recv_buf): inst 0, 4 subscribed to CH by now, inst 3 to CH2
    [(18, CH, 10, 31), (10, CH, 0, 0), (19, CH2, 64, 4), (18, CH, 65, 32), (10, CH, 0, 0)],   # 10 cells; direct 64 cells to inst 3 (id 4); 65 cells: too long for a slot
    [(9, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (9, CH2, 0, 0), (9, CH, 0, 0)],                # kind: 2 = buffer
    [(7, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (20, CH2, 100, 0), (20, CH, 4, 0)],            # scalar recv refuses a buffer; full copy; capacity 4 < 10 -> -10
    [(20, CH, 16, 0), (0, 0, 0, 0), (11, CH2, 0, 0), (11, CH, 0, 0), (16, 0, 0, 0)],           # inst 0 has nothing (own message); drop counters; msg_length
    [(5, CH, 41, 6), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (18, CH, 3, 33)],               # a scalar, then a buffer, on one channel
    [(20, CH, 8, 0), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (20, CH, 8, 0)],                # inst 0: buffer from inst 4; inst 4: scalar at the front -> 0
    [(16, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (7, CH, 0, 0)],
    [(21, CH, 1030, 0), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0)],               # 1030 x 64 cells: 1024 fit the message count, the rest drop
    [(11, CH, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0), (8, CH, 0, 0)],
]
FIELDS = ("ret", "src", "tag", "ma", "mb", "mc", "md", "r0", "r1", "rlast")


def _model_step(bus, state, rows):
    for i, (op, a, b, c) in enumerate(rows):
        bus.begin_block(i)
        st = state[i]
        iid = float(bus.inst[i].id)
        if op == 1: st["ret"] = bus.comm_join(i, a)
        elif op == 2: st["ret"] = bus.subscribe(i, a)
        elif op == 3: st["ret"] = bus.unsubscribe(i, a)
        elif op == 4: st["ret"] = bus.advertise(i, a, b)
        elif op == 5: st["ret"] = bus.send(i, a, b, c, iid, 0.5, -2.0)
        elif op == 6: st["ret"] = bus.send(i, a, c, iid, 1.0, 2.0, 3.0, target=b)
        elif op == 7:
            st["ret"], got = bus.recv(i, a)
            if got:
                st["src"], st["tag"], st["ma"], st["mb"], st["mc"], st["md"] = got
        elif op == 8: st["ret"] = bus.avail(i, a)
        elif op == 9: st["ret"] = bus.kind(i, a)
        elif op == 10: st["ret"] = bus.clear(i, a)
        elif op == 11: st["ret"] = bus.dropped(i, a)
        elif op == 12: st["ret"] = bus.peer_count(i, a, b)
        elif op == 13: st["ret"] = bus.peer_id(i, a, b, c)
        elif op == 14: st["ret"] = bus.peer_caps(i, a)
        elif op == 15: st["ret"] = bus.peer_alive(i, a)
        elif op == 16: st["ret"] = bus.length(i)
        elif op == 17:
            r = 0.0
            for _ in range(int(b)):
                r += bus.send(i, a, 7.0, r, iid, 0.0, 0.0)
            st["ret"] = r
        elif op == 18: st["ret"] = bus.send(i, a, c, 0.0, 0.0, 0.0, 0.0, buf=[iid * 1000 + k for k in range(int(b))])
        elif op == 19: st["ret"] = bus.send(i, a, 5.0, 0.0, 0.0, 0.0, 0.0, target=c, buf=[iid * 1000 + 3 + k for k in range(int(b))])
        elif op == 20:
            rx = [-1.0] * 200
            st["ret"], got = bus.recv_buf(i, a, int(b))
            if got:
                st["src"], st["tag"] = got[0], got[1]
                rx[:len(got[2])] = got[2]
            n = int(min(abs(st["ret"]), b))
            st["r0"], st["r1"], st["rlast"] = rx[0], rx[1], rx[max(0, n - 1)]
            st["ma"], st["mb"] = bus.length(i), rx[n]
        elif op == 21:
            r = 0.0
            for _ in range(int(b)):
                r += bus.send(i, a, 9.0, 0.0, 0.0, 0.0, 0.0, buf=[iid * 1000 + k for k in range(64)])
            st["ret"] = r
    bus.flush_all()


def _expected():
    from oracle import msg_ref
    bus = msg_ref.BusRef(N)
    state = [{k: (-1.0 if k in ("r0", "r1", "rlast") else -7.0) for k in FIELDS} for _ in range(N)]
    out = []
    for rows in SCRIPT:
        _model_step(bus, state, rows)
        out.append([dict(s) for s in state])
    return out


def test_model_is_self_consistent():
    """Spot checks of the model against the semantics listed in its header."""
    exp = _expected()
    assert [s["ret"] for s in exp[1]] == [2.0, 1.0, 3.0, 1.0, 0.0]        # subscribers, advertisers, either; other domain sees none
    assert exp[3][0]["ret"] == 1.0 and exp[3][0]["src"] == 3.0 and exp[3][0]["tag"] == 12.0     # inst 0 gets inst 2's, not its own
    assert exp[3][1]["ret"] == 3.0                                         # inst 1: three broadcasts waiting (0, 2, 3)
    assert exp[3][4]["ret"] == 1.0                                         # inst 4 sent into an empty domain: dropped once
    assert exp[6][3]["ret"] == 1.0 and exp[6][3]["src"] == 1.0 and exp[6][3]["tag"] == 77.0     # direct message, no subscription needed
    assert exp[6][1]["ret"] == 1.0                                         # direct message to a non-instance: dropped
    assert exp[8][1]["ret"] == 32.0 and exp[8][2]["ret"] == 0.0            # 2 still queued + burst of 30; none for the advertiser-only instance
    # buffer messages
    assert exp[15][4]["ret"] == 2.0 and exp[15][3]["ret"] == 2.0           # kind = buffer
    assert exp[16][0]["ret"] == 0.0                                        # inst 0 sent it: nothing for itself
    assert exp[16][3]["ret"] == 64.0 and exp[16][3]["r0"] == 3003.0 and exp[16][3]["rlast"] == 3066.0 and exp[16][3]["mb"] == -1.0
    assert exp[16][4]["ret"] == -10.0 and exp[16][4]["rlast"] == 1003.0 and exp[16][4]["ma"] == 10.0 and exp[16][4]["mb"] == -1.0
    assert exp[17][3]["ret"] == 1.0                                        # the 65-cell payload was dropped at the flush
    assert exp[19][4]["ret"] == 0.0 and exp[20][4]["ret"] == 1.0           # a scalar at the front: recv_buf refuses, recv takes it
    assert exp[21][0]["ret"] == 1024.0 and exp[22][0]["ret"] == 6.0


SOLO = [   # one instance (id 1) talking to itself with direct messages: the port's bus holds a single instance
    (19, CH, 10, 1), (9, CH, 0, 0), (7, CH, 0, 0), (20, CH, 4, 0), (16, 0, 0, 0),
    (19, CH, 64, 1), (20, CH, 100, 0), (19, CH, 65, 1), (20, CH, 100, 0), (11, CH, 0, 0),
    (6, CH, 9, 1), (19, CH, 2, 1), (20, CH, 8, 0), (7, CH, 0, 0), (16, 0, 0, 0), (20, CH, 8, 0), (16, 0, 0, 0),
    (21, CH, 1030, 0), (11, CH, 0, 0),
]


def test_port_buffer_messages_match_model():
    """The same zart_msg.h code compiled by g++ (the CPU port), one instance sending to itself."""
    from oracle import port
    from oracle.msg_ref import BusRef
    if not port.port_path("fx_msgkat").exists():
        pytest.skip("fixture port not built")
    p = port.Port("fx_msgkat", 48000.0, mem_cap=1 << 16)
    p.set_sliders([0, 0, 0, 0]); p.prepare()
    bus = BusRef(1)
    state = [{k: (-1.0 if k in ("r0", "r1", "rlast") else -7.0) for k in FIELDS}]
    for b, row in enumerate(SOLO):
        _model_step(bus, state, [row])
        p.set_sliders(list(row))
        p.process(np.zeros((1, 16), np.float32), 16)
        got = {k: p.var(k) for k in FIELDS}
        assert got == state[0], (b, row, got, state[0])


@pytest.mark.gpu
def test_gpu_bus_matches_model():
    import zabatch
    exp = _expected()
    with zabatch.Engine("fx_msgkat", N, max_block=16) as e:
        e.set_sliders([0, 0, 0, 0]); e.prepare()
        names = e.var_names()
        idx = [names.index(k) for k in FIELDS]
        for b, rows in enumerate(SCRIPT):
            sl = np.zeros((N, 64)); sl[:, :4] = np.array(rows, dtype=np.float64)
            e.set_sliders(sl)
            e.process_host(np.zeros((N, 1, 16), np.float32), block=16)
            v = e.read_vars()
            for i in range(N):
                got = {k: v[i, j] for k, j in zip(FIELDS, idx)}
                assert got == exp[b][i], (b, i, rows[i], got, exp[b][i])


@pytest.mark.gpu
def test_reference_ipc_probe_sender_and_receivers():
    """Control/IPCProbeA is the reference's own two-instance probe (docs/DSP-JSFX-Communication.md:141-156): one sender, here
    two receivers. A message sent in block k is received in block k + 1; the receivers' tone level follows rx_count."""
    import zabatch
    if not zabatch.module_path("IPCProbeA").exists():
        pytest.skip("IPCProbeA not built")
    meta = zabatch.leaf_meta("IPCProbeA")
    n, block, blocks = 3, 64, 12
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows[:, 0] = [0, 1, 1]                                   # sender, receiver, receiver
    with zabatch.Engine("IPCProbeA", n, max_block=block, first_instance_id=40) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(np.zeros((n, 2, block * blocks), np.float32), block=block)
        v = e.read_vars(); nm = e.var_names()
        g = e.gmem_read(0, 8)
    get = lambda i, k: v[i, nm.index(k)]
    assert get(0, "seq") == blocks and get(0, "rx_count") == 0
    for r in (1, 2):
        assert get(r, "rx_count") == blocks - 1 and get(r, "rx_seq") == blocks - 1 and get(r, "rx_src") == 40
    assert [get(i, "last_peer_count") for i in range(n)] == [3, 3, 3]
    assert g[0] == 40 and g[1] == blocks and g[2] == 40 and g[3] == blocks - 1 and g[4] == blocks - 1
    assert not y[0].any() and np.abs(y[1]).max() > 0 and np.array_equal(y[1], y[2])
    amp_first = np.abs(y[1][0, :block]).max()                # nothing received during the first block
    assert amp_first == 0.0
