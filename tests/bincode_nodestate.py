"""A reader for the bincode 1.3 image of librabft-v2's NodeState (node.rs:28-45 and the structs it nests: record_store.rs:93-134,
pacemaker.rs:60-77, record.rs:51-111, configuration.rs:18-22, simulated_context.rs:19-35, smr_context.rs:155-159), written from
the reference's struct definitions: fixed-width little-endian integers (usize as u64), Option = tag byte, Vec / HashMap = u64
length + elements, enum = u32 variant.  Test infrastructure: it cross-checks the two writers (oracle, device)."""
import struct


class Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.o)[0]
        self.o += 8
        return v

    def i64(self):
        v = struct.unpack_from("<q", self.d, self.o)[0]
        self.o += 8
        return v

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.o)[0]
        self.o += 4
        return v

    def f64(self):
        v = struct.unpack_from("<d", self.d, self.o)[0]
        self.o += 8
        return v

    def opt(self, f):
        tag = self.d[self.o]
        self.o += 1
        assert tag in (0, 1), tag
        return f() if tag else None

    def seq(self, f):
        return [f() for _ in range(self.u64())]


def signature(r):
    return (r.u64(), r.u64())  # Signature(usize, u64)


def block(r):
    return dict(command=(r.u64(), r.u64()), time=r.i64(), previous_quorum_certificate_hash=r.u64(), round=r.u64(), author=r.u64(), signature=signature(r))


def vote(r):
    return dict(epoch_id=r.u64(), round=r.u64(), certified_block_hash=r.u64(), state=r.u64(), committed_state=r.opt(r.u64), author=r.u64(), signature=signature(r))


def quorum_certificate(r):
    return dict(epoch_id=r.u64(), round=r.u64(), certified_block_hash=r.u64(), state=r.u64(), committed_state=r.opt(r.u64),
                votes=r.seq(lambda: (r.u64(), signature(r))), author=r.u64(), signature=signature(r))


def timeout(r):
    return dict(epoch_id=r.u64(), round=r.u64(), highest_certified_block_round=r.u64(), author=r.u64(), signature=signature(r))


def record_store(r):
    s = dict(epoch_id=r.u64())
    s["configuration"] = dict(authors=r.seq(lambda: (r.u64(), r.u64())), voting_rights=r.seq(lambda: (r.u64(), r.u64())), total_votes=r.u64())
    s["initial_hash"], s["initial_state"] = r.u64(), r.u64()
    s["blocks"] = r.seq(lambda: (r.u64(), block(r)))
    s["quorum_certificates"] = r.seq(lambda: (r.u64(), quorum_certificate(r)))
    s["current_proposed_block"] = r.opt(r.u64)
    for k in ("highest_quorum_certificate_round", "highest_quorum_certificate_hash", "highest_timeout_certificate_round", "current_round",
              "highest_committed_round"):
        s[k] = r.u64()
    s["highest_commit_certificate_hash"] = r.opt(r.u64)
    s["highest_timeout_certificate"] = r.opt(lambda: r.seq(lambda: timeout(r)))
    s["current_timeouts"] = r.seq(lambda: (r.u64(), timeout(r)))
    s["current_votes"] = r.seq(lambda: (r.u64(), vote(r)))
    s["current_timeouts_weight"] = r.u64()
    variant = r.u32()
    if variant == 0:
        s["current_election"] = ("Ongoing", r.seq(lambda: ((r.u64(), r.u64()), r.u64())))
    elif variant == 1:
        s["current_election"] = ("Won", r.u64(), r.u64())
    else:
        assert variant == 2, variant
        s["current_election"] = ("Closed",)
    return s


def node_state(data):
    """bytes -> dict; asserts that the image is consumed exactly."""
    r = Reader(data)
    n = dict(record_store=record_store(r))
    n["pacemaker"] = dict(active_epoch=r.u64(), active_round=r.u64(), active_leader=r.opt(r.u64), active_round_start_time=r.i64(),
                          active_round_duration=r.i64(), delta=r.i64(), gamma=r.f64(), lambda_=r.f64())
    n["epoch_id"], n["latest_voted_round"], n["locked_round"], n["latest_query_all_time"] = r.u64(), r.u64(), r.u64(), r.i64()
    n["tracker"] = dict(epoch_id=r.u64(), highest_committed_round=r.u64(), latest_commit_time=r.i64(), target_commit_interval=r.i64())
    n["past_record_stores"] = r.seq(lambda: (r.u64(), record_store(r)))
    assert r.o == len(data), (r.o, len(data))
    return n


# ---- the inverse: dict -> bytes (save -> load -> save round trip, librabft-v2/src/unit_tests/node_tests.rs:17-21) ----
class Writer:
    def __init__(self):
        self.b = bytearray()

    def u64(self, v):
        self.b += struct.pack("<Q", v)

    def i64(self, v):
        self.b += struct.pack("<q", v)

    def u32(self, v):
        self.b += struct.pack("<I", v)

    def f64(self, v):
        self.b += struct.pack("<d", v)

    def opt(self, v, f):
        self.b.append(0 if v is None else 1)
        if v is not None:
            f(v)

    def seq(self, items, f):
        self.u64(len(items))
        for it in items:
            f(it)


def _w_signature(w, s):
    w.u64(s[0]); w.u64(s[1])


def _w_block(w, b):
    w.u64(b["command"][0]); w.u64(b["command"][1]); w.i64(b["time"]); w.u64(b["previous_quorum_certificate_hash"]); w.u64(b["round"]); w.u64(b["author"])
    _w_signature(w, b["signature"])


def _w_vote(w, v):
    w.u64(v["epoch_id"]); w.u64(v["round"]); w.u64(v["certified_block_hash"]); w.u64(v["state"]); w.opt(v["committed_state"], w.u64); w.u64(v["author"])
    _w_signature(w, v["signature"])


def _w_qc(w, q):
    w.u64(q["epoch_id"]); w.u64(q["round"]); w.u64(q["certified_block_hash"]); w.u64(q["state"]); w.opt(q["committed_state"], w.u64)
    w.seq(q["votes"], lambda v: (w.u64(v[0]), _w_signature(w, v[1])))
    w.u64(q["author"])
    _w_signature(w, q["signature"])


def _w_timeout(w, t):
    w.u64(t["epoch_id"]); w.u64(t["round"]); w.u64(t["highest_certified_block_round"]); w.u64(t["author"])
    _w_signature(w, t["signature"])


def _w_record_store(w, s):
    w.u64(s["epoch_id"])
    c = s["configuration"]
    w.seq(c["authors"], lambda p: (w.u64(p[0]), w.u64(p[1])))
    w.seq(c["voting_rights"], lambda p: (w.u64(p[0]), w.u64(p[1])))
    w.u64(c["total_votes"])
    w.u64(s["initial_hash"]); w.u64(s["initial_state"])
    w.seq(s["blocks"], lambda kv: (w.u64(kv[0]), _w_block(w, kv[1])))
    w.seq(s["quorum_certificates"], lambda kv: (w.u64(kv[0]), _w_qc(w, kv[1])))
    w.opt(s["current_proposed_block"], w.u64)
    for k in ("highest_quorum_certificate_round", "highest_quorum_certificate_hash", "highest_timeout_certificate_round", "current_round",
              "highest_committed_round"):
        w.u64(s[k])
    w.opt(s["highest_commit_certificate_hash"], w.u64)
    w.opt(s["highest_timeout_certificate"], lambda tc: w.seq(tc, lambda t: _w_timeout(w, t)))
    w.seq(s["current_timeouts"], lambda kv: (w.u64(kv[0]), _w_timeout(w, kv[1])))
    w.seq(s["current_votes"], lambda kv: (w.u64(kv[0]), _w_vote(w, kv[1])))
    w.u64(s["current_timeouts_weight"])
    e = s["current_election"]
    if e[0] == "Ongoing":
        w.u32(0)
        w.seq(e[1], lambda en: (w.u64(en[0][0]), w.u64(en[0][1]), w.u64(en[1])))
    elif e[0] == "Won":
        w.u32(1); w.u64(e[1]); w.u64(e[2])
    else:
        w.u32(2)


def dump_node_state(n):
    """dict (as node_state returns it) -> the bincode image."""
    w = Writer()
    _w_record_store(w, n["record_store"])
    p = n["pacemaker"]
    w.u64(p["active_epoch"]); w.u64(p["active_round"]); w.opt(p["active_leader"], w.u64); w.i64(p["active_round_start_time"])
    w.i64(p["active_round_duration"]); w.i64(p["delta"]); w.f64(p["gamma"]); w.f64(p["lambda_"])
    w.u64(n["epoch_id"]); w.u64(n["latest_voted_round"]); w.u64(n["locked_round"]); w.i64(n["latest_query_all_time"])
    t = n["tracker"]
    w.u64(t["epoch_id"]); w.u64(t["highest_committed_round"]); w.i64(t["latest_commit_time"]); w.i64(t["target_commit_interval"])
    w.seq(n["past_record_stores"], lambda kv: (w.u64(kv[0]), _w_record_store(w, kv[1])))
    return bytes(w.b)
