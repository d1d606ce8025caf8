"""RepNothing replica + KV state machine (BASELINE config 1): host-only mirror of
`RepNothingReplica` (src/protocols/rep_nothing/) over the C-ABI.  One call =
`handle_req_batch` with its WAL completion and command execution inline (LS-1
rule 0); replies come back in submission order."""
import ctypes as C

from . import _lib
from ._lib import check

GET, PUT = 0, 1


class RepNothingReplica:
    def __init__(self):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.smr_repnothing_create(C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_repnothing_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def handle_req_batch(self, reqs):
        """reqs: list of (client_id, req_id, ("get", key) | ("put", key, value)); keys / values are
        bytes or str.  Returns (instance index, [(client, req_id, kind, value-or-None), ...])."""
        n = len(reqs)
        b = lambda x: x.encode() if isinstance(x, str) else bytes(x)
        client = (C.c_uint64 * n)(*[r[0] for r in reqs])
        rid = (C.c_uint64 * n)(*[r[1] for r in reqs])
        kinds = [PUT if r[2][0] == "put" else GET for r in reqs]
        kind = (C.c_uint8 * n)(*kinds)
        keys = [b(r[2][1]) for r in reqs]
        vals = [b(r[2][2]) if k == PUT else b"" for r, k in zip(reqs, kinds)]
        key = (C.c_char_p * n)(*keys)
        klen = (C.c_uint32 * n)(*[len(k) for k in keys])
        val = (C.c_char_p * n)(*vals)
        vlen = (C.c_uint32 * n)(*[len(v) for v in vals])
        idx = C.c_uint64()
        check(self._L.smr_repnothing_submit_batch(self._h, n, client, rid, kind, key, klen, val, vlen, C.byref(idx)))
        return idx.value, self.poll_replies()

    def poll_replies(self, cap=1 << 16):
        out = []
        buf = C.create_string_buffer(cap)
        c, r, k, hv, ln = C.c_uint64(), C.c_uint64(), C.c_uint8(), C.c_int(), C.c_uint32()
        while True:
            rc = self._L.smr_repnothing_poll_reply(self._h, C.byref(c), C.byref(r), C.byref(k), C.byref(hv), buf, cap,
                                                   C.byref(ln))
            if rc == 0:
                return out
            if rc < 0:
                if ln.value > cap:                       # value larger than the buffer: grow and retry
                    cap = ln.value
                    buf = C.create_string_buffer(cap)
                    continue
                check(rc)
            out.append((c.value, r.value, "put" if k.value == PUT else "get", buf.raw[:ln.value] if hv.value else None))

    def stats(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._L.smr_repnothing_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"instances": a.value, "wal_offset": b.value, "executed": c.value, "keys": d.value}
