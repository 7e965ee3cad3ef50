"""Read what the reference writes with JLD2 - without an HDF5 library (there is none in the image).

The reference keeps data sets and trained models in ``.jld2`` files (``jldopen`` / ``@save``: docs/src/tools.md:61-62,
src/Training/RealRealHighDimension.jl:309-316, src/Structs/options.jl:422-427 ``TrainedMPS``).  JLD2 files are HDF5 files
of a narrow shape: superblock version 2 at offset 512, version-2 object headers, link messages in the root group,
committed (shared) compound datatypes whose member names are the Julia field names, object references for mutable
values and arrays of them, variable-length strings in global heap collections, contiguous or compact data layouts
(JLD2 does not compress unless asked to).  That subset is what this module reads:

* ``JLD2File(path).keys()`` / ``.read(name)``: numeric arrays come back as NumPy arrays in Julia's shape (column-major
  data, so ``X_train`` is (N, T) as in Julia), ``Complex`` as complex arrays, structs as dicts keyed by field name
  (tuples by "1", "2", ...), Strings / Symbols as ``str``, arrays of structs as lists;
* ``load_trained_mps_jld2(path)``: a saved ``TrainedMPS`` (the ITensors MPS, the MPSOptions, the EncodedTimeSeriesSet) as
  this package's ``TrainedMPS`` - site tensors as (Dl, d, Dr) arrays with the label index on the site that carries it,
  options by field name.

Chunked / compressed data sets and version-1 object headers raise ``NotImplementedError`` with the name of the feature.
Writing is out of scope (models leave this package as ``.npz``); checked against the reference's own
fixture ``test/Data/ecg200/mps_saves/test_dataset.jld2`` (tests/test_jld2.py).
"""
from __future__ import annotations

import struct
from typing import Any, Dict, List, Optional

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class JLD2File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.f = fh.read()
        f = self.f
        sb = next((o for o in (512, 0, 1024, 2048) if f[o:o + 8] == _SIG), None)
        if sb is None:
            raise ValueError(f"{path}: no HDF5 superblock (not a JLD2 file)")
        if f[sb + 8] not in (2, 3):
            raise NotImplementedError(f"HDF5 superblock version {f[sb + 8]} (JLD2 writes version 2)")
        if f[sb + 9] != 8 or f[sb + 10] != 8:
            raise NotImplementedError("HDF5 files with offsets / lengths other than 8 bytes")
        self.base, _, self.eof, self.root = struct.unpack("<QQQQ", f[sb + 12:sb + 44])
        self._links = self._read_links(self.root)
        self._cache: Dict[int, Any] = {}

    # ---- object headers -----------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """(type, flags, body) of every message of the version-2 object header at file address `addr`."""
        f, base = self.f, self.base
        pos = base + addr
        if f[pos:pos + 4] != b"OHDR" or f[pos + 4] != 2:
            raise NotImplementedError("object header that is not version 2 (JLD2 writes version 2)")
        flags = f[pos + 5]
        p = pos + 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
        szb = 1 << (flags & 3)
        size = int.from_bytes(f[p:p + szb], "little")
        p += szb
        out = []

        def walk(p, end):
            while p + 4 <= end:
                t, sz, mfl = f[p], struct.unpack("<H", f[p + 1:p + 3])[0], f[p + 3]
                p += 4 + (2 if flags & 0x04 else 0)
                body = f[p:p + sz]
                if t == 0x10:                                       # continuation block
                    off, ln = struct.unpack("<QQ", body[:16])
                    if f[base + off:base + off + 4] != b"OCHK":
                        raise ValueError("corrupt object header continuation")
                    walk(base + off + 4, base + off + ln - 4)
                elif t != 0:
                    out.append((t, mfl, body))
                p += sz

        walk(p, p + size)
        return out

    def _read_links(self, addr):
        out = {}
        for t, _, d in self._messages(addr):
            if t != 0x06:
                continue
            fl, p, ltype = d[1], 2, 0
            if fl & 0x08:
                ltype = d[p]
                p += 1
            if fl & 0x04:
                p += 8
            if fl & 0x10:
                p += 1
            lsz = 1 << (fl & 3)
            ln = int.from_bytes(d[p:p + lsz], "little")
            p += lsz
            name = d[p:p + ln].decode()
            p += ln
            if ltype == 0:
                out[name] = struct.unpack("<Q", d[p:p + 8])[0]
        return out

    # ---- datatypes ----------------------------------------------------------------------------------------------------
    def _parse_dt(self, d, p=0):
        cls, ver = d[p] & 15, d[p] >> 4
        b0, b1 = d[p + 1], d[p + 2]
        size = struct.unpack("<I", d[p + 4:p + 8])[0]
        q = p + 8
        if cls == 0:
            return dict(cls="int", size=size, signed=bool(b0 & 8)), q + 4
        if cls == 1:
            return dict(cls="float", size=size), q + 12
        if cls == 3:
            return dict(cls="string", size=size), q
        if cls == 4:
            return dict(cls="bitfield", size=size), q + 4
        if cls == 5:
            return dict(cls="opaque", size=size), q + ((b0 + 7) // 8) * 8
        if cls == 7:
            return dict(cls="ref", size=size), q
        if cls == 9:
            bt, q2 = self._parse_dt(d, q)
            return dict(cls="vlen", size=size, string=(b0 & 15) == 1, base=bt), q2
        if cls == 6:
            members = []
            for _ in range(b0 | (b1 << 8)):
                e = d.index(b"\0", q)
                name = d[q:e].decode()
                if ver == 3:
                    q = e + 1
                    nb = 1 if size < 256 else 2 if size < 65536 else 4 if size < 2 ** 32 else 8
                    off = int.from_bytes(d[q:q + nb], "little")
                    q += nb
                else:
                    q += ((e - q + 1 + 7) // 8) * 8
                    off = struct.unpack("<I", d[q:q + 4])[0]
                    q += 4 + (28 if ver == 1 else 0)
                mt, q = self._parse_dt(d, q)
                members.append((name, off, mt))
            return dict(cls="compound", size=size, members=members), q
        if cls == 10:
            rank = d[q]
            q += 1 + (3 if ver == 2 else 0)
            dims = struct.unpack("<%dI" % rank, d[q:q + 4 * rank])
            q += 4 * rank * (2 if ver == 2 else 1)
            bt, q = self._parse_dt(d, q)
            return dict(cls="array", size=size, dims=dims, base=bt), q
        raise NotImplementedError(f"HDF5 datatype class {cls}")

    def _datatype(self, msgs):
        for t, fl, d in msgs:
            if t == 0x03:
                if fl & 2:                                           # shared message -> committed datatype
                    for t2, _, d2 in self._messages(struct.unpack("<Q", d[2:10])[0]):
                        if t2 == 0x03:
                            return self._parse_dt(d2)[0]
                    raise ValueError("committed datatype without a datatype message")
                return self._parse_dt(d)[0]
        raise ValueError("object without a datatype message")

    # ---- values -------------------------------------------------------------------------------------------------------
    def _heap(self, addr, idx):
        f, p = self.f, self.base + addr
        if f[p:p + 4] != b"GCOL":
            raise ValueError("corrupt global heap reference")
        size = struct.unpack("<Q", f[p + 8:p + 16])[0]
        q = p + 16
        while q + 16 <= p + size:
            i, _, _, sz = struct.unpack("<HHIQ", f[q:q + 16])
            if i == 0:
                break
            if i == idx:
                return f[q + 16:q + 16 + sz]
            q += 16 + ((sz + 7) // 8) * 8
        raise KeyError(f"global heap object {idx} not found")

    def _decode(self, dt, b):
        c = dt["cls"]
        if c == "int":
            return int.from_bytes(b[:dt["size"]], "little", signed=dt["signed"])
        if c == "float":
            return float(np.frombuffer(b[:dt["size"]], dtype="<f%d" % dt["size"])[0])
        if c == "bitfield":
            return bool(b[0]) if dt["size"] == 1 else int.from_bytes(b[:dt["size"]], "little")
        if c == "string":
            return b[:dt["size"]].split(b"\0")[0].decode()
        if c == "opaque":
            return bytes(b[:dt["size"]])
        if c == "ref":
            a = struct.unpack("<Q", b[:8])[0]
            return None if a == 0 else self._read_obj(a)
        if c == "vlen":
            n, addr, idx = struct.unpack("<IQI", b[:16])
            if n == 0 or addr == 0:
                return "" if dt["string"] else []
            raw = self._heap(addr, idx)
            if dt["string"]:
                return raw[:n].decode()
            bs = dt["base"]["size"]
            return [self._decode(dt["base"], raw[i * bs:(i + 1) * bs]) for i in range(n)]
        if c == "compound":
            names = [m[0] for m in dt["members"]]
            if names == ["re", "im"] and all(m[2]["cls"] == "float" for m in dt["members"]):
                return complex(*np.frombuffer(b[:dt["size"]], dtype="<f%d" % (dt["size"] // 2)))
            return {name: self._decode(mt, b[off:off + mt["size"]]) for name, off, mt in dt["members"]}
        if c == "array":
            bs = dt["base"]["size"]
            n = int(np.prod(dt["dims"]))
            return [self._decode(dt["base"], b[i * bs:(i + 1) * bs]) for i in range(n)]
        raise NotImplementedError(c)

    def _read_obj(self, addr):
        if addr in self._cache:
            return self._cache[addr]
        ms = self._messages(addr)
        dt = self._datatype(ms)
        space = next(d for t, _, d in ms if t == 0x01)
        lay = next(d for t, _, d in ms if t == 0x08)
        if space[0] != 2:
            raise NotImplementedError(f"dataspace message version {space[0]}")
        rank, stype = space[1], space[3]
        dims = struct.unpack("<%dQ" % rank, space[4:4 + 8 * rank])
        n = 0 if stype == 2 else (int(np.prod(dims)) if rank else 1)
        es = dt["size"]
        if lay[0] not in (3, 4):
            raise NotImplementedError(f"data layout message version {lay[0]}")
        if lay[1] == 0:
            raw = lay[4:4 + struct.unpack("<H", lay[2:4])[0]]
        elif lay[1] == 1:
            a, sz = struct.unpack("<QQ", lay[2:18])
            raw = self.f[self.base + a:self.base + a + sz] if n else b""
        else:
            raise NotImplementedError("chunked (compressed) data set: re-save it with JLD2's default compress=false")
        shape = dims[::-1]                                          # HDF5 lists the slowest dimension first, Julia is column-major
        if rank == 0 and stype != 2:
            val = self._decode(dt, raw)
        elif dt["cls"] in ("int", "float"):
            code = ("<f%d" % es) if dt["cls"] == "float" else ("<%s%d" % ("i" if dt["signed"] else "u", es))
            val = np.frombuffer(raw[:n * es], dtype=code).reshape(shape, order="F").copy() if n else np.zeros(shape, dtype=code)
        elif dt["cls"] == "bitfield" and es == 1:
            val = (np.frombuffer(raw[:n], dtype="u1") != 0).reshape(shape, order="F")
        elif dt["cls"] == "compound" and [m[0] for m in dt["members"]] == ["re", "im"]:
            val = np.frombuffer(raw[:n * es], dtype="<c%d" % es).reshape(shape, order="F").copy()
        else:
            vals = [self._decode(dt, raw[i * es:(i + 1) * es]) for i in range(n)]
            if rank > 1 and n:
                arr = np.empty(n, dtype=object)
                arr[:] = vals
                val = arr.reshape(shape, order="F")
            else:
                val = vals
        self._cache[addr] = val
        return val

    # ---- public -------------------------------------------------------------------------------------------------------
    def keys(self) -> List[str]:
        return [k for k in self._links if k != "_types"]

    def __contains__(self, name):
        return name in self._links and name != "_types"

    def read(self, name: str):
        if name not in self:
            raise KeyError(f"{name!r} not in file; it holds {self.keys()}")
        return self._read_obj(self._links[name])


def read_jld2(path, name: Optional[str] = None):
    """One named entry of a JLD2 file, or (name=None) all of them as a dict."""
    f = JLD2File(path)
    return f.read(name) if name is not None else {k: f.read(k) for k in f.keys()}


# ---- TrainedMPS -------------------------------------------------------------------------------------------------------
def _tags(index) -> List[str]:
    """ITensors TagSet: up to four tags, each a UInt256 holding its characters (2 bytes each) from the top."""
    ts = index["tags"]
    data = ts["data"]
    while isinstance(data, dict) and "data" in data:
        data = data["data"]
    vals = [data[k] for k in sorted(data, key=int)] if isinstance(data, dict) else list(data)
    out = []
    for v in vals[:int(ts.get("length", len(vals)))]:
        raw = v if isinstance(v, (bytes, bytearray)) else int(v).to_bytes(32, "little")
        out.append(bytes(c for c in raw[::-1] if c).decode())
    return out


def _find_trained(f: JLD2File, key):
    if key is not None:
        return f.read(key)
    for k in f.keys():
        v = f.read(k)
        if isinstance(v, dict) and {"mps", "opts", "train_data"} <= set(v):
            return v
    raise KeyError("no TrainedMPS (a struct with fields mps, opts, train_data) in this file")


def load_trained_mps_jld2(path, key: Optional[str] = None):
    """A ``TrainedMPS`` saved by the reference with JLD2 (``jldsave(path; mps=trained)`` or ``@save``) as this package's
    ``TrainedMPS``: src/Structs/options.jl:422-427 (fields), src/Structs/structs.jl:12-33 (PState, EncodedTimeSeriesSet)."""
    from .encodings import EncodedTimeSeriesSet
    from .options import MPSOptions
    from .training import TrainedMPS

    tm = _find_trained(JLD2File(path), key)
    # ---- MPS: ITensor = (tensor = (storage = Dense(data), inds = (Index...)))
    sites = tm["mps"]["data"]
    tens = []
    for it in sites:
        t = it["tensor"] if "tensor" in it else it
        inds = t["inds"]
        inds = [inds[k] for k in sorted(inds, key=int)] if isinstance(inds, dict) else list(inds)
        data = t["storage"]["data"]
        tens.append((np.asarray(data), [(int(i["id"]), int(i["space"]), _tags(i)) for i in inds]))
    T = len(tens)
    W = []
    for j, (data, inds) in enumerate(tens):
        ids_prev = {i[0] for i in tens[j - 1][1]} if j > 0 else set()
        ids_next = {i[0] for i in tens[j + 1][1]} if j + 1 < T else set()
        dims = [i[1] for i in inds]
        arr = data.reshape(dims, order="F")                           # Julia: first index fastest
        role = {}
        for ax, (iid, dim, tags) in enumerate(inds):
            if iid in ids_prev:
                role["l"] = ax
            elif iid in ids_next:
                role["r"] = ax
            elif "Site" in tags:
                role["s"] = ax
            else:
                role["c"] = ax                                        # the label index ("f(x)")
        if "s" not in role or len(role) != len(inds):
            raise ValueError(f"site {j}: cannot tell the roles of the indices {[(d, t) for _, d, t in inds]}")
        order = [role[k] for k in ("l", "s", "r", "c") if k in role]
        arr = arr.transpose(order)
        shape = [arr.shape[order.index(role[k])] if k in role else 1 for k in ("l", "s", "r")]
        if "c" in role:
            shape.append(arr.shape[-1])
        W.append(np.ascontiguousarray(arr.reshape(shape)))
    # ---- options, by field name
    o = dict(tm["opts"])
    known = set(MPSOptions.__dataclass_fields__)
    dt_name = o["dtype"]["name"] if isinstance(o.get("dtype"), dict) else str(o.get("dtype", ""))
    o["dtype"] = dt_name.split(".")[-1]
    if o["dtype"].startswith("Complex") and isinstance(tm["opts"].get("dtype"), dict) and tm["opts"]["dtype"].get("parameters"):
        o["dtype"] = "ComplexF64"
    for k in ("rescale", "data_bounds"):
        if isinstance(o.get(k), dict):
            o[k] = tuple(o[k][i] for i in sorted(o[k], key=int))
    opts = MPSOptions(**{k: v for k, v in o.items() if k in known})
    # ---- training set
    td = tm["train_data"]
    ts = td["timeseries"] or []
    if ts:
        phi = np.stack([np.stack([np.asarray(v) for v in p["pstate"]]) for p in ts])
        labels = np.array([p["label"] for p in ts])
        label_index = np.array([int(p["label_index"]) - 1 for p in ts], dtype=np.int32)      # Julia counts from 1
        train = EncodedTimeSeriesSet(phi, labels, label_index, np.asarray(td["original_data"]),
                                     np.asarray(td["class_distribution"]))
    else:
        train = EncodedTimeSeriesSet.empty()
    return TrainedMPS(W, opts, train)
