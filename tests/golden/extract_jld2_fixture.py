#!/usr/bin/env python3
"""Extract golden vectors from the one fixture of the reference that holds real reference-produced
tensors: /root/reference/test/Data/ecg200/mps_saves/test_dataset.jld2 - a serialised
``MPSTime.TrainedMPS`` (ITensors MPS trained on ECG200 with the default MPSOptions: d=5, chi_max=25,
Legendre_No_Norm) together with its ``EncodedTimeSeriesSet`` (class-sorted original data and the
encoded product states).  The file is not referenced by any test of the reference snapshot
(SURVEY.md section 4), but it is reference OUTPUT, so it pins parts of the oracle that nothing else can:
preprocessing + Legendre encoding (original_data -> pstate), the MPS container conventions
(index order, label on the last site, canonical form after fitMPS) and contract_mps / classify.

No HDF5 library is installed, so this is a minimal reader of exactly what JLD2 0.x wrote here
(HDF5 superblock v2, version-2 object headers, contiguous/compact layouts).  Output:
tests/golden/ref_ecg200_trained_mps.npz (data only; no reference source text).

Run (build container only - /root/reference does not exist on the GPU box):
    python tests/golden/extract_jld2_fixture.py
"""
import os
import re
import struct

import numpy as np

SRC = "/root/reference/test/Data/ecg200/mps_saves/test_dataset.jld2"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    f = open(SRC, "rb").read()
    assert f[512:520] == b"\x89HDF\r\n\x1a\n" and f[520] == 2, "expected an HDF5 v2 superblock at offset 512"
    base = struct.unpack("<Q", f[524:532])[0]          # base address: every file address is relative to it

    def messages(pos):
        """(type, data) of every header message of the version-2 object header at `pos`."""
        assert f[pos:pos + 4] == b"OHDR" and f[pos + 4] == 2
        flags = f[pos + 5]
        p = pos + 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
        szb = 1 << (flags & 3)
        size = int.from_bytes(f[p:p + szb], "little")
        p += szb
        out = []

        def walk(p, end):
            while p + 4 <= end:
                t, sz = f[p], struct.unpack("<H", f[p + 1:p + 3])[0]
                p += 4 + (2 if flags & 0x04 else 0)
                data = f[p:p + sz]
                if t == 0x10:                               # continuation block
                    off, ln = struct.unpack("<QQ", data[:16])
                    if f[base + off:base + off + 4] == b"OCHK":
                        walk(base + off + 4, base + off + ln - 4)
                else:
                    out.append((t, data))
                p += sz
        walk(p, p + size)
        return out

    objs = []
    for m in re.finditer(b"OHDR", f):
        try:
            msgs = messages(m.start())
        except Exception:
            continue
        ds = dt = lay = None
        for t, d in msgs:
            ds, dt, lay = (d if t == 1 else ds), (d if t == 3 else dt), (d if t == 8 else lay)
        if ds is None or dt is None or lay is None or ds[0] != 2:
            continue
        rank = ds[1]
        dims = struct.unpack("<%dQ" % rank, ds[4:4 + 8 * rank])
        objs.append(dict(pos=m.start(), cls=dt[0] & 0x0F, size=struct.unpack("<I", dt[4:8])[0], dims=dims, lay=lay))

    def contiguous(o, dtype):
        assert o["lay"][1] == 1
        addr = struct.unpack("<Q", o["lay"][2:10])[0] + base
        n = int(np.prod(o["dims"]))
        return np.frombuffer(f[addr:addr + 8 * n], dtype=dtype)

    f64_1d = [o for o in objs if o["cls"] == 1 and o["size"] == 8 and len(o["dims"]) == 1]
    f64_2d = [o for o in objs if o["cls"] == 1 and o["size"] == 8 and len(o["dims"]) == 2]
    i64_1d = [o for o in objs if o["cls"] == 0 and o["size"] == 8 and len(o["dims"]) == 1]
    # ---- tensors: compact objects holding [storage ref][Index]*k, Index = id u64, dim i64, dir i32,
    #      tags 4 x UInt256 (UTF-16 chars packed from the top), ntags i64, plev i64  (164 bytes)
    tensors = []
    for o in objs:
        if o["lay"][1] != 0:
            continue
        nb = struct.unpack("<H", o["lay"][2:4])[0]
        if nb not in (336, 500):
            continue
        dd = o["lay"][4:4 + nb]
        inds = []
        for k in range((nb - 8) // 164):
            b = 8 + 164 * k
            idv, dim = struct.unpack("<Qq", dd[b:b + 16])
            tag0 = dd[b + 20:b + 52]
            name = bytes(c for c in tag0[::-1] if c).decode()           # "Site" / "Link" / "f(x)"
            inds.append((idv, dim, name))
        tensors.append(inds)
    T = len(tensors)
    assert T == 96 and len(f64_1d) == T + 100 * T
    d = 5
    # site order = file order; check through the shared link ids
    chi = [1]
    W = []
    for j, inds in enumerate(tensors):
        names = [n for _, _, n in inds]
        dims = [dm for _, dm, _ in inds]
        if j + 1 < T:
            shared = {i for i, _, _ in inds} & {i for i, _, _ in tensors[j + 1]}
            assert len(shared) == 1
        v = contiguous(f64_1d[j], "<f8")
        t = v.reshape(dims, order="F")                      # Julia column-major array in the stored index order
        if j == 0:
            assert names == ["Site", "Link"]
            t = t.reshape(d, 1, dims[1]).transpose(1, 0, 2)               # -> (Dl, d, Dr)
        elif j < T - 1:
            assert names == ["Site", "Link", "Link"] and inds[1][0] in {i for i, _, _ in tensors[j - 1]}
            t = t.transpose(1, 0, 2)
        else:
            assert names == ["Site", "f(x)", "Link"]
            t = t.transpose(2, 0, 1).reshape(dims[2], d, 1, dims[1])      # -> (Dl, d, 1, C)
        W.append(np.ascontiguousarray(t))
        chi.append(W[-1].shape[2])
    pstates = np.stack([contiguous(o, "<f8") for o in f64_1d[T:]]).reshape(100, T, d)
    mats = [contiguous(o, "<f8").reshape(o["dims"]).T for o in f64_2d]      # Julia (N, T) matrices
    original = mats[0]                                                      # EncodedTimeSeriesSet.original_data
    ints = [contiguous(o, "<i8") for o in i64_1d]
    class_distribution = [a for a in ints if a.size == 2][0]
    out = dict(original_data=original, pstates=pstates, class_distribution=class_distribution,
               chi=np.array(chi), d=d, chi_max=25)
    for j, t in enumerate(W):
        out[f"W_{j}"] = t
    path = os.path.join(HERE, "ref_ecg200_trained_mps.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; chi =", chi[:4], "...", chi[-3:], "classes", class_distribution)


if __name__ == "__main__":
    main()
