#!/usr/bin/env python3
"""tools/profc/covmap.py <counted code object (ELF, amdgcn)> <out.json>

Reads what clang's -fprofile-instr-generate -fcoverage-mapping left in the code object -- the counter section layout
(__llvm_prf_cnts / __llvm_prf_data / __llvm_prf_names) and the coverage mapping (__llvm_covfun) -- and writes, per instrumented
function, where its counters sit in the counter section and which source regions each counter (or counter expression) covers.
The toolchain here ships neither llvm-cov nor llvm-profdata, so the two formats are decoded by hand:
  __llvm_prf_data : records of 64 bytes {NameRef u64, FuncHash u64, RelativeCounterPtr i64, RelativeBitmapPtr, FunctionPointer,
                    Values, NumCounters u32, NumValueSites u16[3], NumBitmapBytes u32}
  __llvm_covfun   : records {NameRef u64, DataSize u32, FuncHash u64, FilenamesRef u64, mapping[DataSize]}, 8-byte aligned;
                    mapping = file-id table, expression table, per file id a list of regions (counter, line/column deltas)
(llvm/ProfileData/Coverage/CoverageMappingReader.cpp is the format's definition.)"""
import hashlib
import json
import re
import struct
import subprocess
import sys
import tempfile
import zlib

LL = "/opt/rocm/lib/llvm/bin/"


def sections(elf):
    out = subprocess.check_output([LL + "llvm-readelf", "-S", "-W", elf], text=True)
    sec = {}
    for l in out.split("\n"):
        m = re.match(r"\s*\[\s*\d+\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", l)
        if m:
            sec[m.group(1)] = (int(m.group(2), 16), int(m.group(3), 16), int(m.group(4), 16))     # addr, file offset, size
    return sec


def uleb(b, p):
    v = 0
    s = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 0x7f) << s
        s += 7
        if not c & 0x80:
            return v, p


def counter(enc):
    tag = enc & 3
    return ["z", 0] if tag == 0 else ["c", enc >> 2] if tag == 1 else ["-", enc >> 2] if tag == 2 else ["+", enc >> 2]


def decode_mapping(b):
    p = 0
    nf, p = uleb(b, p)
    files = []
    for _ in range(nf):
        v, p = uleb(b, p)
        files.append(v)
    ne, p = uleb(b, p)
    exprs = []
    for _ in range(ne):
        l, p = uleb(b, p)
        r, p = uleb(b, p)
        exprs.append([counter(l), counter(r)])
    regions = []
    for fid in range(nf):
        nr, p = uleb(b, p)
        line = 0
        for _ in range(nr):
            enc, p = uleb(b, p)
            kind, c1, c2, exp = "code", counter(enc), None, None
            if (enc & 3) == 0 and (enc >> 2) != 0:
                if enc & 4:
                    kind, exp = "expansion", enc >> 3
                else:
                    k = enc >> 3
                    if k == 2:
                        kind = "skipped"
                    elif k == 4:
                        kind = "branch"
                        a, p = uleb(b, p); c1 = counter(a)
                        a, p = uleb(b, p); c2 = counter(a)
                    elif k == 6:
                        kind = "mcdc_branch"
                        a, p = uleb(b, p); c1 = counter(a)
                        a, p = uleb(b, p); c2 = counter(a)
                        for _i in range(3):
                            _, p = uleb(b, p)
                    elif k == 5:
                        kind = "mcdc_decision"
                        for _i in range(2):
                            _, p = uleb(b, p)
                    else:
                        kind = "pseudo%d" % k
            dl, p = uleb(b, p)
            cs, p = uleb(b, p)
            nl, p = uleb(b, p)
            ce, p = uleb(b, p)
            line += dl
            gap = bool(ce & (1 << 31))
            ce &= ~(1 << 31)
            if gap:
                kind = "gap"
            regions.append({"file": fid, "kind": kind, "c": c1, "c2": c2, "exp": exp, "ls": line, "cs": cs, "le": line + nl, "ce": ce})
    return files, exprs, regions


def main():
    elf, outp = sys.argv[1], sys.argv[2]
    sec = sections(elf)
    raw = open(elf, "rb").read()

    def sbytes(name):
        a, off, sz = sec[name]
        return raw[off:off + sz]

    def dumped(name):       # non-alloc sections (covfun / covmap) sit in the file as well
        return sbytes(name)
    cnts_addr, _, cnts_size = sec["__llvm_prf_cnts"]
    data_addr, _, _ = sec["__llvm_prf_data"]
    # names: [uleb uncompressed size][uleb compressed size][bytes], names joined by \x01; several such blocks may follow each other
    nb = sbytes("__llvm_prf_names")
    names = []
    p = 0
    while p < len(nb):
        if nb[p] == 0:
            p += 1
            continue
        us, p = uleb(nb, p)
        cs_, p = uleb(nb, p)
        if cs_:
            blob = zlib.decompress(nb[p:p + cs_]); p += cs_
        else:
            blob = nb[p:p + us]; p += us
        names += blob.decode("latin1").split("\x01")
    by_hash = {struct.unpack("<Q", hashlib.md5(n.encode("latin1")).digest()[:8])[0]: n for n in names}
    # data records
    db = sbytes("__llvm_prf_data")
    funcs = {}
    for i in range(0, len(db), 64):
        nameref, fhash, relc = struct.unpack_from("<QQq", db, i)
        ncnt = struct.unpack_from("<I", db, i + 48)[0]
        off = (data_addr + i + relc) - cnts_addr
        funcs[nameref] = {"name": by_hash.get(nameref, "?%016x" % nameref), "hash": fhash, "cnt_off": off // 8, "ncnt": ncnt, "cov": []}
        assert 0 <= off < cnts_size and off % 8 == 0, (hex(nameref), off)
    # coverage records
    cb = dumped("__llvm_covfun")
    p = 0
    nrec = 0
    while p + 28 <= len(cb):
        nameref, dsz, fhash, fref = struct.unpack_from("<QIQQ", cb, p)
        body = cb[p + 28:p + 28 + dsz]
        p = (p + 28 + dsz + 7) & ~7
        if dsz == 0:
            continue
        nrec += 1
        files, exprs, regions = decode_mapping(body)
        f = funcs.get(nameref)
        if f is None:
            continue
        f["cov"].append({"files": files, "exprs": exprs, "regions": regions, "fhash": fhash})
    # filenames of the translation unit (__llvm_covmap: header {u32 0, u32 FilenamesSize, u32 0, u32 Version}, then
    # [n][uncompressed len][compressed len][zlib or plain: n x (len, bytes)]); a function's `files` index into this list
    mb = dumped("__llvm_covmap")
    fsz = struct.unpack_from("<I", mb, 4)[0]
    fb = mb[16:16 + fsz]
    q = 0
    nfn, q = uleb(fb, q)
    ulen, q = uleb(fb, q)
    clen, q = uleb(fb, q)
    blob = zlib.decompress(fb[q:q + clen]) if clen else fb[q:q + ulen]
    filenames = []
    q = 0
    for _ in range(nfn):
        ln, q = uleb(blob, q)
        filenames.append(blob[q:q + ln].decode("latin1")); q += ln
    # the anchor: offset of cn_profc_kernel's own counters
    anchor = [f for f in funcs.values() if f["name"] == "cn_profc_kernel"]
    out = {"filenames": filenames, "cnts_u64": cnts_size // 8, "anchor_before_bytes": anchor[0]["cnt_off"] * 8 if anchor else None,
           "functions": sorted(funcs.values(), key=lambda f: f["cnt_off"])}
    json.dump(out, open(outp, "w"))
    print("covmap: %d functions with counters (%d counters), %d coverage records, anchor at byte %s" % (
        len(funcs), cnts_size // 8, nrec, out["anchor_before_bytes"]))


if __name__ == "__main__":
    main()
