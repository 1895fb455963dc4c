"""Test helper: writes BAM + BAI + FASTA + .fai files from plain Python data, straight from the SAM/BAM specification
(SAMv1.pdf 4.1, 4.2, 5.2).  Independent of uvc_io.cpp, which only reads."""
import struct
import zlib

import numpy as np

NT16 = {0: 1, 1: 2, 2: 4, 3: 8, 4: 15}   # A C G T N codes -> 4-bit BAM encoding


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def bgzf_block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def record_bytes(tid, pos, qname, flag, mapq, cigar, bases, quals, mtid, mpos, tlen, nm=None, extra_aux=b""):
    """cigar: list of (op, len); bases: codes 0..4"""
    end = pos + sum(l for o, l in cigar if o in (0, 2, 3, 7, 8))
    if end == pos: end = pos + 1
    name = qname.encode() + b"\0"
    seq = bytearray((len(bases) + 1) // 2)
    for i, b in enumerate(bases):
        seq[i >> 1] |= NT16[int(b)] << (0 if i & 1 else 4)
    aux = extra_aux
    if nm is not None and nm >= 0:
        aux += b"NM" + (b"C" + struct.pack("<B", nm) if nm < 256 else b"i" + struct.pack("<i", nm))
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, reg2bin(pos, end), len(cigar), flag, len(bases), mtid, mpos, tlen) + name \
        + b"".join(struct.pack("<I", (l << 4) | o) for o, l in cigar) + bytes(seq) + bytes(int(q) for q in quals) + aux
    return struct.pack("<i", len(body)) + body, end


def write_bam(path, refs, records, block_bytes=30000, with_index=True, packed=False):
    """refs: [(name, length)]; records: dicts with tid, pos, qname, flag, mapq, cigar, bases, quals, mtid, mpos, tlen, nm -- sorted by (tid, pos)"""
    text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)).encode()
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)) + b"".join(struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln) for n, ln in refs)
    out = bytearray(bgzf_block(hdr))
    cur = bytearray()
    block_addr = len(out)
    index = [dict(bins={}, linear={}) for _ in refs]

    def flush():
        nonlocal cur, block_addr
        if cur:
            out.extend(bgzf_block(bytes(cur))); cur = bytearray(); block_addr = len(out)
    for r in records:
        b, end = record_bytes(r["tid"], r["pos"], r["qname"], r["flag"], r["mapq"], r["cigar"], r["bases"], r["quals"], r.get("mtid", -1), r.get("mpos", -1), r.get("tlen", 0), r.get("nm"), r.get("aux", b""))
        if not packed and len(cur) + len(b) > block_bytes: flush()   # htslib: a record never straddles two blocks (bgzf_flush_try)
        v0 = (block_addr << 16) | len(cur)
        cur.extend(b)
        while packed and len(cur) >= block_bytes:                        # packed: blocks of exactly block_bytes, records straddle them
            rest = cur[block_bytes:]; cur = cur[:block_bytes]; flush(); cur = rest
        v1 = (block_addr << 16) | len(cur)
        if r["tid"] >= 0:
            ix = index[r["tid"]]
            ix["bins"].setdefault(reg2bin(r["pos"], end), []).append([v0, v1])
            for w in range(r["pos"] >> 14, ((end - 1) >> 14) + 1):
                ix["linear"].setdefault(w, v0)
    flush()
    out.extend(bgzf_block(b""))   # EOF marker
    open(path, "wb").write(bytes(out))
    if with_index:
        bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
        for ix in index:
            bai += struct.pack("<i", len(ix["bins"]))
            for bn, chunks in sorted(ix["bins"].items()):
                merged = []
                for c in chunks:
                    if merged and c[0] <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], c[1])
                    else: merged.append(list(c))
                bai += struct.pack("<Ii", bn, len(merged)) + b"".join(struct.pack("<QQ", a, e) for a, e in merged)
            nw = (max(ix["linear"]) + 1) if ix["linear"] else 0
            lin, last = [], 0
            for w in range(nw):
                last = ix["linear"].get(w, last); lin.append(last)
            for w in range(nw - 2, -1, -1):   # empty windows take the offset of the next non-empty one (htslib fills backwards)
                if w not in ix["linear"]: lin[w] = lin[w + 1]
            bai += struct.pack("<i", nw) + b"".join(struct.pack("<Q", v) for v in lin)
        open(path + ".bai", "wb").write(bytes(bai))


def write_fasta(path, seqs, width=60):
    """seqs: [(name, text)]"""
    fai = []
    with open(path, "w") as fh:
        off = 0
        for name, s in seqs:
            head = ">%s\n" % name
            fh.write(head); off += len(head)
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(s), off, width, width + 1))
            for i in range(0, len(s), width):
                line = s[i:i + width] + "\n"
                fh.write(line); off += len(line)
    open(path + ".fai", "w").write("".join(fai))


def records_from_reads(reads, tid=0, qname_fmt="r%d", umis=None):
    """synth / fuzz read dict (UvcReadSoA columns) -> BAM records sorted by position; mates share the read name of their fragment"""
    recs = []
    for i in range(int(reads["n_reads"])):
        so, lq = int(reads["seq_off"][i]), int(reads["l_qseq"][i])
        co, nc = int(reads["cigar_off"][i]), int(reads["n_cigar"][i])
        frag = int(reads["frag_id"][i])
        name = qname_fmt % frag + (("#" + umis[int(reads["fam_id"][i])]) if umis is not None else "")
        fl = int(reads["flag"][i])
        recs.append(dict(tid=tid, pos=int(reads["pos"][i]), qname=name, flag=fl, mapq=int(reads["mapq"][i]),
                         cigar=[(int(c) & 0xF, int(c) >> 4) for c in reads["cigars"][co:co + nc]], bases=reads["bases"][so:so + lq], quals=reads["quals"][so:so + lq],
                         mtid=(tid if fl & 1 else -1), mpos=int(reads["mpos"][i]), tlen=int(reads["isize"][i]), nm=int(reads["nm"][i])))
    recs.sort(key=lambda r: r["pos"])
    return recs
