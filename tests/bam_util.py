"""BAM files made in Python for the record-decoder tests (sbgpu_bam_decode_*, oracle/bamdecode_oracle.c): raw alignment
records of any shape -- CIGAR operations, auxiliary tags of every type, flags -- packed as the SAM/BAM specification lays
them out, and BGZF-compressed so that the reference's own BAMHitFactory (samtools 0.1.19's bam_read1) reads the file.

  record(...)            one alignment record, block_size prefix included
  write_bam(path, ...)   header + records as a BGZF file
  read_bam_records(path) -> (refs, the uncompressed record bytes): what a caller hands to the decoder
  random_records(...)    a mixed bag of records that reaches every branch of the reference's getHitFromBuf
"""
import gzip
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=XB"
QUERY_OPS = set("MIS=X")


def _tag_bytes(tag, typ, val):
    t = tag.encode()
    if typ == "A":
        return t + b"A" + bytes([ord(val)])
    if typ in "cCsSiI":
        return t + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], val)
    if typ == "f":
        return t + b"f" + struct.pack("<f", val)
    if typ == "d":
        return t + b"d" + struct.pack("<d", val)
    if typ in "ZH":
        return t + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub, vals = val
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        return t + b"B" + sub.encode() + struct.pack("<i", len(vals)) + b"".join(struct.pack("<" + fmt, v) for v in vals)
    raise ValueError(typ)


def record(tid, pos, flag, name, cigar, mapq=30, mtid=-1, mpos=-1, tlen=0, tags=(), with_seq=True):
    """cigar: [(op char, length)]; pos / mpos 0-based as in the file; tags: [(two letters, type char, value)]."""
    qlen = sum(n for op, n in cigar if op in QUERY_OPS) if with_seq else 0
    qname = (name if isinstance(name, bytes) else name.encode()) + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in cigar)
    seq = bytes([0x12] * ((qlen + 1) // 2))
    qual = bytes([30] * qlen)
    aux = b"".join(_tag_bytes(*t) for t in tags)
    core = struct.pack("<iiIIiiii", tid, pos, (0 << 16) | (mapq << 8) | len(qname), (flag << 16) | len(cigar), qlen, mtid, mpos, tlen)
    body = core + qname + cig + seq + qual + aux
    return struct.pack("<i", len(body)) + body


def header_bytes(refs):
    """refs: [(name, length)] -> the BAM header (magic, text with one @SQ line per reference, binary reference list)."""
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    return out


def _bgzf_block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bgzf_compress(data, block=0xff00):
    out = b"".join(_bgzf_block(data[i:i + block]) for i in range(0, len(data), block))
    return out + _bgzf_block(b"")    # the EOF marker


def write_bam(path, refs, records):
    with open(path, "wb") as f:
        f.write(bgzf_compress(header_bytes(refs) + b"".join(records)))


def read_bam_records(path):
    """-> ([(name, length)], record bytes as a uint8 array): BGZF is a chain of gzip members."""
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", raw, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p)
        name = raw[p + 4:p + 4 + l_name - 1].decode()
        ln, = struct.unpack_from("<i", raw, p + 4 + l_name)
        refs.append((name, ln))
        p += 8 + l_name
    return refs, np.frombuffer(raw[p:], np.uint8).copy()


REFS = [("chr1", 5_000_000), ("chr2", 3_000_000), ("scaffold_3", 400_000)]


def random_records(rng, n, n_ref=len(REFS)):
    """Records that reach every branch of BAMHitFactory::getHitFromBuf (src/read.cpp:480-715): unmapped, every CIGAR
    operation (also of length zero, also the ones the reference refuses), introns inside / below / above the limits,
    insertions and deletions in every position, soft and hard clips, mates on the same / another / no reference,
    NH / NM / XS / ZF tags in every integer width, present or not, behind tags of the other types."""
    out = []
    for i in range(n):
        kind = rng.integers(0, 100)
        tid = int(rng.integers(0, n_ref))
        pos = int(rng.integers(0, 200_000))
        flag = 0
        paired = rng.random() < 0.8
        if paired:
            flag |= 1 | (0x40 if rng.random() < 0.5 else 0x80)
            if rng.random() < 0.6:
                flag |= 2
        if rng.random() < 0.5:
            flag |= 0x10
        if rng.random() < 0.05:
            flag |= 0x100
        if rng.random() < 0.05:
            flag |= 0x400
        if kind < 3:
            flag |= 4
        if kind == 3:
            tid = -1
        # CIGAR
        cig = []
        if rng.random() < 0.1:
            cig.append(("H", int(rng.integers(1, 20))))
        if rng.random() < 0.2:
            cig.append(("S", int(rng.integers(1, 12))))
        shape = rng.integers(0, 100)
        if shape < 35:
            cig.append(("M", int(rng.integers(1 if rng.random() < 0.1 else 20, 101))))
        else:
            for _ in range(int(rng.integers(1, 6))):
                cig.append(("M", int(rng.integers(1, 60))))
                r = rng.random()
                if r < 0.45:
                    lo, hi = (1, 40) if rng.random() < 0.2 else ((20, 5000) if rng.random() < 0.9 else (299_000, 302_000))
                    cig.append(("N", int(rng.integers(lo, hi))))
                elif r < 0.6:
                    cig.append(("I", int(rng.integers(1, 6))))
                elif r < 0.75:
                    cig.append(("D", int(rng.integers(1, 6))))
                elif r < 0.8:
                    cig.append(("P", int(rng.integers(1, 3))))
                elif r < 0.84:
                    cig.append((("=", "X")[int(rng.integers(0, 2))], int(rng.integers(1, 9))))
                elif r < 0.87:
                    cig.append(("S", int(rng.integers(1, 5))))
            if rng.random() < 0.75:
                cig.append(("M", int(rng.integers(1, 60))))
        if rng.random() < 0.03:
            k = int(rng.integers(0, len(cig)))
            cig[k] = (cig[k][0], 0)                      # a zero-length operation
        if rng.random() < 0.15:
            cig.append(("S", int(rng.integers(1, 12))))
        if rng.random() < 0.05:
            cig.append(("H", int(rng.integers(1, 20))))
        if rng.random() < 0.02:
            cig = []                                     # no CIGAR at all
        # the mate
        r = rng.random()
        if not paired or r < 0.1:
            mtid, mpos = -1, -1
        elif r < 0.85:
            mtid, mpos = tid, int(max(0, pos + rng.integers(-400, 400)))
        elif r < 0.95:
            mtid, mpos = int(rng.integers(0, n_ref)), int(rng.integers(0, 200_000))
        else:
            mtid, mpos = tid, -1
        # tags, in random order among fillers of the other types
        tags = []
        if rng.random() < 0.85:
            v = int(rng.choice([1, 1, 1, 1, 2, 3, 10, 300, 70000, 0]))
            typ = "C" if v < 256 and rng.random() < 0.7 else ("S" if v < 65536 and rng.random() < 0.5 else str(rng.choice(["i", "I"])))
            if v < 128 and rng.random() < 0.2:
                typ = "c"
            if v < 32768 and rng.random() < 0.1:
                typ = "s"
            tags.append(("NH", typ, v))
        if rng.random() < 0.7:
            tags.append(("NM", "C", int(rng.integers(0, 9))))
        if rng.random() < 0.6:
            tags.append(("XS", "A", str(rng.choice(["+", "-", "+", "-", "?", "."]))))
        if rng.random() < 0.05:
            tags.append(("XS", "Z", "+"))               # the wrong type for a strand: bam_aux2A gives 0
        if rng.random() < 0.1:
            tags.append(("ZF", "i", int(rng.integers(-3, 50))))
        fillers = [("MD", "Z", "10A5^AC6"), ("AS", "i", -17), ("XF", "f", 1.5), ("XH", "H", "1AE301"), ("XB", "B", ("s", [1, -2, 3])),
                   ("XC", "B", ("C", [])), ("YT", "Z", ""), ("XA", "A", "Q"), ("XI", "B", ("I", [7, 8])), ("Xf", "B", ("f", [0.5]))]
        for k in rng.permutation(len(fillers))[:int(rng.integers(0, 5))]:
            tags.append(fillers[int(k)])
        order = rng.permutation(len(tags))
        tags = [tags[int(k)] for k in order]
        if rng.random() < 0.02:
            # samtools 0.1.19 skips a double as if it had no payload (bam_aux.c:28-34, bam.h:772-778) and parses the payload's
            # bytes as tags.  Kept LAST: behind it that scan would wander through whatever follows, here it runs out.
            tags.append(("XD", "d", 2.5))
        name = "r%d:%s" % (i // 2 if paired else i, "x" * int(rng.integers(0, 25)))
        if rng.random() < 0.04:   # bytes >= 0x80 (the reference's hash xors a signed char), a name of 254 bytes, an empty one
            name = bytes(rng.integers(1, 256, int(rng.choice([0, 3, 30, 254])), dtype=np.uint8).tolist())
        out.append(record(tid, pos, flag, name, cig, mapq=int(rng.integers(0, 61)), mtid=mtid, mpos=mpos,
                          tlen=int(rng.integers(-500, 500)), tags=tags, with_seq=rng.random() < 0.9))
    return out


def toy_run_as_bam_records(d, names):
    """The read pairs of a toy run (reads.npz: what tools/make_e2e_golden.py wrote as SAM for the reference binary) as BAM
    records in the file's order -> (record bytes, reference names, clusters as sbgpu_assign_reads wants them)."""
    import os
    import e2e_util as U
    z = dict(np.load(os.path.join(d, "reads.npz")))
    ordered, _, _, _ = U.load(d)
    genes = list(U.parse_annotation(os.path.join(d, "toy.gtf")))
    strands, chroms = U.gene_strands(d), U.gene_chroms(d)
    chrom_names = sorted(set(chroms.values()))
    chrom_id = {c: i for i, c in enumerate(chrom_names)}
    recs = []
    serial = 0
    for k in range(len(z["gene"])):
        g = genes[int(z["gene"][k])]
        tid = chrom_id[chroms[g]]
        xs = "+" if strands[g] == "+" else "-"
        left = list(zip(z["left_l"][z["left_off"][k]:z["left_off"][k + 1]].tolist(), z["left_r"][z["left_off"][k]:z["left_off"][k + 1]].tolist()))
        right = list(zip(z["right_l"][z["right_off"][k]:z["right_off"][k + 1]].tolist(), z["right_r"][z["right_off"][k]:z["right_off"][k + 1]].tolist()))

        def cigar(blocks):
            out = []
            for i, (a, b) in enumerate(blocks):
                if i:
                    out.append(("N", a - blocks[i - 1][1] - 1))
                out.append(("M", b - a + 1))
            return out
        for nh in z["nh"][z["nh_off"][k]:z["nh_off"][k + 1]]:
            serial += 1
            name = "frag%d" % serial
            tags = [("NH", "C", int(nh)), ("XS", "A", xs)]
            recs.append((tid, left[0][0], record(tid, left[0][0] - 1, 1 | 2 | 0x20 | 0x40, name, cigar(left), mtid=tid, mpos=right[0][0] - 1, tags=tags)))
            recs.append((tid, right[0][0], record(tid, right[0][0] - 1, 1 | 2 | 0x10 | 0x80, name, cigar(right), mtid=tid, mpos=left[0][0] - 1, tags=tags)))
    recs.sort(key=lambda r: (r[0], r[1]))                      # the BAM's order: (reference, position), stable
    ref_of = [chrom_id[chroms[g]] for g in names]
    c_left = [min(e[0] for _, ex in ordered[g] for e in ex) for g in names]
    c_right = [max(e[1] for _, ex in ordered[g] for e in ex) for g in names]
    c_strand = [1 if strands[g] == "+" else 2 for g in names]
    return np.frombuffer(b"".join(r[2] for r in recs), np.uint8), chrom_names, (ref_of, c_left, c_right, c_strand)


def garbage_records(rng, n):
    """Records whose bytes are noise behind a valid size word; every other one with a head that passes the decoder's first
    tests (mapped, short name, 1-3 CIGAR operations, no sequence), so that the noise is walked as tags."""
    recs = []
    for k in range(n):
        size = int(rng.integers(0, 400))
        body = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        if k % 2 and size >= 32:
            n_cig = int(rng.integers(1, 4))
            l_qname = int(rng.integers(1, 9))
            if 32 + l_qname + 4 * n_cig <= size:
                core = struct.pack("<iiIIiiii", int(rng.integers(0, 3)), int(rng.integers(0, 1000)), (30 << 8) | l_qname,
                                   (int(rng.integers(0, 4)) << 20) | n_cig, 0, int(rng.integers(-1, 3)), int(rng.integers(-1, 1000)), 0)
                cig = b"".join(struct.pack("<I", (int(rng.integers(1, 90)) << 4) | int(rng.choice([0, 0, 0, 3, 4]))) for _ in range(n_cig))
                body = core + body[32:32 + l_qname] + cig + body[32 + l_qname + 4 * n_cig:]
        recs.append(struct.pack("<i", len(body)) + body)
    return recs
