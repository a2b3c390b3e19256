"""BAM alignment records -> the read stream (BAMHitFactory::getHitFromBuf, /root/reference/src/read.cpp:480-715), CPU side:
the oracle against what the REFERENCE made of the committed records (tests/golden/bamdecode_cases.npz, written by
tools/make_bamdecode_golden.py through the reference's own BAMHitFactory), against the reference itself where oracle/_ref
is built, and the library's host entry against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import bam_util as B
from conftest import GOLDEN
from make_bamdecode_golden import SETTINGS


def against_reference(o, z, pre=""):
    """oracle output `o` (per record) against the reference's answers `z` (dict, keys prefixed)."""
    g = lambda k: z[pre + k]
    acc = o["status"] == 0
    np.testing.assert_array_equal(acc, g("accepted") == 1)
    a = np.flatnonzero(acc)
    for k in ("read_id", "ref", "left", "right", "strand", "partner_same_ref", "partner_pos", "nm", "nh", "read_len"):
        np.testing.assert_array_equal(o[k][a], g(k)[a], err_msg=k)
    np.testing.assert_array_equal(o["mass"][a], g("mass")[a])                                  # 1 / NH or 0.5 / NH, bit for bit
    fb = g("flag_bits")[a]
    np.testing.assert_array_equal(o["singleton"][a], (fb >> 31).astype(np.uint8))
    np.testing.assert_array_equal(o["sam_flag"][a] & (16 | 64 | 128), fb & (16 | 64 | 128))   # the bits ReadHit shows
    for k in ("cig_off", "cig_type", "cig_len", "feat_off", "feat_code", "feat_left", "feat_right"):
        np.testing.assert_array_equal(o[k], g(k), err_msg=k)
    assert o["any_paired"] == 1 - int(g("single_end"))


@pytest.mark.parametrize("name,kw", SETTINGS)
def test_oracle_equals_reference_goldens(oracle, name, kw):
    z = np.load(os.path.join(GOLDEN, "bamdecode_cases.npz"))
    o = oracle.bam_decode(z["rec_bytes"], n_ref=int(z["n_ref"]), **kw)
    assert o["n"] == 1500
    against_reference(o, z, name + "/")
    # every way of refusing a record occurs among them (TRUNCATED and BAD_REF are not the reference's: see below)
    if name == "introns_5_3000_multi":
        assert set(np.unique(o["status"])) >= {0, 1, 3, 4, 5, 6, 7, 8}


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_equals_live_reference(oracle, reflib, tmp_path, seed, capfd):
    """Fresh random records through the reference's own BAMHitFactory (oracle/_ref) and through the oracle."""
    rng = np.random.default_rng(900 + seed)
    recs = B.random_records(rng, 2500)
    path = str(tmp_path / "t.bam")
    B.write_bam(path, B.REFS, recs)
    refs, raw = B.read_bam_records(path)
    for kw in (dict(), dict(unique_only=False, library=2), dict(min_intron=30, max_intron=1000, library=1)):
        z = reflib.bam_decode(path, len(recs) + 8, raw.size // 4 + 8, **kw)
        o = oracle.bam_decode(raw, n_ref=len(refs), **kw)
        against_reference(o, z)
    capfd.readouterr()   # (the reference reports zero-length operations on stderr)


def check_library_against_oracle(d, o):
    np.testing.assert_array_equal(d.status, o["status"])
    a = np.flatnonzero(o["status"] == 0)
    np.testing.assert_array_equal(d.record, a)
    for k in ("read_id", "ref", "left", "right", "partner_pos", "nm", "nh", "read_len", "sam_flag"):
        np.testing.assert_array_equal(getattr(d, k), o[k][a], err_msg=k)
    flags = ((o["sam_flag"][a] >> 4) & 1) | ((1 - o["partner_same_ref"][a].astype(np.uint32)) << 1) | (o["strand"][a].astype(np.uint32) << 2)
    np.testing.assert_array_equal(d.flags, flags.astype(np.uint8))
    m = o["feat_code"] == 0                              # the blocks are readhit_2_genomicFeats' MATCH features
    np.testing.assert_array_equal(d.block_left, o["feat_left"][m])
    np.testing.assert_array_equal(d.block_right, o["feat_right"][m])
    per_read = np.diff(np.concatenate([[0], np.cumsum(m)])[o["feat_off"]])[a]
    np.testing.assert_array_equal(np.diff(d.block_off), per_read)
    assert d.any_paired == bool(o["any_paired"])
    assert d.n_reads == a.size and sum(d.by_status.values()) == d.n_records
    for k, name in enumerate(("OK", "UNMAPPED", "BAD_REF", "ZERO_OP", "OP", "INTRON_LONG", "INTRON_SHORT", "INDEL", "SHORT", "MULTI", "TRUNCATED")):
        assert d.by_status[name] == int((o["status"] == k).sum()), name


@pytest.mark.parametrize("name,kw", SETTINGS)
def test_host_decode_equals_oracle_on_the_goldens(oracle, name, kw):
    from strawberry_amd import bam
    z = np.load(os.path.join(GOLDEN, "bamdecode_cases.npz"))
    raw = z["rec_bytes"]
    off = bam.index(raw)
    np.testing.assert_array_equal(off, oracle.bam_index(raw))
    d = bam.decode(raw, off, bam.BamOptions(n_ref=int(z["n_ref"]), **kw))
    check_library_against_oracle(d, oracle.bam_decode(raw, n_ref=int(z["n_ref"]), **kw))
    d.close()


def test_host_decode_edge_cases(oracle):
    """No records; a reference id beyond the header; records whose own lengths do not fit their size word; a stream cut
    inside a record; a `B` tag with a negative count."""
    import struct
    from strawberry_amd import _lib, bam
    d = bam.decode(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    assert d.n_records == 0 and d.n_reads == 0 and d.block_off.tolist() == [0]
    good = B.record(1, 100, 0, "ok", [("M", 50)], tags=[("NH", "C", 1)])
    far_ref = B.record(7, 100, 0, "far", [("M", 50)])
    lying = bytearray(B.record(0, 5, 0, "liar", [("M", 30)]))
    lying[16:20] = struct.pack("<I", (0 << 16) | 4000)              # claims 4000 CIGAR operations
    tiny = struct.pack("<i", 8) + b"\0" * 8                         # a "record" shorter than the fixed part
    neg_b = B.record(0, 9, 0, "negB", [("M", 40)], tags=[("XB", "B", ("i", [1]))])
    neg_b = bytearray(neg_b)
    neg_b[-8:-4] = struct.pack("<i", -5)                            # the array's count
    neg_b += b""                                                    # (NH, if any, would sit behind it: none here)
    raw = np.frombuffer(good + far_ref + bytes(lying) + tiny + bytes(neg_b) + good, np.uint8)
    off = bam.index(raw)
    assert off.size == 7
    o = oracle.bam_decode(raw, n_ref=3)
    d = bam.decode(raw, off, bam.BamOptions(n_ref=3))
    check_library_against_oracle(d, o)
    assert d.status.tolist() == [0, 2, 10, 10, 0, 0]
    with pytest.raises(_lib.SbgpuError):
        bam.index(raw[:-5])                                         # the stream ends inside the last record
    # without the header's reference count the far reference is taken as it is
    assert bam.decode(raw, off, bam.BamOptions(n_ref=0)).status.tolist() == [0, 0, 10, 10, 0, 0]


def test_indel_position_quirk_and_blocks():
    """The reference refuses an insertion / deletion among the first two kept operations (`i-1 <= 0`, read.cpp:594): a soft
    clip in front makes the same alignment acceptable; hard clips and pads do not count as kept.  A deletion extends the
    block in front of it, an insertion leaves two blocks that touch."""
    from strawberry_amd import bam
    cases = [([("M", 10), ("I", 2), ("M", 10)], 7, None),
             ([("S", 3), ("M", 10), ("I", 2), ("M", 10)], 0, [(101, 110), (111, 120)]),
             ([("H", 3), ("M", 10), ("I", 2), ("M", 10)], 7, None),
             ([("S", 3), ("M", 10), ("D", 4), ("M", 10)], 0, [(101, 114), (115, 124)]),
             ([("M", 10), ("N", 100), ("M", 5), ("D", 1), ("M", 5), ("N", 50), ("M", 8)], 0, [(101, 110), (211, 216), (217, 221), (272, 279)]),
             ([("M", 10), ("N", 100), ("M", 5), ("I", 1), ("P", 2), ("M", 5)], 0, [(101, 110), (211, 215), (216, 220)]),
             ([("M", 10), ("N", 100), ("I", 1), ("M", 5)], 7, None),
             ([("M", 10), ("N", 100), ("M", 5), ("D", 1)], 7, None),
             ([("M", 10), ("N", 100), ("M", 5), ("D", 1), ("S", 4)], 7, None),
             ([("M", 1)], 8, None), ([("M", 2)], 0, [(101, 102)]), ([("M", 1), ("N", 30), ("M", 1)], 0, [(101, 101), (132, 132)])]
    raw = np.frombuffer(b"".join(B.record(0, 100, 0, "q%d" % k, c) for k, (c, _, _) in enumerate(cases)), np.uint8)
    d = bam.decode(raw)
    assert d.status.tolist() == [s for _, s, _ in cases]
    want = [b for _, s, b in cases if s == 0]
    got = [list(zip(d.block_left[d.block_off[k]:d.block_off[k + 1]].tolist(), d.block_right[d.block_off[k]:d.block_off[k + 1]].tolist()))
           for k in range(d.n_reads)]
    assert got == want
    np.testing.assert_array_equal(d.right, [b[-1][1] for b in want])


def test_touching_blocks_make_no_intron_like_the_reference(reflib):
    """An insertion leaves two aligned blocks that touch; readhit_2_genomicFeats (src/contig.cpp:12-53) then makes two MATCH
    features side by side, Contig(PairedHit) keeps them apart when the mates do not overlap and refuses the pair when they do
    (merge_genomicFeats, include/contig.h:111-137).  The library's feature lists from blocks (host pairing) and its
    Contig(PairedHit) restatement (sbgpu_hit_features) against the reference's own classes."""
    from strawberry_amd import exonbin as eb
    left = [(100, 109), (110, 130), (400, 420)]          # 10M 2I 21M 269N 21M
    right = [(600, 640), (641, 660)]                      # 41M 1I 20M
    assert eb.mate_features(left) == ([0, 0, 1, 0], [100, 110, 131, 400], [109, 130, 399, 420])
    for lb, rb in ((left, right), (left, []), ([], right), (left, [(415, 450), (451, 470)]), (left, [(405, 430)])):
        want = reflib.pairedhit_features(lb, rb)
        assert eb.hit_features(lb, rb) == want, (lb, rb)
    assert reflib.pairedhit_features(left, right)[0] == [0, 0, 1, 0, 2, 0, 0]
    assert reflib.pairedhit_features(left, [(415, 450), (451, 470)]) is None     # overlapping mates, blocks that only touch
    # the host pairing writes the mates' features that way
    reads = eb.Reads([0, 0], [7, 7], [left, right], [600, 100], [1 << 2, 1 | (1 << 2)], [1, 1])
    got = eb.pair_mates(1, reads)
    assert got["info"]["complete"] == 1
    assert [x.tolist() for x in got["left"]] == [list(x) for x in eb.mate_features(left)]
    assert [x.tolist() for x in got["right"]] == [list(x) for x in eb.mate_features(right)]


@pytest.mark.parametrize("which", ["E2E", "E2E_MASS", "E2E_MINUS", "E2E_CHROMS"])
def test_bam_bytes_to_unique_hits_on_the_host_equal_reference_runs(which):
    """The read pairs of the reference binary's toy runs as BAM records, in the file's order -> sbgpu_bam_decode_host ->
    sbgpu_assign_reads_host -> sbgpu_pair_mates_host -> sbgpu_collapse_pairs_host: the unique hits (features, masses) and the
    mapped-read total of the reference run (tests/test_bamdecode_gpu.py does the same on the device)."""
    import ctypes as C
    import e2e_util as U
    import exonbin_util as XU
    from strawberry_amd import _lib, bam, exonbin as eb
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    raw, chrom_names, (c_ref, c_left, c_right, c_strand) = B.toy_run_as_bam_records(d, names)
    dec = bam.decode(raw, None, bam.BamOptions(unique_only=which != "E2E_MASS", n_ref=len(chrom_names)))   # (the mass run: --allow-multimapped-hits)
    assert dec.n_reads == dec.n_records and dec.any_paired
    got_cluster, off, fl = eb.assign_reads(c_ref, c_left, c_right, c_strand, dec.ref, dec.left, dec.right, dec.flags)
    L = _lib.load()
    fl = np.ascontiguousarray(fl, np.uint8)
    rs = _lib.sbgpu_reads_t(dec.n_reads, dec.read_id.ctypes.data, dec.block_off.ctypes.data, dec.block_left.ctypes.data,
                            dec.block_right.ctypes.data, dec.partner_pos.ctypes.data, fl.ctypes.data, dec.nh.ctypes.data)
    mh, uh, dp, poff = C.c_void_p(), C.c_void_p(), _lib.sbgpu_pairs_t(), C.c_void_p()
    _lib.check(L.sbgpu_pair_mates_host(len(names), C.byref(rs), off.ctypes.data, C.byref(mh)), "sbgpu_pair_mates_host")
    _lib.check(L.sbgpu_matepairs_pairs(mh, C.byref(dp), C.byref(poff)), "sbgpu_matepairs_pairs")
    _lib.check(L.sbgpu_collapse_pairs_host(len(names), C.byref(dp), C.byref(uh)), "sbgpu_collapse_pairs_host")
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_uniq_info(uh, info), "sbgpu_uniq_info")
    nh_, nf_ = int(info[0]), int(info[1])
    hl, fo = np.zeros(nh_, np.int32), np.zeros(nh_ + 1, np.int64)
    fc, fl_, fr, ms = np.zeros(nf_, np.uint8), np.zeros(nf_, np.uint32), np.zeros(nf_, np.uint32), np.zeros(nh_, np.float32)
    cm = np.zeros(len(names))
    _lib.check(L.sbgpu_uniq_export(uh, hl.ctypes.data, fo.ctypes.data, fc.ctypes.data, fl_.ctypes.data, fr.ctypes.data, ms.ctypes.data,
                                   cm.ctypes.data), "sbgpu_uniq_export")
    L.sbgpu_uniq_destroy(uh)
    L.sbgpu_matepairs_destroy(mh)
    np.testing.assert_array_equal(hl, hits.hit_locus)
    np.testing.assert_array_equal(fo, hits.feat_off)
    np.testing.assert_array_equal(fc, hits.feat_code)
    np.testing.assert_array_equal(fl_, hits.feat_left)
    np.testing.assert_array_equal(fr, hits.feat_right)
    np.testing.assert_array_equal(ms, hits.mass)
    assert int(info[4]) == rows[0]["total_mapped"] == hits.total_mapped


def test_host_decode_on_garbage_records(oracle):
    """Records whose bytes are noise behind a valid size word (and some with a plausible head and noise for tags): nothing may
    be read outside a record, and the library must take the same way through the noise as the oracle."""
    from strawberry_amd import bam
    recs = B.garbage_records(np.random.default_rng(31), 4000)
    raw = np.frombuffer(b"".join(recs), np.uint8)
    off = bam.index(raw)
    assert off.size - 1 == len(recs)
    for kw in (dict(), dict(unique_only=False, library=2)):
        d = bam.decode(raw, off, bam.BamOptions(n_ref=3, **kw))
        o = oracle.bam_decode(raw, off, n_ref=3, **kw)
        check_library_against_oracle(d, o)
        assert d.by_status["OK"] > 50 and d.by_status["TRUNCATED"] > 500


def test_host_decode_on_several_threads(oracle, monkeypatch):
    """sbgpu_bam_decode_host splits the records over host threads (count, then write): the same arrays whatever their number."""
    from strawberry_amd import bam
    recs = B.random_records(np.random.default_rng(77), 5000)
    raw = np.frombuffer(b"".join(recs) * 6, np.uint8)
    off = bam.index(raw)
    o = oracle.bam_decode(raw, off, n_ref=3, unique_only=False)
    got = []
    for nt in ("1", "3", "7"):
        monkeypatch.setenv("SBGPU_HOST_THREADS", nt)
        d = bam.decode(raw, off, bam.BamOptions(n_ref=3, unique_only=False))
        check_library_against_oracle(d, o)
        got.append(d)
    for k in ("status", "record", "read_id", "left", "right", "flags", "block_off", "block_left", "block_right"):
        for d in got[1:]:
            np.testing.assert_array_equal(getattr(d, k), getattr(got[0], k), err_msg=k)


def test_names_and_long_cigars_against_the_live_reference(oracle, reflib, tmp_path, capfd):
    """Read names with bytes >= 0x80 (hashString xors a SIGNED char, include/read.hpp:164-173), names of 254 bytes and of none,
    CIGARs of hundreds of operations: the reference's own BAMHitFactory, the oracle, the library's host decoder."""
    import struct
    from strawberry_amd import bam
    rng = np.random.default_rng(5)
    recs = []
    for k in range(400):
        nm = bytes(rng.integers(1, 256, int(rng.choice([0, 1, 5, 40, 254])), dtype=np.uint8).tolist())
        n_ops = int(rng.choice([1, 3, 60, 400]))
        cig = []
        for j in range(n_ops):
            cig.append(("M", int(rng.integers(1, 30))))
            if j + 1 < n_ops:
                cig.append(("N", int(rng.integers(20, 200))))
        r = bytearray(B.record(int(rng.integers(0, 3)), int(rng.integers(0, 10 ** 6)), int(rng.choice([0, 16, 99, 147])), "x", cig,
                               mtid=int(rng.integers(-1, 3)), mpos=int(rng.integers(-1, 10 ** 6)), tags=[("NH", "C", 1), ("XS", "A", "+")], with_seq=False))
        # swap the one-letter name for the bytes wanted (the name's length byte and the record's size word follow)
        body = bytes(r[4:36]) + nm + b"\0" + bytes(r[36 + 2:])
        core = bytearray(body[:32])
        core[8:12] = struct.pack("<I", (struct.unpack("<I", core[8:12])[0] & ~0xff) | (len(nm) + 1))
        body = bytes(core) + body[32:]
        recs.append(struct.pack("<i", len(body)) + body)
    path = str(tmp_path / "names.bam")
    B.write_bam(path, B.REFS, recs)
    refs, raw = B.read_bam_records(path)
    z = reflib.bam_decode(path, len(recs) + 8, raw.size // 4 + 8)
    o = oracle.bam_decode(raw, n_ref=len(refs))
    against_reference(o, z)
    assert int((o["status"] == 0).sum()) > 380 and set(np.unique(o["status"])) <= {0, 8}   # (a lone 1M is too short)
    assert len(set(o["read_id"].tolist())) > 300          # (the empty names share one id)
    d = bam.decode(raw, None, bam.BamOptions(n_ref=len(refs)))
    check_library_against_oracle(d, o)
    capfd.readouterr()
