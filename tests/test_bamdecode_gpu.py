"""GPU tests of sbgpu_bam_decode_device (BAMHitFactory::getHitFromBuf on the GPU, /root/reference/src/read.cpp:480-715,
csrc/bamdecode_device.h) against the oracle (oracle/bamdecode_oracle.c, pinned to the reference's own BAMHitFactory by
tests/test_bamdecode.py), and the front of the path from the file's BYTES: BAM records of the reference's toy runs ->
sbgpu_bam_decode_device -> sbgpu_assign_reads_device -> sbgpu_pair_mates_device -> sbgpu_collapse_pairs_device must give the
unique hits of the reference binary's runs, bit for bit, without the records leaving the device."""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import bam_util as B
import e2e_util as U
import exonbin_util as XU
import matepair_util as MU
from conftest import GOLDEN
from make_bamdecode_golden import SETTINGS
from test_bamdecode import check_library_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


@pytest.mark.parametrize("name,kw", SETTINGS)
def test_device_decode_equals_oracle_on_the_goldens(ctx, oracle, name, kw):
    from strawberry_amd import bam
    z = np.load(os.path.join(GOLDEN, "bamdecode_cases.npz"))
    raw = z["rec_bytes"]
    d = bam.decode(raw, None, bam.BamOptions(n_ref=int(z["n_ref"]), **kw), device=ctx)
    assert d.on_device
    check_library_against_oracle(d, oracle.bam_decode(raw, n_ref=int(z["n_ref"]), **kw))
    d.close()


def test_device_decode_large_stream_and_edges(ctx, oracle):
    """4 x 10^5 records (a random bag of 50 000, eight times over, with the edge records of the host test mixed in)."""
    from strawberry_amd import bam
    rng = np.random.default_rng(4)
    recs = B.random_records(rng, 50000)
    lying = bytearray(B.record(0, 5, 0, "liar", [("M", 30)]))
    lying[16:20] = struct.pack("<I", (0 << 16) | 4000)
    tiny = struct.pack("<i", 8) + b"\0" * 8
    recs[100:100] = [bytes(lying), tiny, B.record(9, 1, 0, "far", [("M", 40)])]
    raw = np.frombuffer(b"".join(recs) * 8, np.uint8)
    off = bam.index(raw)
    assert off.size - 1 == 8 * len(recs)
    for kw in (dict(), dict(unique_only=False, library=1, min_intron=10, max_intron=4000)):
        o = oracle.bam_decode(raw, off, n_ref=3, **kw)
        d = bam.decode(raw, off, bam.BamOptions(n_ref=3, **kw), device=ctx)
        check_library_against_oracle(d, o)
        h = bam.decode(raw, off, bam.BamOptions(n_ref=3, **kw))           # the host entry: the same decoder body
        for k in ("status", "record", "read_id", "ref", "left", "right", "partner_pos", "flags", "nh", "nm", "read_len", "sam_flag",
                  "block_off", "block_left", "block_right"):
            np.testing.assert_array_equal(getattr(d, k), getattr(h, k), err_msg=k)
        assert d.by_status["TRUNCATED"] == 16 and d.by_status["BAD_REF"] == 8
        d.close(), h.close()
    e = bam.decode(np.zeros(0, np.uint8), np.zeros(1, np.int64), device=ctx)
    assert e.n_records == 0 and e.n_reads == 0


def test_device_decode_fallback_routes(ctx, oracle, monkeypatch):
    """sbgpu_bam_decode_device decodes in ONE pass (bam_onepass_kernel: tiles of 1 024 records, a decoupled look-back for a
    tile's place) and keeps the two-kernel form for what the one pass declines: here a stream whose accepted records hold MORE
    than two blocks each on average (the one pass' arrays have room for 2 n + 1 024 blocks), and the hook that forces the
    two kernels (SBGPU_BAM_TWO_PASS=1) on an ordinary stream -- all equal to the oracle, and the two forms equal to each
    other array by array."""
    from strawberry_amd import bam
    rng = np.random.default_rng(41)
    many = [B.record(0, 10 + 7 * i, 0, "r%d" % i, [("M", 10), ("N", 100), ("M", 10), ("N", 120), ("M", 10), ("N", 90), ("M", 12)])
            for i in range(5000)]
    raw = np.frombuffer(b"".join(many), np.uint8)
    o = oracle.bam_decode(raw, n_ref=3)
    d = bam.decode(raw, None, bam.BamOptions(n_ref=3), device=ctx)
    assert d.n_reads == 5000 and d.n_blocks == 20000          # 4 blocks per record > the one pass' room: the two kernels ran
    check_library_against_oracle(d, o)
    d.close()
    recs = B.random_records(rng, 30000)
    raw = np.frombuffer(b"".join(recs), np.uint8)
    off = bam.index(raw)
    o = oracle.bam_decode(raw, off, n_ref=3)
    one = bam.decode(raw, off, bam.BamOptions(n_ref=3), device=ctx)
    monkeypatch.setenv("SBGPU_BAM_TWO_PASS", "1")
    two = bam.decode(raw, off, bam.BamOptions(n_ref=3), device=ctx)
    monkeypatch.delenv("SBGPU_BAM_TWO_PASS")
    check_library_against_oracle(one, o)
    check_library_against_oracle(two, o)
    for k in ("status", "record", "read_id", "ref", "left", "right", "partner_pos", "flags", "nh", "nm", "read_len", "sam_flag",
              "block_off", "block_left", "block_right"):
        assert np.array_equal(getattr(one, k), getattr(two, k)), k
    assert one.by_status == two.by_status and one.any_paired == two.any_paired
    one.close()
    two.close()


def test_device_decode_on_garbage_records(ctx, oracle):
    """Noise behind valid size words (tests/test_bamdecode.py has the host form under AddressSanitizer): the device must take
    the oracle's way through it -- staged in LDS or walked in global memory (the second form is reached through records too long
    for the buffer)."""
    from strawberry_amd import bam
    recs = B.garbage_records(np.random.default_rng(32), 20000)
    long_tail = [B.record(1, 50, 0, "long%d" % k, [("M", 30000)], tags=[("NH", "C", 1), ("XS", "A", "-")]) for k in range(70)]   # 45 KB each
    raw = np.frombuffer(b"".join(recs + long_tail + recs[:500]), np.uint8)
    off = bam.index(raw)
    for kw in (dict(), dict(unique_only=False, library=1)):
        d = bam.decode(raw, off, bam.BamOptions(n_ref=3, **kw), device=ctx)
        check_library_against_oracle(d, oracle.bam_decode(raw, off, n_ref=3, **kw))
        assert d.by_status["OK"] > 300
        d.close()


def test_device_decode_of_long_records_only(ctx, oracle):
    """Streams whose AVERAGE record is long (long reads: kilobases of sequence and qualities in the record): 64 of them fit no
    staging buffer, every record is walked in global memory -- and the launch must not ask for more LDS than a workgroup has."""
    from strawberry_amd import bam
    rng = np.random.default_rng(8)
    for length in (900, 3000, 20000):
        recs = []
        for k in range(300):
            n = int(rng.integers(length // 2, length))
            cig = [("S", 5), ("M", n // 2), ("N", 500), ("M", n - n // 2)] if k % 3 else [("M", n)]
            recs.append(B.record(int(rng.integers(0, 3)), int(rng.integers(0, 10 ** 6)), 0, "long.%d" % k, cig, tags=[("NH", "C", 1), ("XS", "A", "-"), ("NM", "C", 3)]))
        raw = np.frombuffer(b"".join(recs), np.uint8)
        d = bam.decode(raw, None, bam.BamOptions(n_ref=3), device=ctx)
        o = oracle.bam_decode(raw, n_ref=3)
        check_library_against_oracle(d, o)
        assert d.n_reads == 300 and (d.read_len > length // 2 - 1).all()
        d.close()


def test_device_decode_refuses_offsets_outside_the_stream(ctx, oracle):
    """The device form cannot look at the caller's offsets beforehand (they are device data): a record whose offsets do not
    ascend inside the stream comes back TRUNCATED and is not read; its neighbours are decoded as ever."""
    from strawberry_amd import bam
    z = np.load(os.path.join(GOLDEN, "bamdecode_cases.npz"))
    raw = z["rec_bytes"]
    off = bam.index(raw)
    o = oracle.bam_decode(raw, off, n_ref=3)
    bad = off.copy()
    bad[200] = off[203]              # record 199 is given too much room (it knows its own size); 200 starts behind its end
    bad[700] = -5                    # 699 and 700
    bad[-1] = raw.size + 4096        # the last record claims bytes beyond the stream
    d = bam.decode(raw, bad, bam.BamOptions(n_ref=3), device=ctx)
    want = o["status"].copy()
    want[[200, 699, 700, off.size - 2]] = 10
    np.testing.assert_array_equal(d.status, want)
    keep = np.flatnonzero(want == 0)
    np.testing.assert_array_equal(d.record, keep)
    np.testing.assert_array_equal(d.left, o["left"][keep])
    np.testing.assert_array_equal(d.read_id, o["read_id"][keep])
    d.close()


def test_device_pairing_with_insertions(ctx, oracle):
    """Mates whose aligned blocks touch (an insertion in the read): no INTRON between them, on the device as on the host."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(77)
    clusters = []
    for l in range(12):
        c = MU.random_cluster(rng, int(rng.integers(20, 400)), base=300000 * (l + 1), exotic=l % 2 == 0)
        for r in c:
            if rng.random() < 0.3:                       # split a block in two that touch
                b = list(r["blocks"])
                k = int(rng.integers(0, len(b)))
                lo, hi = b[k]
                if hi - lo >= 4:
                    cut = int(rng.integers(lo + 1, hi))
                    b[k:k + 1] = [(lo, cut - 1), (cut, hi)]
                    r["blocks"] = b
        clusters.append(c)
    loc = [l for l, c in enumerate(clusters) for _ in c]
    reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
    got = eb.pair_mates(len(clusters), reads, device=ctx)
    from test_matepair_gpu import check_against_oracle
    check_against_oracle(oracle, clusters, got)
    host = eb.pair_mates(len(clusters), reads)
    for k in ("pair_off", "mass", "left_off", "right_off"):
        np.testing.assert_array_equal(got[k], host[k], err_msg=k)
    for side in ("left", "right"):
        for x, y in zip(got[side], host[side]):
            np.testing.assert_array_equal(x, y)
    assert int((np.diff(reads.block_off) > 1).sum()) > 100


@pytest.mark.parametrize("which", ["E2E", "E2E_MASS", "E2E_MINUS", "E2E_CHROMS"])
def test_bam_bytes_to_unique_hits_on_the_device_equal_reference_runs(ctx, which):
    import torch
    from strawberry_amd import _lib, bam
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    raw, chrom_names, (c_ref, c_left, c_right, c_strand) = B.toy_run_as_bam_records(d, names)
    L = _lib.load()
    dec = bam.decode(raw, None, bam.BamOptions(unique_only=which != "E2E_MASS", n_ref=len(chrom_names)), device=ctx)   # (the mass run: --allow-multimapped-hits)
    assert dec.on_device and dec.n_reads == dec.n_records and dec.any_paired
    rs, d_ref, d_left, d_right = dec.device_reads()
    # the read stream: which cluster every record is offered to (Sample::nextClusterRefDemand's pass), flags in place
    c_ref, c_left, c_right = (np.ascontiguousarray(x, t) for x, t in ((c_ref, np.int32), (c_left, np.uint32), (c_right, np.uint32)))
    c_strand = np.ascontiguousarray(c_strand, np.uint8)
    cl = _lib.sbgpu_clusters_t(len(names), c_ref.ctypes.data, c_left.ctypes.data, c_right.ctypes.data, c_strand.ctypes.data)
    dev = torch.device("cuda", ctx.device)
    d_cluster = torch.zeros(max(dec.n_reads, 1), dtype=torch.int32, device=dev)
    off = np.zeros(len(names) + 1, np.int64)
    _lib.check(L.sbgpu_assign_reads_device(ctx.h, C.byref(cl), dec.n_reads, d_ref, d_left, d_right, rs.flags, d_cluster.data_ptr(),
                                           off.ctypes.data, None), "sbgpu_assign_reads_device")
    assert off[-1] == dec.n_reads
    mh = C.c_void_p()
    _lib.check(L.sbgpu_pair_mates_device(ctx.h, len(names), C.byref(rs), off.ctypes.data, None, C.byref(mh)), "sbgpu_pair_mates_device")
    dp = _lib.sbgpu_pairs_t()
    poff = C.c_void_p()
    _lib.check(L.sbgpu_matepairs_pairs(mh, C.byref(dp), C.byref(poff)), "sbgpu_matepairs_pairs")
    uh = C.c_void_p()
    _lib.check(L.sbgpu_collapse_pairs_device(ctx.h, len(names), C.byref(dp), poff, None, C.byref(uh)), "sbgpu_collapse_pairs_device")
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_uniq_dev_info(uh, info), "sbgpu_uniq_dev_info")
    nh_, nf_ = int(info[0]), int(info[1])
    hl, fo = np.zeros(nh_, np.int32), np.zeros(nh_ + 1, np.int64)
    fc, fl, fr, ms = np.zeros(nf_, np.uint8), np.zeros(nf_, np.uint32), np.zeros(nf_, np.uint32), np.zeros(nh_, np.float32)
    cm = np.zeros(len(names))
    _lib.check(L.sbgpu_uniq_dev_export(uh, hl.ctypes.data, fo.ctypes.data, fc.ctypes.data, fl.ctypes.data, fr.ctypes.data, ms.ctypes.data,
                                       cm.ctypes.data), "sbgpu_uniq_dev_export")
    L.sbgpu_uniq_dev_destroy(uh)
    L.sbgpu_matepairs_destroy(mh)
    dec.close()
    np.testing.assert_array_equal(hl, hits.hit_locus)
    np.testing.assert_array_equal(fo, hits.feat_off)
    np.testing.assert_array_equal(fc, hits.feat_code)
    np.testing.assert_array_equal(fl, hits.feat_left)
    np.testing.assert_array_equal(fr, hits.feat_right)
    np.testing.assert_array_equal(ms, hits.mass)
    assert int(info[4]) == rows[0]["total_mapped"] == hits.total_mapped
