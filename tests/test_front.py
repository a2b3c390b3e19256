"""The synthetic BAM record stream of bench.py's `c3-front` workload (strawberry_amd/front.py::pack_bam_records), checked on
the CPU: the records torch packs for a small chain sample are well-formed BAM records (the library's host indexer and
decoder AND the oracle's decoder -- pinned on the reference's own BAMHitFactory -- read every one of them as an accepted,
paired, uniquely mapped read with the blocks the sample drew), and the library's host forms of the stages behind the decode
(read stream -> pairs -> unique hits) give back exactly the unique hits the sample was made of: the input of the resident
front-end line is what it claims to be."""
import ctypes as C

import numpy as np
import pytest


def test_packed_records_decode_pair_and_collapse_back_to_the_samples_unique_hits(oracle):
    import torch
    from strawberry_amd import _lib, bam, chain, front
    from strawberry_amd import exonbin as eb
    n_loci = 80
    s = chain.DeviceSample(torch, torch.device("cpu"), n_loci=n_loci, n_frags=9000, seed=5)
    raw_t, off_t = front.pack_bam_records(torch, s, chunk=1000)          # (several chunks)
    raw, off = raw_t.numpy(), off_t.numpy()
    n = len(off) - 1
    assert n == 2 * s.n_fragments and s.n_fragments > s.n_hits           # duplicates: some hits stand for several pairs
    np.testing.assert_array_equal(bam.index(raw), off)                    # block_size fields chain through the stream
    dec = bam.decode(raw, off, bam.BamOptions(n_ref=1))
    assert dec.n_reads == n and dec.any_paired and dec.by_status["OK"] == n
    o = oracle.bam_decode(raw, off, n_ref=1)
    np.testing.assert_array_equal(dec.status, o["status"])
    for k in ("read_id", "left", "right", "partner_pos", "nh", "read_len"):
        np.testing.assert_array_equal(getattr(dec, k), o[k], err_msg=k)
    assert (dec.read_len == 75).all() and (dec.nh == 1).all()
    assert (np.diff(dec.left.astype(np.int64)) >= 0).all()                # coordinate-sorted
    # mates share a read id, pairs do not
    ids, cnt = np.unique(dec.read_id, return_counts=True)
    assert len(ids) == s.n_fragments and (cnt == 2).all()
    # ---- the read stream -> clusters -> pairs -> unique hits, host forms of the library
    a = s.annot
    c_left = np.minimum.reduceat(np.minimum.reduceat(a.exon_left, a.exon_off[:-1]), a.iso_off[:-1])
    c_right = np.maximum.reduceat(np.maximum.reduceat(a.exon_right, a.exon_off[:-1]), a.iso_off[:-1])
    cluster, read_off, flags = eb.assign_reads(np.zeros(n_loci, np.int32), c_left, c_right, np.ones(n_loci, np.uint8), dec.ref, dec.left, dec.right, dec.flags)
    assert read_off[-1] == n and (cluster >= 0).all()
    L = _lib.load()
    p = lambda x: x.ctypes.data  # noqa: E731
    rs = _lib.sbgpu_reads_t(n, p(dec.read_id), p(dec.block_off), p(dec.block_left), p(dec.block_right), p(dec.partner_pos), p(flags), p(dec.nh))
    hm = C.c_void_p()
    _lib.check(L.sbgpu_pair_mates_host(n_loci, C.byref(rs), p(read_off), C.byref(hm)), "sbgpu_pair_mates_host")
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_matepairs_info(hm, info), "sbgpu_matepairs_info")
    assert int(info[0]) == int(info[1]) == s.n_fragments and int(info[3]) == int(info[4]) == 0     # every record found its mate
    pairs, poff = _lib.sbgpu_pairs_t(), C.c_void_p()
    _lib.check(L.sbgpu_matepairs_pairs(hm, C.byref(pairs), C.byref(poff)), "sbgpu_matepairs_pairs")
    hu = C.c_void_p()
    _lib.check(L.sbgpu_collapse_pairs_host(n_loci, C.byref(pairs), C.byref(hu)), "sbgpu_collapse_pairs_host")
    ui = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_uniq_info(hu, ui), "sbgpu_uniq_info")
    nh, nf = int(ui[0]), int(ui[1])
    hit_locus, feat_off = np.zeros(nh, np.int32), np.zeros(nh + 1, np.int64)
    code, left, right = np.zeros(nf, np.uint8), np.zeros(nf, np.uint32), np.zeros(nf, np.uint32)
    hmass, cmass = np.zeros(nh, np.float32), np.zeros(n_loci, np.float64)
    _lib.check(L.sbgpu_uniq_export(hu, p(hit_locus), p(feat_off), p(code), p(left), p(right), p(hmass), p(cmass)), "sbgpu_uniq_export")
    L.sbgpu_uniq_destroy(hu)
    L.sbgpu_matepairs_destroy(hm)
    want = s.host_hits(n_loci)
    assert int(ui[2]) == 0, "the span filter dropped pairs of a sample whose fragment lengths are N(250, 30)"
    np.testing.assert_array_equal(hit_locus, want.hit_locus)
    np.testing.assert_array_equal(feat_off, want.feat_off)
    np.testing.assert_array_equal(code, want.feat_code)
    np.testing.assert_array_equal(left, want.feat_left)
    np.testing.assert_array_equal(right, want.feat_right)
    np.testing.assert_array_equal(hmass, want.mass)


def test_packed_records_through_the_references_own_bam_reader(oracle, reflib, tmp_path, capfd):
    """The same packed stream as a BGZF BAM file through the REFERENCE's BAMHitFactory::getHitFromBuf (samtools 0.1.19
    underneath; oracle/_ref): every record accepted, and read id, interval, strand, mate position, NH, read length, the kept
    CIGAR and readhit_2_genomicFeats' features equal the oracle decoder's on the bytes -- the synthetic input of the c3-front
    line is a BAM file the reference itself reads as the reads the sample drew."""
    import torch
    import bam_util as B
    from test_bamdecode import against_reference
    from strawberry_amd import chain, front
    s = chain.DeviceSample(torch, torch.device("cpu"), n_loci=40, n_frags=3000, seed=9)
    raw_t, off_t = front.pack_bam_records(torch, s)
    raw = raw_t.numpy()
    n = int(off_t.numel()) - 1
    top = int(s.feat_right.max()) + 10000
    path = str(tmp_path / "packed.bam")
    with open(path, "wb") as f:
        f.write(B.bgzf_compress(B.header_bytes([("chr1", top)]) + raw.tobytes()))
    z = reflib.bam_decode(path, n + 8, raw.size // 4 + 8)
    o = oracle.bam_decode(raw, off_t.numpy(), n_ref=1)
    assert (o["status"] == 0).all() and o["n"] == n
    against_reference(o, z)
    capfd.readouterr()
