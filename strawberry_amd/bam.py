"""BAM alignment records -> the read stream: the host-side mirror of sbgpu_bam_index_host / sbgpu_bam_decode_host /
sbgpu_bam_decode_device (include/sbgpu.h), i.e. of the reference's BAMHitFactory::getHitFromBuf
(/root/reference/src/read.cpp:480-715).

    off = bam.index(rec_bytes)                       # record offsets of the uncompressed record stream
    rd = bam.decode(rec_bytes, off, device=ctx)      # DecodedReads: accepted records as arrays, why the others were refused
    rd.reads()                                       # -> exonbin.Reads-like arguments for pair_mates / assign_reads

The record stream is what follows the BAM header once the BGZF blocks are inflated; inflating stays with the caller
(`split_header` below does it with Python's gzip for small files -- tests and examples)."""
import ctypes as C
import gzip
import struct

import numpy as np

from . import _lib


class BamOptions:
    """The reference's option globals that decide a record's fate (src/common.cpp:19-21,67-69)."""

    def __init__(self, min_intron=20, max_intron=300000, unique_only=True, library=0, n_ref=0):
        self.min_intron, self.max_intron, self.unique_only, self.library, self.n_ref = min_intron, max_intron, unique_only, library, n_ref

    def c(self):
        return _lib.sbgpu_bam_opts_t(int(self.min_intron), int(self.max_intron), int(bool(self.unique_only)), int(self.library), int(self.n_ref))


def split_header(bam_bytes):
    """A whole (small) BAM file's bytes -> ([(reference name, length)], the record stream as uint8)."""
    raw = gzip.decompress(bam_bytes)
    if raw[:4] != b"BAM\1":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack_from("<i", raw, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p)
        refs.append((raw[p + 4:p + 4 + l_name - 1].decode(), struct.unpack_from("<i", raw, p + 4 + l_name)[0]))
        p += 8 + l_name
    return refs, np.frombuffer(raw, np.uint8, offset=p).copy()


def index(rec_bytes):
    """-> int64[n + 1]: where every record starts (and where the stream ends)."""
    b = np.ascontiguousarray(rec_bytes, np.uint8)
    cap = b.size // 4 + 1      # the indexer accepts any block_size (a malformed stream comes back as TRUNCATED records, not as an error)
    off = np.zeros(cap + 1, np.int64)
    n = _lib.load().sbgpu_bam_index_host(b.ctypes.data if b.size else None, b.size, off.ctypes.data, cap)
    if n < 0:
        raise _lib.SbgpuError("sbgpu_bam_index_host: " + (_lib.load().sbgpu_last_error() or b"").decode())
    return off[:n + 1].copy()


class DecodedReads:
    """Host copies of an sbgpu_bamreads_t."""

    def __init__(self, handle, keep=None):
        L = _lib.load()
        info = (C.c_int64 * 16)()
        _lib.check(L.sbgpu_bamreads_info(handle, info), "sbgpu_bamreads_info")
        self.n_records, self.n_reads, self.n_blocks = int(info[0]), int(info[1]), int(info[2])
        self.any_paired, self.on_device = bool(info[3]), bool(info[4])
        self.by_status = {name: int(info[5 + k]) for k, name in enumerate(_lib.BAM_STATUS_NAMES)}
        n, m, nb = self.n_records, self.n_reads, self.n_blocks
        self.status = np.zeros(n, np.uint8)
        self.record = np.zeros(m, np.int64)
        self.read_id = np.zeros(m, np.uint64)
        self.ref, self.nh, self.nm, self.read_len = (np.zeros(m, np.int32) for _ in range(4))
        self.left, self.right, self.partner_pos, self.sam_flag = (np.zeros(m, np.uint32) for _ in range(4))
        self.flags = np.zeros(m, np.uint8)
        self.block_off = np.zeros(m + 1, np.int64)
        self.block_left, self.block_right = np.zeros(nb, np.uint32), np.zeros(nb, np.uint32)
        p = lambda a: a.ctypes.data if a.size else None
        _lib.check(L.sbgpu_bamreads_export(handle, p(self.status), p(self.record), p(self.read_id), p(self.ref), p(self.left), p(self.right),
                                           p(self.partner_pos), p(self.flags), p(self.nh), p(self.nm), p(self.read_len), p(self.sam_flag),
                                           p(self.block_off), p(self.block_left), p(self.block_right)), "sbgpu_bamreads_export")
        self._handle, self._keep = handle, keep

    def close(self):
        if self._handle is not None:
            _lib.load().sbgpu_bamreads_destroy(self._handle)
            self._handle = None

    __del__ = close

    def device_reads(self):
        """(sbgpu_reads_t, ref ptr, left ptr, right ptr) over the handle's own arrays (device arrays after a device decode)."""
        L = _lib.load()
        rs = _lib.sbgpu_reads_t()
        r, l, rr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(L.sbgpu_bamreads_reads(self._handle, C.byref(rs), C.byref(r), C.byref(l), C.byref(rr)), "sbgpu_bamreads_reads")
        return rs, r.value, l.value, rr.value


def decode(rec_bytes, rec_off=None, options=None, device=None):
    """Decode an uncompressed record stream.  device: a Context -> sbgpu_bam_decode_device (bytes and offsets uploaded
    through torch), else the host form."""
    L = _lib.load()
    opts = (options or BamOptions()).c()
    b = np.ascontiguousarray(rec_bytes, np.uint8)
    off = index(b) if rec_off is None else np.ascontiguousarray(rec_off, np.int64)
    n = off.size - 1
    h = C.c_void_p()
    if device is None:
        _lib.check(L.sbgpu_bam_decode_host(b.ctypes.data if b.size else None, b.size, off.ctypes.data, n, C.byref(opts), C.byref(h)),
                   "sbgpu_bam_decode_host")
        return DecodedReads(h)
    import torch
    dev = torch.device("cuda", device.device)
    hb = b if b.size else np.zeros(1, np.uint8)
    db = torch.from_numpy(hb if hb.flags.writeable else hb.copy()).to(dev)
    do = torch.from_numpy(off).to(dev)
    _lib.check(L.sbgpu_bam_decode_device(device.h, db.data_ptr(), b.size, do.data_ptr(), n, C.byref(opts), None, C.byref(h)),
               "sbgpu_bam_decode_device")
    return DecodedReads(h, keep=(db, do))
