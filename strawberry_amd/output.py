"""GTF output exactly as the reference prints it.

Reference: Contig::print2gtf, /root/reference/src/contig.cpp:636-721; the FPKM / Frac / TPM
strings are std::to_string(double) cut to 11 characters.  Formatting is host work behind
sbgpu_format_value / sbgpu_format_gtf_transcript.
"""
import ctypes as C

import numpy as np

from . import _lib


def format_value(v):
    """first 11 characters of "%f" % v"""
    out = C.create_string_buffer(12)
    _lib.check(_lib.load().sbgpu_format_value(float(v), out), "sbgpu_format_value")
    return out.value.decode()


def gtf_transcript(chrom, strand, gene_id, transcript_id, exons, fpkm, frac, tpm, keep=1, ref_gene_id="",
                   ref_gene_name=""):
    """The transcript line + exon lines of one isoform, as print2gtf writes them."""
    L = _lib.load()
    left = np.ascontiguousarray([e[0] for e in exons], np.int32)
    right = np.ascontiguousarray([e[1] for e in exons], np.int32)
    cap = 4096 + 400 * len(exons)
    buf = C.create_string_buffer(cap)
    n = L.sbgpu_format_gtf_transcript(buf, cap, chrom.encode(), strand.encode(), gene_id.encode(),
                                      transcript_id.encode(), ref_gene_id.encode(), ref_gene_name.encode(), len(exons),
                                      left.ctypes.data, right.ctypes.data, float(fpkm), float(frac), float(tpm),
                                      int(keep))
    if n < 0:
        _lib.check(n, "sbgpu_format_gtf_transcript")
    return buf.value.decode()
