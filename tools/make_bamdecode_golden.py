#!/usr/bin/env python3
"""tests/golden/bamdecode_cases.npz: BAM alignment records and what the REFERENCE makes of them.

Run where /root/reference is mounted (`make -C oracle ref` first).  tests/bam_util.py packs 1 500 records that reach
every branch of BAMHitFactory::getHitFromBuf (src/read.cpp:480-715) into a BGZF file; oracle/ref_shim.cpp's
ref_bam_decode opens that file with the reference's own BAMHitFactory (samtools 0.1.19 underneath) and reports, record by
record, whether getHitFromBuf took it and the ReadHit it built -- under four settings of the option globals.  Stored: the
uncompressed record stream (the decoder's input) and, per setting, the reference's answers.

  python tools/make_bamdecode_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SETTINGS = [  # (name, keyword arguments of RefLib.bam_decode / OracleLib.bam_decode / bam.BamOptions)
    ("default", dict()),
    ("multi_fr", dict(unique_only=False, library=1)),
    ("rf", dict(library=2)),
    ("introns_5_3000_multi", dict(min_intron=5, max_intron=3000, unique_only=False)),
]


def main():
    import bam_util as B
    from oracle import RefLib
    rng = np.random.default_rng(20260)
    recs = B.random_records(rng, 1500)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "cases.bam")
    B.write_bam(path, B.REFS, recs)
    refs, raw = B.read_bam_records(path)
    assert raw.tobytes() == b"".join(recs)
    R = RefLib()
    out = {"rec_bytes": raw, "n_ref": np.int32(len(refs))}
    for name, kw in SETTINGS:
        z = R.bam_decode(path, len(recs) + 8, raw.size // 4 + 8, **kw)
        assert z["n"] == len(recs)
        for k in ("accepted", "read_id", "ref", "left", "right", "strand", "partner_same_ref", "partner_pos", "nm", "nh", "flag_bits", "mass",
                  "read_len", "cig_off", "cig_type", "cig_len", "feat_off", "feat_code", "feat_left", "feat_right"):
            out[name + "/" + k] = z[k]
        out[name + "/single_end"] = np.int32(z["single_end"])
        print(name, "accepted", int(z["accepted"].sum()), "of", z["n"])
    dst = os.path.join(ROOT, "tests", "golden", "bamdecode_cases.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
