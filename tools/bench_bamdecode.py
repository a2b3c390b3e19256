#!/usr/bin/env python3
"""GPU box: sbgpu_bam_decode_device on a record stream RESIDENT in HBM -- n records of a realistic mix (paired reads of
100 bases, 30 % spliced, 3 % with an insertion or deletion, NH / NM / XS / MD tags, sequence and qualities in the record:
~330 bytes each), a random bag of 100 000 repeated -- with the HBM roofline of the decode (the record bytes in, the read
arrays out) and the oracle's loop on a host core beside it.  Prints one JSON line.
usage: bench_bamdecode.py [n_records=4e6] [--no-cpu-baseline]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bam_util as B
from strawberry_amd import _lib, bam, em

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_want = int(float(args[0])) if args else 4_000_000
rng = np.random.default_rng(3)
bag = []
for i in range(100_000):
    pos = int(rng.integers(0, 2_000_000))
    cig = [("M", 100)]
    u = rng.random()
    if u < 0.30:
        cut = int(rng.integers(10, 90))
        cig = [("M", cut), ("N", int(rng.integers(60, 20000))), ("M", 100 - cut)]
    elif u < 0.33:
        cig = [("S", 4), ("M", 40), (("I", "D")[int(rng.integers(0, 2))], 2), ("M", 54)]
    rev = i & 1
    tags = [("NH", "C", 1 if rng.random() < 0.9 else 3), ("NM", "C", int(rng.integers(0, 4))), ("XS", "A", "+-"[int(rng.integers(0, 2))]),
            ("MD", "Z", "100"), ("AS", "i", -3)]
    bag.append(B.record(int(rng.integers(0, 3)), pos, 1 | 2 | (0x10 if rev else 0x20) | (0x80 if rev else 0x40), "read.%d" % (i // 2), cig,
                        mtid=-2, mpos=pos + 180, tags=tags))
one = b"".join(bag)
reps = max(1, n_want // len(bag))
raw = np.frombuffer(one * reps, np.uint8)
off = bam.index(raw)
n = off.size - 1
ctx = em.default_context(0)
L, dev = ctx.L, torch.device("cuda", ctx.device)
d_bytes = torch.from_numpy(raw.copy()).to(dev)
d_off = torch.from_numpy(off).to(dev)
opts = bam.BamOptions(n_ref=3).c()
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
times = []
info = (C.c_int64 * 16)()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(8):
    h = C.c_void_p()
    torch.cuda.synchronize()
    t = time.time()
    _lib.check(L.sbgpu_bam_decode_device(ctx.h, d_bytes.data_ptr(), raw.size, d_off.data_ptr(), n, C.byref(opts), stream, C.byref(h)), "sbgpu_bam_decode_device")
    torch.cuda.synchronize()
    times.append((time.time() - t) * 1e3)
    _lib.check(L.sbgpu_bamreads_info(h, info), "info")
    L.sbgpu_bamreads_destroy(h)
ms = float(np.median(times[2:]))
m, nb = int(info[1]), int(info[2])
bytes_in = raw.size + 8 * (n + 1)
bytes_out = n + m * (8 + 8 + 4 * 8 + 1 + 8) + nb * 8
out = {"metric": "BAM records decoded/s (records resident in HBM)", "value": n / (ms * 1e-3), "unit": "records/s", "n_gpus": 1,
       "ms_per_call": ms, "dtype": "u8", "data": "synthetic",
       "config": {"workload": "bam-decode", "records": n, "bytes": int(raw.size), "bytes_per_record": raw.size / n, "accepted": m, "blocks": nb},
       "roofline": {"bound": "hbm", "achieved": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                    "algorithmic_bytes": bytes_in + bytes_out,
                    "note": "whole call (two kernels, two scans, one 16-byte read-back between them), wall clock around a synchronised call"}}
if "--no-cpu-baseline" not in sys.argv:
    from oracle import OracleLib
    O = OracleLib()
    k = min(n, 2_000_000)
    sub = raw[:off[k]]
    t = time.time()
    o = O.bam_decode(sub, off[:k + 1], n_ref=3)
    dt = time.time() - t
    hd = bam.decode(sub, off[:k + 1], bam.BamOptions(n_ref=3))
    ok = bool(np.array_equal(hd.status, o["status"]))
    out["cpu_baseline"] = {"value": k / dt, "unit": "records/s", "cores": 1, "kind": "port",
                           "sample": "the first %d records of the same stream, oracle/bamdecode_oracle.c, 1 thread, %.2f s (record bytes already inflated: "
                                     "the reference's own BAMHitFactory also pays zlib and bam_read1)" % (k, dt), "parity": ok}
print(json.dumps(out))
