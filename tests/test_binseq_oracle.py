"""A8 (per-bin sequence statistics of the `-f` table) on the CPU: the oracle's restatement of the
reference's include/kmer.h against (1) the reference's own templates where oracle/_ref is built,
(2) golden vectors made from them (tests/golden/binseq_cases.npz, tools/make_binseq_goldens.py) and
(3) the six columns the reference BINARY printed for every bin of a run with `-b genome.fa`
(tests/golden/e2e_toy_bias).  GC ratio and flags are exact; the entropy is compared bit for bit too
(same operations in the same order)."""
import os

import numpy as np

import e2e_util as U

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binseq_cases.npz")


def cases():
    g = dict(np.load(GOLD))
    seq = g["seq"].tobytes()
    return [(seq[g["off"][k]:g["off"][k + 1]], g["gc"][k], g["entropy"][k], int(g["flags"][k]))
            for k in range(len(g["off"]) - 1)]


def test_oracle_equals_reference_goldens(oracle):
    cs = cases()
    assert len(cs) > 600 and len(set(c[3] for c in cs)) >= 6      # every flag pattern that can occur, several times
    for seq, gc, ent, fl in cs:
        o = oracle.seq_stats(seq)
        assert o[0] == gc and o[2] == fl, (len(seq), o, gc, fl)
        assert o[1] == ent, (len(seq), o[1], ent)


def test_oracle_equals_reference_live(oracle, reflib):
    rng = np.random.Generator(np.random.PCG64(99))
    alpha = np.frombuffer(b"ACGTacgtN\x01\x02x", np.uint8)
    for t in range(400):
        n = int(rng.integers(41, 700))
        gc = float(rng.choice([0.2, 0.5, 0.8, 0.9]))
        p = np.concatenate([np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]) * 0.92, np.full(8, 0.01)])
        seq = rng.choice(alpha, size=n, p=p / p.sum()).tobytes()
        assert oracle.seq_stats(seq) == reflib.kmer_stats(seq)


def test_cutoffs_are_strict(oracle):
    """(double)gc / w > cutoff (kmer.h:72,85): 16 of 20 is not above 0.8, 17 is; 18 / 19 for 0.9; 32 / 33, 36 / 37 of 40."""
    for w, k, bit in ((20, 16, 0), (20, 18, 1), (40, 32, 2), (40, 36, 3)):
        at = oracle.seq_stats(b"A" * 50 + b"G" * k + b"A" * (w - k + 50))[2]
        above = oracle.seq_stats(b"A" * 50 + b"G" * (k + 1) + b"A" * (w - k + 49))[2]
        assert not (at >> bit) & 1 and (above >> bit) & 1


def test_batch_form_concatenates_segments(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=5000).tobytes()
    start = 1001                                   # genome[0] is base 1001
    bins = [[(1001, 1100)], [(1001, 1030), (1200, 1239)], [(2000, 2000), (2002, 2002), (2500, 2600)], [(5990, 6000), (1001, 1040)]]
    off = np.cumsum([0] + [len(b) for b in bins])
    sl = [a for b in bins for a, _ in b]
    sr = [c for b in bins for _, c in b]
    gc, ent, fl = oracle.binseq_batch(genome, start, off, sl, sr)
    for k, b in enumerate(bins):
        seq = b"".join(genome[a - start:c - start + 1] for a, c in b)
        assert (gc[k], ent[k], int(fl[k])) == oracle.seq_stats(seq)


def load_bias_run():
    """-> (genome bytes, rows of the reference's table) of tests/golden/e2e_toy_bias."""
    from strawberry_amd.binseq import read_fasta
    return read_fasta(os.path.join(U.E2E_BIAS, "genome.fa"))["chr1"], U.parse_ctx(os.path.join(U.E2E_BIAS, "ctx.tsv"))


def test_oracle_reproduces_the_reference_table(oracle):
    genome, rows = load_bias_run()
    assert len(rows) > 100 and len(set(tuple(r["seq"][2:]) for r in rows)) >= 4
    off = np.cumsum([0] + [len(r["coords"]) for r in rows])
    gc, ent, fl = oracle.binseq_batch(genome, 1, off, [a for r in rows for a, _ in r["coords"]],
                                      [b for r in rows for _, b in r["coords"]])
    for k, r in enumerate(rows):
        assert ["%f" % gc[k], "%f" % ent[k]] + [str((int(fl[k]) >> q) & 1) for q in range(4)] == r["seq"], r["coords"]
