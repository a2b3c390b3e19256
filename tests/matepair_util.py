"""Random clusters of alignment records for the mate-pairing tests: proper pairs (spliced or not), PCR duplicates, single
reads, mates on another reference, reads whose mate never arrives, multi-mapped reads (several alignments under one
read id), strands that disagree, partners at the read's own position, a record longer than kMaxFragSpan."""
import numpy as np

REVERSE, ELSEWHERE = 1, 2


def random_cluster(rng, n_frag, base=100000, exotic=True):
    """-> records (list of dicts, arrival order = sorted by start, stable)"""
    recs = []
    rid = int(rng.integers(1, 1 << 40))
    # every fragment starts at a position of its own: no two DIFFERENT hits share (left end, right end), whose order the
    # reference leaves to std::sort (PCR duplicates do share theirs, and are equal)
    starts = rng.permutation(max(4000, 2 * n_frag))[:2 * n_frag] + base
    for f in range(n_frag):
        rid += int(rng.integers(1, 1000))
        s = int(starts[2 * f])
        lb = [(s, s + 74)]
        if rng.random() < 0.3:
            cut, gap = int(rng.integers(10, 60)), int(rng.choice([200, 350]))
            lb = [(s, s + cut - 1), (s + cut + gap, s + gap + 74)]
        ins = int(rng.choice([180, 200, 230, 260]))
        rs = lb[-1][1] + 1 + ins - 75
        rb = [(rs, rs + 74)]
        nh = int(rng.choice([1, 1, 1, 2, 3]))
        xs = int(rng.choice([1, 1, 2, 0]))
        u = rng.random() if exotic else 1.0
        copies = int(rng.choice([1, 1, 2, 3]))
        for c in range(copies):
            this = rid + (c << 44)               # PCR duplicates are different reads
            left = {"id": this, "blocks": lb, "ppos": rb[0][0], "flags": (xs << 2), "nh": nh}
            right = {"id": this, "blocks": rb, "ppos": lb[0][0], "flags": REVERSE | (xs << 2), "nh": nh}
            if u < 0.06:                          # single read
                left["ppos"] = 0
                recs.append(left)
            elif u < 0.10:                        # mate on another reference
                right["flags"] |= ELSEWHERE
                recs.append(right)
            elif u < 0.14:                        # the mate never arrives
                recs.append(left)
            elif u < 0.17:                        # strands disagree: the two never pair
                right["flags"] = REVERSE | ((3 - xs if xs else 1) << 2)
                recs += [left, right] if xs else [left, right]
            elif u < 0.19:                        # partner at the read's own position
                left["ppos"] = lb[0][0]
                recs.append(left)
            elif u < 0.24:                        # a multi-mapped read: a second alignment pair under the same id
                s2 = int(starts[2 * f + 1]) + 6000
                l2 = {"id": this, "blocks": [(s2, s2 + 74)], "ppos": s2 + 200, "flags": (xs << 2), "nh": 2}
                r2 = {"id": this, "blocks": [(s2 + 200, s2 + 274)], "ppos": s2, "flags": REVERSE | (xs << 2), "nh": 2}
                left["nh"] = right["nh"] = 2
                recs += [left, right, l2, r2]
            else:
                recs += [left, right]
    if exotic and n_frag > 5:
        recs.append({"id": rid + 7, "blocks": [(base + 10, base + 40), (base + 1500000, base + 1500043)], "ppos": 0, "flags": 0, "nh": 1})
    order = np.argsort([r["blocks"][0][0] for r in recs], kind="stable")
    return [recs[i] for i in order]


def arrays(recs):
    return ([r["id"] for r in recs], [r["blocks"] for r in recs], [r["ppos"] for r in recs], [r["flags"] for r in recs],
            [r["nh"] for r in recs])
