#!/usr/bin/env python3
"""Lane-step model of the score pass (sw_fwd_*) on a sample of the bench workload: what an inter-sequence layout could gain for the tiles
of at most 64 rows, and what it needs.  CPU only: the prefilter hit lists come from the reference harness (oracle/_ref/ref_harness) on
the first contigs of bench.py's synthetic workload; nothing here touches the product path.

Modelled, per tile class, in VALU wave-instructions (the unit the SW stage is bound by):
  anti   the kernel as built (swp_kernel<R, RP, 16>): a wave = at most 8 pairs of ONE query, falling target length; every wave runs
         steps = roundup16(tMax + 15) of (8 hand-over + 10 R) instructions; rows padded to the tile (32 / 48 / 64)
  narrow the same layout on 8- or 4-lane groups (more rows per lane: the hand-over is amortised over more cells, the ramp is shorter), 16 / 32
         pairs of one query per wave
  inter  one lane = one packed pair of targets, all rows of the query in its registers (rows padded to a multiple of 8); a G-lane group
         works on one query at a time (its profile in LDS), its 2 G half-lanes take the query's pairs in falling length, dynamically
         (a half-lane pair that ends takes the next two targets); the group moves on when the query is done; the wave's NG = 64 / G groups
         progress independently.  A column costs 10 rows + 12 instructions for every lane of the WAVE while any group in it is busy;
         here: wave time = the sum over its columns, groups assumed to stay busy (queries dealt to groups by a work counter), so the
         loss is the idle half-lanes INSIDE a group: makespan x 2 G against the sum of the lengths.
  LDS    int16 profiles per wave = NG x 22 x rows x 2 bytes (the prefilter's two workgroups leave ~ 50 KB of a CU's 160 KB)
"""
import sys
import numpy as np


def load(tmp):
    tlen = np.load(tmp + "/tlen.npy")
    qlen = np.load(tmp + "/qlen.npy")
    hits = []
    cur = None
    for line in open(tmp + "/o/pref.txt"):
        if line[0] == ">":
            cur = []
            hits.append(cur)
        elif line.strip():
            cur.append(int(line.split("\t", 1)[0]))
    return tlen, qlen, hits


def tile_of(q):
    for r in (32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
        if q <= r:
            return r
    return 1024


def anti(tl_sorted, rows):
    """wave-instructions of the built kernel for one query's pairs (falling length)"""
    R = rows // 16
    per_step = 8 + 10 * R + (2 if rows == 48 else 0)
    ins = 0
    for i in range(0, len(tl_sorted), 8):
        tmax = tl_sorted[i]
        steps = (tmax + 15 + 15) & ~15
        ins += steps * per_step
    return ins


def anti_narrow(tl_sorted, rows, G):
    """the same kernel on G-lane groups (G = 8: 16 pairs per wave, G = 4: 32): three more instructions per step for the group borders
    inside a DPP row, steps in blocks of G"""
    R = rows // G
    per_step = 11 + 10 * R
    ins = 0
    per_wave = 2 * (64 // G)
    for i in range(0, len(tl_sorted), per_wave):
        steps = (tl_sorted[i] + G - 1 + G - 1) // G * G
        ins += steps * per_step
    return ins


def group_makespan(tl_sorted, slots):
    """pairs of consecutive targets (falling length) dealt to `slots` lanes, longest first, each lane taking the next pair when it ends"""
    pairs = [tl_sorted[i] for i in range(0, len(tl_sorted), 2)]          # a pair runs for its longer target
    lanes = [0] * slots
    for p in pairs:                                                       # greedy: the lane that is free first
        k = min(range(slots), key=lanes.__getitem__)
        lanes[k] += p
    return max(lanes), sum(pairs)


def make_sample(tmp, contigs=100):
    """the first contigs of bench.py's workload (seed 11) against its 100 000 proteins, through the reference harness"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import bench, oracle
    from metaeuk_amd import synth
    targets, queries = bench.make_inputs(contigs, 100000, 11, 0)
    os.makedirs(tmp, exist_ok=True)
    tf, qf = os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt")
    open(tf, "w").write("\n".join(synth.codes_to_str(t) for t in targets) + "\n")
    open(qf, "w").write("\n".join(synth.codes_to_str(q) for q in queries) + "\n")
    matdir = os.path.join(tmp, "mat")
    oracle.write_matrix_files(matdir)
    subprocess.check_call([oracle.REF, "pipeline", matdir, tf, qf, os.path.join(tmp, "o"), "--threads", str(os.cpu_count() or 1)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    np.save(os.path.join(tmp, "tlen.npy"), np.array([len(t) for t in targets]))
    np.save(os.path.join(tmp, "qlen.npy"), np.array([len(q) for q in queries]))


def position_pass(tmp):
    """columns the position pass still runs when the score pass bounds the end cell's column by the block of 16 steps in which the
    maximum last rose (swp_kernel -> gate_emit_kernel), on the alignments the reference reports for the sample"""
    full = cut = n = 0
    for line in open(tmp + "/o/aln.txt"):
        if line[0] == ">" or not line.strip():
            continue
        f = line.split("\t")
        dbend, dblen = int(f[8]), int(f[9])
        full += dblen
        cut += min(dblen, (dbend // 16 + 1) * 16 + 15)
        n += 1
    print("position pass: %d survivors, %.0f columns per job, %.0f behind the score pass's bound (%.2f)" % (n, full / n, cut / n, cut / full))


def main():
    tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp/swmodel"
    import os
    if not os.path.exists(tmp + "/o/pref.txt"):
        make_sample(tmp)
    tlen, qlen, hits = load(tmp)
    n = np.array([len(h) for h in hits])
    print("# sample: %d queries, %d pairs, %.1f pairs per query; queries with <8 / <16 / <32 pairs: %.2f / %.2f / %.2f; pairs in them: %.2f / %.2f / %.2f" % (
        len(hits), n.sum(), n.mean(), (n < 8).mean(), (n < 16).mean(), (n < 32).mean(),
        n[n < 8].sum() / n.sum(), n[n < 16].sum() / n.sum(), n[n < 32].sum() / n.sum()))
    print("# target length of a pair: mean %.0f, median %.0f, p90 %.0f, max %d" % tuple(
        f(np.concatenate([tlen[h] for h in hits if h])) for f in (np.mean, np.median, lambda x: np.percentile(x, 90), np.max)))
    for rows in (32, 48, 64):
        qs = [i for i in range(len(hits)) if tile_of(qlen[i]) == rows and hits[i]]
        cells = sum(int(qlen[i]) * int(tlen[hits[i]].sum()) for i in qs)
        a = 0
        narrow = {8: 0, 4: 0}
        inter = {}
        for i in qs:
            tl = sorted((int(x) for x in tlen[hits[i]]), reverse=True)
            a += anti(tl, rows)
            for G in narrow:
                narrow[G] += anti_narrow(tl, rows, G)
            r8 = (int(qlen[i]) + 7) // 8 * 8
            for G in (4, 8, 16):
                mk, tot = group_makespan(tl, G)
                col = 10 * r8 + 12
                e = inter.setdefault(G, [0.0, 0.0, 0.0])
                e[0] += mk * col / (64 // G)          # the group holds 1 / NG of the wave for its makespan
                e[1] += mk * G
                e[2] += tot
        print("rows %2d: %6d queries, %.3g cells | built: %.3g wave-instr = %.1f lane-instructions per useful cell (5 = the recurrence itself)" % (
            rows, len(qs), cells, a, a * 64 / cells))
        for G in (8, 4):
            print("          the built layout on %d-lane groups (%2d pairs of one query per wave, %2d rows per lane): %.3g wave-instr (%.2f x fewer)" % (
                G, 2 * (64 // G), rows // G, narrow[G], a / narrow[G]))
        for G in (4, 8, 16):
            e = inter[G]
            print("          inter-sequence, %2d-lane groups (%2d profiles = %5.1f KB per wave): %.3g wave-instr (%.2f x fewer), lanes busy %.2f" % (
                G, 64 // G, (64 // G) * 22 * rows * 2 / 1024, e[0], a / e[0], e[2] / e[1]))
    position_pass(tmp)


if __name__ == "__main__":
    main()
