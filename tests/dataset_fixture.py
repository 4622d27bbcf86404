"""Rebuilds the synthetic sequences of tests/golden/dataset_items.npz (the inputs the
reference's DROWDataset2.__getitem__ was run on by tools/gen_golden.py)."""
import numpy as np

CUTOUT_KW = dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5, num_cutout_pts=56,
                 padding_val=29.99, area_mode=True)


def _ragged(cnt, val):
    out, o = [], 0
    for c in cnt:
        out.append([list(map(float, v)) for v in val[o:o + c]])
        o += c
    return out


def load_sequences(g):
    seqs = []
    for q in range(2):
        seqs.append({
            "scans": g["scans%d" % q], "scans_ns": g["scans_ns%d" % q], "scans_t": g["scans_t%d" % q],
            "odoms_t": g["odoms_t%d" % q], "odoms": g["odoms%d" % q], "dets_ns": g["dets_ns%d" % q],
            "dets_wc": _ragged(g["wc_cnt%d" % q], g["wc_val%d" % q]),
            "dets_wa": _ragged(g["wa_cnt%d" % q], g["wa_val%d" % q]),
            "dets_wp": _ragged(g["wp_cnt%d" % q], g["wp_val%d" % q]),
        })
    return seqs


def flat_samples(seqs):
    """(sequence index, scan index, wc, wa, wp) in the reference's flat order."""
    out = []
    for q, s in enumerate(seqs):
        for d_ns, wc, wa, wp in zip(s["dets_ns"], s["dets_wc"], s["dets_wa"], s["dets_wp"]):
            out.append((q, int(np.where(s["scans_ns"] == d_ns)[0][0]), wc, wa, wp))
    return out
