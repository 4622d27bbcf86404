"""Synthetic JRDB directory tree (the layout JRDBHandle reads), shared by tools/gen_golden.py -- which runs
the reference's own JRDBHandle on it -- and the tests, which rebuild the identical tree from the seed and
compare against the recorded outputs.

<root>/train_dataset/timestamps/<seq>/frames_pc_laser.json
<root>/train_dataset/labels/labels_3d/<seq>.json
<root>/train_dataset/pointclouds/upper_velodyne/<seq>/<frame>.pcd
<root>/train_dataset/lasers/<seq>/<frame>.txt
"""
import json
import os

import numpy as np

from planar_optical_flow_amd import pcd_io

PCD_KINDS = ("binary", "ascii", "binary_compressed")


def make_tree(root, sequences, seed=11, frames_per_seq=3, pcd_kinds=("binary", "ascii")):
    """Writes the tree; -> list of (sequence, frame file name, n_annotations) for the labelled frames in
    index order.  Frame 1 of every sequence is unlabelled; the last frame of the first sequence is labelled
    with an empty annotation list."""
    rng = np.random.default_rng(seed)
    base = os.path.join(root, "train_dataset")
    labelled = []
    for q, seq in enumerate(sequences):
        os.makedirs(os.path.join(base, "timestamps", seq), exist_ok=True)
        os.makedirs(os.path.join(base, "labels", "labels_3d"), exist_ok=True)
        os.makedirs(os.path.join(base, "pointclouds", "upper_velodyne", seq), exist_ok=True)
        os.makedirs(os.path.join(base, "lasers", seq), exist_ok=True)
        frames, labels = [], {}
        for k in range(frames_per_seq):
            stem = "%06d" % k
            n_people = 0 if (q == 0 and k == frames_per_seq - 1) else int(rng.integers(1, 4))
            people = np.column_stack([rng.uniform(-4, 4, n_people), rng.uniform(-4, 4, n_people)])
            # cloud: a blob per person + clutter, base-frame-ish coordinates, float32 like the sensor files
            blobs = [p + rng.normal(0, 0.25, (int(rng.integers(20, 60)), 2)) for p in people]
            xy = np.concatenate(blobs + [rng.uniform(-6, 6, (150, 2))])
            z = rng.uniform(-0.9, 0.9, len(xy))
            cols = {"x": xy[:, 0].astype(np.float32), "y": xy[:, 1].astype(np.float32), "z": z.astype(np.float32),
                    "intensity": rng.uniform(0, 255, len(xy)).astype(np.float32)}
            kind = pcd_kinds[(q + k) % len(pcd_kinds)]
            pcd_io.write_pcd(os.path.join(base, "pointclouds", "upper_velodyne", seq, stem + ".pcd"), cols, data=kind)
            ranges = rng.uniform(0.3, 25.0, 180)
            with open(os.path.join(base, "lasers", seq, stem + ".txt"), "w") as f:
                f.write(" ".join("%.4f" % v for v in ranges) + "\n")
            frames.append({"frame_id": k, "timestamp": 0.0667 * k,
                           "pointclouds": {"upper_velodyne": {"url": "pointclouds/upper_velodyne/%s/%s.pcd" % (seq, stem),
                                                              "timestamp": 0.0667 * k}},
                           "laser": {"url": "lasers/%s/%s.txt" % (seq, stem), "timestamp": 0.0667 * k}})
            if k == 1:
                continue
            anns = []
            for j, p in enumerate(people):
                anns.append({"label_id": "pedestrian:%d" % j, "file_id": stem + ".pcd", "observation_angle": 0.0,
                             "attributes": {"num_points": 50},
                             "box": {"cx": round(float(p[0]), 4), "cy": round(float(p[1]), 4),
                                     "cz": round(float(rng.uniform(-0.3, 0.1)), 4),
                                     "l": round(float(rng.uniform(0.4, 1.0)), 3), "w": round(float(rng.uniform(0.4, 1.0)), 3),
                                     "h": round(float(rng.uniform(1.4, 1.9)), 3),
                                     "rot_z": round(float(rng.uniform(-4.0, 4.0)), 4)}})
            labels[stem + ".pcd"] = anns
            labelled.append((seq, stem + ".pcd", len(anns)))
        with open(os.path.join(base, "timestamps", seq, "frames_pc_laser.json"), "w") as f:
            json.dump({"data": frames}, f)
        with open(os.path.join(base, "labels", "labels_3d", seq + ".json"), "w") as f:
            json.dump({"labels": labels}, f)
    return labelled
