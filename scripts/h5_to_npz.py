#!/usr/bin/env python3
"""
Feature-store converter: acoss / DA-TACOS per-track HDF5 files (feature_dir/work_id/track_id.h5,
written by deepdish -- /root/reference/README.md:116-150, algorithm_template.py:90) -> the .npz files
acoss_amd reads natively, next to the originals (same stem).  Run it where deepdish or h5py is
installed; acoss_amd itself reads .h5 directly whenever h5py is importable, so the conversion is
only needed on machines without it (like the build container).

    python scripts/h5_to_npz.py FEATURE_DIR [--keys hpcp,crema,mfcc_htk,madmom_features,label,track_id]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_any(path):
    try:
        import deepdish as dd
        return dd.io.load(path)
    except ImportError:
        from acoss_amd.featurestore import _load_h5
        return _load_h5(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("feature_dir")
    ap.add_argument("--keys", default="", help="comma-separated keys to keep (default: all)")
    args = ap.parse_args()
    from acoss_amd.featurestore import save_track
    keep = set(k for k in args.keys.split(",") if k)
    n = 0
    for root, _, files in os.walk(args.feature_dir):
        for name in files:
            if not name.endswith((".h5", ".hdf5")):
                continue
            path = os.path.join(root, name)
            feats = load_any(path)
            if keep:
                feats = {k: v for k, v in feats.items() if k in keep}
            save_track(path, feats)
            n += 1
    print("converted %d track files under %s" % (n, args.feature_dir))


if __name__ == "__main__":
    main()
