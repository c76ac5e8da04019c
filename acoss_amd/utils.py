"""
Host utilities mirroring the two helpers of acoss/utils.py that sit on the benchmark
path: create_dataset_filepaths (utils.py:87-102) and log (utils.py:24-35).
"""
import logging
import os

__all__ = ["create_dataset_filepaths", "log"]


def log(log_file):
    """Logger with a file and a console handler (same format as the reference).  Handlers
    are attached once per log file, so repeated calls do not duplicate output."""
    logger = logging.getLogger("acoss_amd.%s" % os.path.basename(log_file))
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
        fh = logging.FileHandler(log_file)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
        ch = logging.StreamHandler()
        ch.setFormatter(fmt)
        logger.addHandler(ch)
    return logger


def create_dataset_filepaths(dataset_csv, root_audio_dir, file_format=".mp3"):
    """CSV with exactly the columns work_id, track_id -> root + work_id/track_id + ext.
    Like the reference, `root_audio_dir` is string-concatenated (it must end with '/'),
    and any other column raises IOError."""
    import pandas as pd                       # (here, not at module level: importing the package must stay light)
    table = pd.read_csv(dataset_csv)
    for col in table.columns.tolist():
        if col not in ("work_id", "track_id"):
            raise IOError("Wrong input dataset csv annotation file '%s'. Expected a csv file with the "
                          "columns of key 'work_id', 'track_id'" % dataset_csv)
    return [root_audio_dir + str(w) + "/" + str(t) + file_format
            for w, t in zip(table["work_id"], table["track_id"])]


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (a container on a
    256-thread host is often allowed a dozen of them; os.cpu_count() still says 256)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n
