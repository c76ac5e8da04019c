"""
benchmark(): the user-facing entry point, same signature as acoss.coverid.benchmark
(coverid.py:22-28).  Only the algorithms on the accelerated path are available; the
reference's "EarlyFusionTraile" / "EarlyFusion" naming mismatch (coverid.py:19 vs :72,
which makes the reference silently do nothing) is resolved by accepting both names.
"""
import time

from .utils import log

__all__ = ["benchmark", "algorithm_names"]

_LOG_FILE_PATH = "acoss.coverid.log"

# names the reference advertises (coverid.py:19)
algorithm_names = ["Serra09", "EarlyFusionTraile", "LateFusionChen", "FTM2D", "SiMPle"]
_device_algorithms = ("Serra09", "SiMPle", "EarlyFusionTraile", "EarlyFusion", "LateFusionChen")


def benchmark(dataset_csv, feature_dir, feature_type="hpcp", algorithm="Serra09", shortname="covers80",
              parallel=True, n_workers=-1):
    """Run one cover-id algorithm over a dataset annotation csv and print / store the
    evaluation statistics.  `parallel` / `n_workers` are accepted and ignored: the unit of
    parallelism is the GPU (launch one process per GPU with torch.distributed to use more
    than one).  Raises NotImplementedError for an unknown algorithm, like the reference,
    and for the reference algorithms that are outside this engine's scope."""
    logger = log(_LOG_FILE_PATH)
    if algorithm not in algorithm_names and algorithm not in _device_algorithms:
        warn = ("acoss.coverid: Couldn't find '%s' algorithm in acoss. Available cover id algorithms are %s"
                % (algorithm, str(algorithm_names)))
        logger.debug(warn)
        raise NotImplementedError(warn)
    if algorithm not in _device_algorithms:
        raise NotImplementedError("'%s' is not part of the MI355X engine (available: %s)"
                                  % (algorithm, list(_device_algorithms)))
    logger.info("Running acoss cover identification benchmarking for the algorithm - '%s'" % algorithm)
    start = time.monotonic()
    results = {}
    if algorithm == "Serra09":
        from .algorithms.rqa_serra09 import Serra09
        algo = Serra09(dataset_csv=dataset_csv, datapath=feature_dir, chroma_type=feature_type,
                       shortname=shortname)
        logger.info("Computing pairwise similarity...")
        algo.all_pairwise(parallel, n_cores=n_workers, symmetric=True)
        algo.normalize_by_length()
    elif algorithm == "LateFusionChen":
        from .algorithms.latefusion_chen import ChenFusion
        algo = ChenFusion(dataset_csv=dataset_csv, datapath=feature_dir, chroma_type=feature_type,
                          shortname=shortname)
        logger.info("Computing pairwise similarity...")
        algo.all_pairwise(parallel, n_cores=n_workers, symmetric=True)
        algo.normalize_by_length()
        algo.do_late_fusion()
    elif algorithm == "SiMPle":
        from .algorithms.simple_silva import Simple
        algo = Simple(dataset_csv=dataset_csv, datapath=feature_dir, chroma_type=feature_type,
                      shortname=shortname)
        for i in range(len(algo.filepaths)):
            algo.load_features(i)
        logger.info("Feature loading done...")
        algo.all_pairwise(parallel, n_cores=n_workers, symmetric=False)
    else:
        from .algorithms.earlyfusion_traile import EarlyFusion
        algo = EarlyFusion(dataset_csv=dataset_csv, datapath=feature_dir, chroma_type=feature_type,
                           shortname=shortname)
        for i in range(len(algo.filepaths)):
            algo.load_features(i)
        logger.info("Feature loading done...")
        algo.all_pairwise(parallel, n_cores=n_workers, symmetric=True)
        algo.do_late_fusion()
    logger.info("Running benchmark evaluations on the given dataset - %s" % dataset_csv)
    for similarity_type in list(algo.Ds.keys()):
        results[similarity_type] = algo.getEvalStatistics(similarity_type)
    algo.cleanup_memmap()
    logger.info("acoss.coverid benchmarking finsihed in %s" % (time.monotonic() - start))
    logger.info("Log file located at '%s'" % _LOG_FILE_PATH)
    return results
