"""
acoss_amd -- MI355X-native all-pairwise cover-song similarity engine behind the
acoss.coverid.benchmark() / CoverAlgorithm.all_pairwise() / similarity() surface.
Importing the package does not touch the GPU; the first similarity() call loads
libacx.so (HIP kernels, C ABI in include/acx.h) and raises if it or the device is missing.
"""
from .coverid import benchmark, algorithm_names  # noqa: F401

__all__ = ["benchmark", "algorithm_names"]
