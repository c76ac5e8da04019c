from .algorithm_template import CoverAlgorithm  # noqa: F401
from .rqa_serra09 import Serra09  # noqa: F401
