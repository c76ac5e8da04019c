from .algorithm_template import CoverAlgorithm  # noqa: F401
from .rqa_serra09 import Serra09  # noqa: F401
from .simple_silva import Simple  # noqa: F401
from .earlyfusion_traile import EarlyFusion  # noqa: F401
from .latefusion_chen import ChenFusion  # noqa: F401
