from .opt import pad_length_to_make_divisible, pad_mat_to_make_divisible
from .adaptive_scaling import (
    AdaptiveScalingInferencingConfig,
    AdaptiveScalingInferencingRoughInferResult,
    AdaptiveScalingInferencingPresiceInferResult,
    AdaptiveScalingInferencing,
)
from .graphs import GraphCache, param_stamp
