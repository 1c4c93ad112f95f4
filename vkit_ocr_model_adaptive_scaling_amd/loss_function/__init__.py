from .adaptive_scaling import (
    Box,
    AdaptiveScalingRoughLossFunctionConifg,
    AdaptiveScalingRoughLossFunction,
    AdaptiveScalingPreciseLossFunctionConifg,
    AdaptiveScalingPreciseLossFunction,
)
