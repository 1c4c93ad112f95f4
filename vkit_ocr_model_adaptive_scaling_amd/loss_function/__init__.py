from .primitives import (
    WeightedBceWithLogitsLossFunction,
    CrossEntropyWithLogitsLossFunction,
    FocalWithLogitsLossFunction,
    L1LossFunction,
    L2LossFunction,
    WeightAdaptiveHeatmapRegressionLossFunction,
    DiceLossFunction,
)
from .adaptive_scaling import (
    Box,
    AdaptiveScalingRoughLossFunctionConifg,
    AdaptiveScalingRoughLossFunction,
    AdaptiveScalingPreciseLossFunctionConifg,
    AdaptiveScalingPreciseLossFunction,
)
