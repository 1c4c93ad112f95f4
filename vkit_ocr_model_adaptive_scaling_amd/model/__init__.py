from .convnext import ConvNext
from .upernext import UperNextNeck, UperNextHead
from .fpn import FpnNeck, FpnHead
from .adaptive_scaling import (
    AdaptiveScalingSize,
    AdaptiveScalingNeckHeadType,
    AdaptiveScalingConfig,
    AdaptiveScaling,
)
from .helper import set_compute_dtype
