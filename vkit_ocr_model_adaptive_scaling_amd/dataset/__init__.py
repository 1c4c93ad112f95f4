from .adaptive_scaling import (RoughSample, PreciseSample, adaptive_scaling_dataset_collate_fn,
                               SyntheticAdaptiveScalingIterableDataset)
