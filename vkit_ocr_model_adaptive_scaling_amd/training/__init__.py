from .flat import FlatBuffers
from .optimizer import FlatAdamW, cosine_warm_restarts_lr
from .ddp import BucketedGradReducer, TwoPassStep, adaptive_scaling_buckets
