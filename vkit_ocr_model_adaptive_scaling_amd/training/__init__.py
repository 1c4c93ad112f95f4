from .flat import FlatBuffers
from .optimizer import FlatAdamW, cosine_warm_restarts_lr
from .ddp import BucketedGradReducer, TwoPassStep, adaptive_scaling_buckets
from .checkpoint import (RestoreState, save_restore_state, load_restore_state, optimizer_state_dict,
                         load_optimizer_state_dict, scheduler_state_dict, build_model_from_state_dict_path)
from .metrics import Metrics
from .opt import (batch_to_device, device_is_cuda, enable_cudnn_benchmark, enable_cudnn_deterministic, setup_seeds,
                  calculate_iterable_dataset_num_samples)
from .harness import EpochConfig, OptimizerConfig, MetricsTag, EpochResult, evaluate, run_training
