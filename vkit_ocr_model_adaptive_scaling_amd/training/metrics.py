"""``vkit_open_model.training.Metrics`` (training/metrics.py:19-55): per tag, the mean of the last ``avg_num_batches``
values passed to ``update`` - what the train loop logs as L_rough / L_precise (train.py:415,453,519,552)."""
from collections import deque
from enum import Enum
from typing import Deque, Dict, Generic, Optional, Sequence, Type, TypeVar

_T = TypeVar('_T', bound=Enum)


class Metrics(Generic[_T]):

    def __init__(self, tag_enum_cls: Type[_T], avg_num_batches: int):
        if avg_num_batches < 1:
            raise ValueError('avg_num_batches must be positive')
        self.tag_enum_cls = tag_enum_cls
        self.avg_num_batches = avg_num_batches
        self._window: Dict[_T, Deque[float]] = {}
        self.tag_to_avg_value: Dict[_T, Optional[float]] = {}
        self.reset()

    def reset(self, tags: Optional[Sequence[_T]] = None):
        for tag in (tuple(self.tag_enum_cls) if tags is None else tags):
            self._window[tag] = deque(maxlen=self.avg_num_batches)
            self.tag_to_avg_value[tag] = None

    def update(self, tag: _T, value: float) -> float:
        window = self._window[tag]
        window.append(float(value))  # the oldest value drops out once the window is full
        avg = sum(window) / len(window)
        self.tag_to_avg_value[tag] = avg
        return avg
