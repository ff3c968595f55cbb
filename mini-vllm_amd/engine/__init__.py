from .block_manager import BlockManager
from .scheduler import Scheduler
from .sequence import Sequence, SequenceStatus

__all__ = ["BlockManager", "Scheduler", "Sequence", "SequenceStatus"]
