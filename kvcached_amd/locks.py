"""Lock stand-ins for synchronous scheduling (reference: kvcached/locks.py:51-97).

With a synchronous scheduler only one thread ever touches the KVCacheManager, so its
`@synchronized` methods take a lock object whose operations do nothing.
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Protocol


class LockLike(Protocol):
    """Structural type of what `@synchronized` needs: threading.RLock and NoOpLock both fit (reference :7-19)."""

    def acquire(self, blocking: bool = True, timeout: float = -1) -> bool: ...
    def release(self) -> None: ...
    def __enter__(self) -> bool: ...
    def __exit__(self, exc_type: Any, exc_val: Any, exc_tb: Any) -> None: ...


class ConditionLike(Protocol):
    """Structural type shared by threading.Condition and NoOpCondition (reference :22-48)."""

    def wait(self, timeout: Optional[float] = None) -> bool: ...
    def wait_for(self, predicate: Callable[[], bool], timeout: Optional[float] = None) -> bool: ...
    def notify(self, n: int = 1) -> None: ...
    def notify_all(self) -> None: ...


class NoOpLock:
    """Same interface as threading.RLock; every operation succeeds immediately."""

    def acquire(self, blocking: bool = True, timeout: float = -1) -> bool:
        return True

    def release(self) -> None:
        return None

    def __enter__(self) -> bool:
        return True

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        return None


class NoOpCondition:
    """Same interface as threading.Condition over a (no-op) lock."""

    def __init__(self, lock):
        self.lock = lock

    def wait(self, timeout: Optional[float] = None) -> bool:
        return True

    def wait_for(self, predicate: Callable[[], bool], timeout: Optional[float] = None) -> bool:
        return predicate()

    def notify(self, n: int = 1) -> None:
        return None

    def notify_all(self) -> None:
        return None

    def acquire(self, blocking: bool = True, timeout: float = -1) -> bool:
        return self.lock.acquire(blocking, timeout)

    def release(self) -> None:
        return self.lock.release()

    def __enter__(self):
        return self.lock.__enter__()

    def __exit__(self, exc_type, exc_val, exc_tb):
        return self.lock.__exit__(exc_type, exc_val, exc_tb)
