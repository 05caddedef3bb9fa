"""Configuration (environment variables, read once at import) and small helpers.

Mirrors the names and semantics of the reference's kvcached/utils.py so that engine patches and
tools importing them keep working:
  PAGE_SIZE, GPU_UTILIZATION, PAGE_PREALLOC_ENABLED, MIN/MAX_RESERVED_PAGES, MAX_CACHED_BLOCKS,
  MAX_CACHED_TOKENS, SANITY_CHECK, CONTIGUOUS_LAYOUT, DEFAULT_IPC_NAME, SHM_DIR,
  KVCachedConfigError, normalize_gpu_device, align_to, align_up_to_page, get_kvcached_logger.
Reference lines are cited per item.
"""
from __future__ import annotations

import importlib.util
import logging
import os

SHM_DIR = "/dev/shm"
_TWO_MIB = 2 * 1024 * 1024


class KVCachedConfigError(RuntimeError):
    """A misconfiguration the user has to fix (e.g. a KV block larger than the page size).
    Integrations re-raise it instead of silently running without kvcached (utils.py:9-12)."""


def _env_flag(name: str, default: str) -> bool:
    return os.getenv(name, default).lower() == "true"


# ---- page size: KVCACHED_PAGE_SIZE_MB, a positive multiple of 2 MiB (utils.py:95-124)
def _get_page_size() -> int:
    raw = os.getenv("KVCACHED_PAGE_SIZE_MB")
    if raw is None:
        return _TWO_MIB
    try:
        size = int(raw) * 1024 * 1024
    except ValueError:
        raise ValueError(f"Invalid KVCACHED_PAGE_SIZE_MB: {raw}. Must be an integer.")
    if size <= 0 or size % _TWO_MIB != 0:
        raise ValueError(f"PAGE_SIZE must be a positive multiple of 2MB (2097152 bytes), got: {size}")
    return size


PAGE_SIZE = _get_page_size()

# ---- allocator knobs (utils.py:127-147); the native core reads the same variables itself
GPU_UTILIZATION = float(os.getenv("KVCACHED_GPU_UTILIZATION", "0.95"))
PAGE_PREALLOC_ENABLED = _env_flag("KVCACHED_PAGE_PREALLOC_ENABLED", "true")
# addition: an alloc() that needs several new pages backs them with ONE map call (same page ids, same offsets in the
# same order as the reference's page-by-page loop; one TLB invalidation / one TP broadcast instead of one per page)
BATCH_PAGE_ALLOC = _env_flag("KVCACHED_BATCH_PAGE_ALLOC", "true")
MIN_RESERVED_PAGES = int(os.getenv("KVCACHED_MIN_RESERVED_PAGES", "5"))
MAX_RESERVED_PAGES = int(os.getenv("KVCACHED_MAX_RESERVED_PAGES", "10"))
MAX_CACHED_BLOCKS = int(os.getenv("KVCACHED_MAX_CACHED_BLOCKS", "1000"))
SANITY_CHECK = _env_flag("KVCACHED_SANITY_CHECK", "false")
# < 0 unlimited, 0 disabled, > 0 cap on evictable cached tokens (used by the prefix-cache layers)
MAX_CACHED_TOKENS = int(os.getenv("KVCACHED_MAX_CACHED_TOKENS", "16000"))


# ---- layout default: explicit env wins; otherwise non-contiguous on ROCm (utils.py:150-171).
# On MI355X the per-layer layout is what vLLM's ROCm attention kernels read correctly; the
# compound-page layout stays available (and is ~10x cheaper to back) behind
# KVCACHED_CONTIGUOUS_LAYOUT=true.
def _default_contiguous_layout() -> bool:
    explicit = os.getenv("KVCACHED_CONTIGUOUS_LAYOUT")
    if explicit is not None:
        return explicit.lower() == "true"
    try:
        import torch
        if getattr(torch.version, "hip", None):
            return False
    except Exception:
        pass
    return True


CONTIGUOUS_LAYOUT = _default_contiguous_layout()


# ---- shm / socket namespace (utils.py:15-92)
def _sanitize_segment(segment: str) -> str:
    return "".join(ch if (ch.isalnum() or ch in "_-") else "-" for ch in segment)[:64]


def _detect_engine_tag() -> str:
    for mod, tag in (("vllm", "vLLM"), ("sglang", "SGLang")):
        if importlib.util.find_spec(mod) is not None:
            return tag
    return "proc"


def _ipc_segment_exists(name: str) -> bool:
    try:
        return os.path.exists(os.path.join(SHM_DIR, name))
    except Exception:
        return False


def _first_unused(base: str) -> str:
    if not _ipc_segment_exists(base):
        return base
    for i in range(1, 100):
        cand = f"{base}_{i}"
        if not _ipc_segment_exists(cand):
            return cand
    return f"{base}_{os.getpid()}"


def _obtain_default_ipc_name() -> str:
    """kvcached_<Engine>_<PGID>, or KVCACHED_IPC_NAME if that segment does not exist yet; a
    numeric suffix keeps separate launches apart."""
    tag = _detect_engine_tag()
    try:
        gid = os.getpgid(0)
    except Exception:
        try:
            gid = os.getsid(0)
        except Exception:
            gid = os.getpid()
    explicit = os.getenv("KVCACHED_IPC_NAME")
    if explicit:
        preferred = _sanitize_segment(explicit)
        if not _ipc_segment_exists(preferred):
            return preferred
        return _first_unused(f"{preferred}_{tag}_{gid}")
    return _first_unused(f"kvcached_{tag}_{gid}")


DEFAULT_IPC_NAME = _obtain_default_ipc_name()


# ---- device strings and alignment helpers (utils.py:189-209)
def normalize_gpu_device(device: str) -> str:
    """`hip[:N]` -> `cuda[:N]`: PyTorch-ROCm and c10::Device address AMD GPUs as cuda."""
    dev = str(device)
    return "cuda" + dev[3:] if dev.lower().startswith("hip") else dev


def align_to(x: int, a: int) -> int:
    return (x + a - 1) // a * a


def align_up_to_page(n_cells: int, cell_size: int) -> int:
    return align_to(n_cells, PAGE_SIZE // cell_size)


# ---- logging (utils.py:173-258)
LOG_USE_COLOR = _env_flag("KVCACHED_LOG_COLOR", "true")
_UNIFORM_COLOR = os.getenv("KVCACHED_LOG_COLOR_CODE", "\033[36m")
_LEVEL_COLORS = {logging.DEBUG: "\033[36m", logging.INFO: "\033[32m", logging.WARNING: "\033[33m",
                 logging.ERROR: "\033[31m", logging.CRITICAL: "\033[35m"}
_RESET = "\033[0m"


class ColorFormatter(logging.Formatter):
    def format(self, record: logging.LogRecord) -> str:
        text = super().format(record)
        color = _LEVEL_COLORS.get(record.levelno, _UNIFORM_COLOR)
        head, sep, rest = text.partition("] ")
        return f"{color}{head}{sep}{_RESET}{rest}" if sep else f"{color}{text}{_RESET}"


def get_log_level():
    return getattr(logging, os.getenv("KVCACHED_LOG_LEVEL", "INFO").upper(), logging.INFO)


def get_kvcached_logger(name: str = "kvcached") -> logging.Logger:
    logger = logging.getLogger(name)
    if not logger.handlers:
        handler = logging.StreamHandler()
        fmt = f"[{name}][%(levelname)s][%(asctime)s][%(filename)s:%(lineno)d] %(message)s"
        cls = ColorFormatter if (LOG_USE_COLOR and handler.stream.isatty()) else logging.Formatter
        handler.setFormatter(cls(fmt, datefmt="%Y-%m-%d %H:%M:%S"))
        logger.addHandler(handler)
        logger.setLevel(get_log_level())
        logger.propagate = False
    return logger
