"""bench.py's compaction leg alone (roofline_compact_blocks), for a quick look at it on a GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import bench
from kvcached_amd import capi
print(json.dumps(bench.compaction_roofline(capi, "cuda:0")))
