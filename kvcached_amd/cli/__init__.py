"""Only the shared-memory record helpers of the reference's `kvcached.cli` (cli/utils.py) live here: they are the
data format on the control side of the hot path (the `{total, used, prealloc}` triple PageAllocator's resize watcher
reads and its accounting writes). The command line tools themselves (kvctl, kvtop) are out of scope;
the reference's own, installed next to this package, work on these helpers unchanged."""
