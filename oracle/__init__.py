"""oracle — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path.  Only tests/, bench.py's `cpu_baseline`
leg and __graft_entry__.smoke() may import this package; nothing under visiondk_amd/ does, and the
product path fails loudly when the HIP library is missing instead of falling back to it.
"""
