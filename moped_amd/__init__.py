"""moped_amd: MI355X (gfx950) implementation of libmoped's per-frame hot path
(MATCH -> CLUSTER -> POSE [-> FILTER -> POSE2 -> FILTER2]) behind a C ABI
(include/moped_hip.h, moped_amd/libmoped_hip.so).  Python here is only the
ctypes binding (capi), seeded synthetic workloads (synth) and the frame driver
used by tests and bench.py."""
__all__ = ["capi", "synth"]
