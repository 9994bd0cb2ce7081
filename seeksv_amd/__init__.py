"""seeksv_amd - MI355X-native implementation of seeksv's soft-clip breakpoint-clustering hot path.

Layout:
  csrc/   HIP kernels + the C ABI (include/seeksv_hip.h)  -> lib/libseeksv_hip.so
  host/   C++ host side: BAM/BGZF decoding to SoA batches, getsv bookkeeping, the `seeksv` CLI
  _abi.py ctypes struct mirrors; host.py / device.py thin wrappers used by tests and bench.py
"""
