"""stvo_amd — Python plumbing around the MI355X-native PL-StVO hot path (tests / bench only).

The product is the C-ABI shared library built from stvo-pl_amd/csrc (include/stvo_hip.h); this
package only loads it through ctypes, generates synthetic inputs and drives benchmarks.
"""
