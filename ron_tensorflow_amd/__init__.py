"""ron_tensorflow_amd: MI355X-native RON-320 inference hot path.

HIP/CDNA4 kernels behind the C ABI of include/ron_hip.h (libron_hip.so), with a Python host
layer that keeps the reference's nets_factory / RONNet interface.  No CPU fallback.
"""
__version__ = '0.1.0'
