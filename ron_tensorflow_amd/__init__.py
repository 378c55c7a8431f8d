"""ron_tensorflow_amd: MI355X-native RON-320 inference hot path.

HIP/CDNA4 kernels behind the C ABI of include/ron_hip.h (libron_hip.so), with a Python host
layer that keeps the reference's nets_factory / RONNet interface.  No CPU fallback.
"""
__version__ = '0.1.0'

import os as _os

# The two-slot pipeline (pipeline.DetectPipeline) runs slots + consumer + RCCL + the default stream: more streams than the HIP
# runtime's default of 4 hardware queues, and streams that share a queue serialise (-1.8 % images/s measured).  The runtime reads
# this variable when it initialises, so it is set here, at import, unless the application chose a value itself; if HIP was
# initialised before this import it has no effect and DetectPipeline warns.
import sys as _sys

_HW_QUEUES_PRESET = 'GPU_MAX_HW_QUEUES' in _os.environ            # the application chose a value itself
_t = _sys.modules.get('torch')
_HIP_UP_AT_IMPORT = bool(_t is not None and _t.cuda.is_initialized())   # then the setdefault below comes too late
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
