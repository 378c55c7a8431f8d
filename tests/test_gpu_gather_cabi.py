"""GPU: `ron_gather_records` - the path's one collective (ncclAllGather of the detection records) from the C ABI.

No second GPU is visible to the tests, so the communicator has one rank (like tests/test_gpu_torchrun.py's process group): created
here through ctypes against the RCCL that is loaded in this process (the one PyTorch ships, soname librccl.so.1), handed to the
library as a void*.  The entry point resolves ncclAllGather from that same already-loaded library (csrc/gather.cpp)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]          # rccl.h: NCCL_UNIQUE_ID_BYTES


@pytest.fixture(scope='module')
def comm():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.init()
    paths = glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so*')) + ['/opt/rocm/lib/librccl.so.1']
    rccl = C.CDLL(paths[0], mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    c = C.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(c), 1, uid, 0) == 0 and c.value
    yield c
    rccl.ncclCommDestroy(c)


def test_gather_records_one_rank(comm):
    from ron_tensorflow_amd import _lib, ops, parallel
    dev = torch.device('cuda:0')
    n, k = 3, 400
    rs = np.random.RandomState(0)
    det = ops.DetectionBuffers(n, k, dev)
    count = np.array([5, 0, 400], np.int32)
    live = (np.arange(k)[None, :] < count[:, None])                      # rows past an image's count are zero (ron_detect pads with zeros)
    det.count = torch.from_numpy(count).to(dev)
    det.classes = torch.from_numpy((rs.randint(1, 21, (n, k)) * live).astype(np.int32)).to(dev)
    det.scores = torch.from_numpy((rs.rand(n, k) * live).astype(np.float32)).to(dev)
    det.bboxes = torch.from_numpy((rs.rand(n, k, 4) * live[..., None]).astype(np.float32)).to(dev)
    det.anchor_index = torch.from_numpy((rs.randint(0, 21250, (n, k)) * live).astype(np.int32)).to(dev)
    rec = parallel.pack_detections(det)                                  # ron_pack_records
    gathered = torch.zeros((1, n, k + 1, parallel.RECORD_WIDTH), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    _lib.check(_lib.lib().ron_gather_records(_lib.ptr(rec), n, k, _lib.ptr(gathered), comm, C.c_void_p(s.cuda_stream)))
    s.synchronize()
    assert torch.equal(gathered[0], rec)
    # in place: this rank's slice of `gathered` as the send buffer
    gathered[0].copy_(rec * 2)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().ron_gather_records(C.c_void_p(gathered.data_ptr()), n, k, _lib.ptr(gathered), comm, C.c_void_p(s.cuda_stream)))
    s.synchronize()
    assert torch.equal(gathered[0], rec * 2)
    cl, sc, bb, ai, cnt = parallel.unpack_records(gathered[0] / 2)
    assert torch.equal(cnt, det.count) and torch.equal(cl, det.classes) and torch.equal(ai, det.anchor_index)


def test_gather_records_argument_errors(comm):
    from ron_tensorflow_amd import _lib
    lib = _lib.lib()
    assert lib.ron_gather_records(None, 1, 400, None, comm, None) == -1
    x = torch.zeros((1, 401, 7), device='cuda:0')
    assert lib.ron_gather_records(_lib.ptr(x), 0, 400, _lib.ptr(x), comm, None) == -1
    assert lib.ron_gather_records(_lib.ptr(x), 1, 400, _lib.ptr(x), None, None) == -1
    assert b'NULL argument' in lib.ron_last_error()
