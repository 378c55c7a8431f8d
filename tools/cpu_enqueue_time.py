#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one batch (ron_detect through the pipeline)?  If that approaches the GPU time per
batch the bench is host-bound.  python tools/cpu_enqueue_time.py [full|reducedfc|ssd512] [batch] [dtype]"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory
from ron_tensorflow_amd.pipeline import DetectPipeline

variant = sys.argv[1] if len(sys.argv) > 1 else 'full'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dtype = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
if variant == 'ssd512':
    net = nets_factory.get_network('ssd_512_vgg')(dtype=dtype, max_batch=batch, fuse_pools=True)
    net.load_weights(W.ssd_synthetic_weights(seed=5))
    args = dict(select_threshold=0.01, nms_threshold=0.45)
else:
    net = nets_factory.get_network('ron_320_vgg')(variant=variant, dtype=dtype, max_batch=batch, fuse_pools=True)
    net.load_weights(W.synthetic_weights(variant, seed=1))
    args = dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45)
x = torch.from_numpy(W.synthetic_images(batch, seed=3, img_shape=net.params.img_shape)).cuda()
pipe = DetectPipeline(net, slots=2, max_queued=0)           # no host flow control: the pure enqueue cost
for _ in range(6):
    pipe.submit(x, **args)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K):
    pipe.submit(x, **args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s %s batch %d: host enqueue %.3f ms / batch, GPU %.3f ms / batch (two slots)' % (variant, dtype, batch, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
