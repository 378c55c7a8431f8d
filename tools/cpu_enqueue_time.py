#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one batch (ron_detect through the pipeline)?  If that approaches the GPU time per
batch the bench is host-bound.  python tools/cpu_enqueue_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory
from ron_tensorflow_amd.pipeline import DetectPipeline

net = nets_factory.get_network('ron_320_vgg')(variant='full', dtype='bf16', max_batch=32, fuse_pools=True)
net.load_weights(W.synthetic_weights('full', seed=1))
x = torch.from_numpy(W.synthetic_images(32, seed=3)).cuda()
pipe = DetectPipeline(net, slots=2)
for _ in range(6):
    pipe.submit(x)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K):
    pipe.submit(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.3f ms / batch, total %.3f ms / batch' % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
