#!/usr/bin/env python3
"""eval_ron_network.py-style driver on the MI355X path.

Makes exactly the calls of the reference driver (eval_ron_network.py:148-152, :204-210, :226-236) with the same
flag names and defaults (:64-123), on synthetic pre-whitened tensors (or an .npy batch / .npz weights the user
supplies), since the reference's dataset, checkpoint and metric plumbing is out of scope.
Prints 'Time spent per BATCH' like the reference (:365-366)."""
import argparse
import time

import numpy as np
import torch

from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--select_threshold', type=float, default=0.01)      # eval_ron_network.py:64-65
    ap.add_argument('--objectness_thres', type=float, default=0.03)      # :66-67
    ap.add_argument('--select_top_k', type=int, default=200)             # :68-69
    ap.add_argument('--keep_top_k', type=int, default=100)               # :70-71
    ap.add_argument('--nms_threshold', type=float, default=0.4)          # :72-73
    ap.add_argument('--num_classes', type=int, default=21)               # :92
    ap.add_argument('--batch_size', type=int, default=1)                 # :93-94
    ap.add_argument('--max_num_batches', type=int, default=4)
    ap.add_argument('--model_name', default='ron_320_vgg')               # :116-117
    ap.add_argument('--checkpoint_path', default='', help='.npz of TF variables (weights.save_npz); synthetic if empty')
    ap.add_argument('--images', default='', help='.npy [N,320,320,3] pre-whitened float32; synthetic if empty')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--variant', default='reducedfc', help="what RONNet.net builds in the reference (nets/ron_vgg_320.py:144)")
    FLAGS = ap.parse_args()

    # Get the RON network and its anchors.                                eval_ron_network.py:147-152
    ron_class = nets_factory.get_network(FLAGS.model_name)
    ron_params = ron_class.default_params._replace(num_classes=FLAGS.num_classes)
    ron_net = ron_class(ron_params, variant=FLAGS.variant, dtype=FLAGS.dtype, max_batch=FLAGS.batch_size)
    ron_shape = ron_net.params.img_shape
    ron_anchors = ron_net.anchors(ron_shape)
    weights = W.load_npz(FLAGS.checkpoint_path) if FLAGS.checkpoint_path else W.synthetic_weights(FLAGS.variant, FLAGS.num_classes)
    ron_net.load_weights(weights)
    data = np.load(FLAGS.images) if FLAGS.images else W.synthetic_images(FLAGS.batch_size * FLAGS.max_num_batches)

    times = []
    for i in range(0, min(len(data), FLAGS.batch_size * FLAGS.max_num_batches), FLAGS.batch_size):
        b_image = torch.from_numpy(np.ascontiguousarray(data[i:i + FLAGS.batch_size])).to(ron_net.device)
        torch.cuda.synchronize()
        start = time.time()
        with ron_net.arg_scope(weight_decay=0.0005, is_training=False, data_format='NHWC'):                  # :204-208
            predictions, logits, objness_pred, objness_logits, localisations, end_points = \
                ron_net.net(b_image, is_training=False, end_points=())                                          # :209-210
        localisations = ron_net.bboxes_decode(localisations, ron_anchors)                                      # :226
        filtered_predictions = [(objness > FLAGS.objectness_thres).to(torch.float32) * predictions[k]
                                for k, objness in enumerate(objness_pred)]                                      # :227-229
        rscores, rbboxes = ron_net.detected_bboxes(filtered_predictions, localisations,
                                                   select_threshold=FLAGS.select_threshold,
                                                   nms_threshold=FLAGS.nms_threshold,
                                                   clipping_bbox=[0., 0., 1., 1.],
                                                   top_k=FLAGS.select_top_k, keep_top_k=FLAGS.keep_top_k)     # :230-236
        torch.cuda.synchronize()
        times.append(time.time() - start)
        kept = sum(int((v > 0).sum().item()) for v in rscores.values())
        print('batch %d: %d detections over %d classes' % (i // FLAGS.batch_size, kept, len(rscores)))
    print('Time spent per BATCH: %.3f seconds.' % (sum(times[1:]) / max(len(times) - 1, 1)))


if __name__ == '__main__':
    main()
