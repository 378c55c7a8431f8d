#!/usr/bin/env python3
"""eval_ron_network.py-style driver on the MI355X path.

Makes exactly the calls of the reference driver (eval_ron_network.py:148-158, :204-210, :226-257, :289-335) with the
same flag names and defaults (:64-123): preprocess_for_eval -> net -> bboxes_decode -> objectness gate ->
detected_bboxes -> bboxes_matching_batch -> streaming TP/FP -> AP VOC07 / VOC12 -> mAP.  Data is synthetic (decoded
uint8 images of ragged sizes with random ground truth) or user supplied (.npy batch / .npz weights): the reference's
TFRecord dataset and checkpoint plumbing are out of scope.
Prints 'Time spent per BATCH' like the reference (:365-366)."""
import argparse
import time

import numpy as np
import torch

from ron_tensorflow_amd import metrics as tfe_metrics
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.preprocessing import ssd_vgg_preprocessing
from ron_tensorflow_amd.nets import nets_factory


def main(argv=None):
    """Runs the evaluation; `argv` None = the command line.  Returns what it printed as data: {'AP_VOC07/mAP', 'AP_VOC12/mAP',
    'seconds_per_batch', 'detections': per batch {class: (scores, bboxes) as numpy}} (tests/test_gpu_eval.py runs it in-process)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--select_threshold', type=float, default=0.01)      # eval_ron_network.py:64-65
    ap.add_argument('--objectness_thres', type=float, default=0.03)      # :66-67
    ap.add_argument('--select_top_k', type=int, default=200)             # :68-69
    ap.add_argument('--keep_top_k', type=int, default=100)               # :70-71
    ap.add_argument('--nms_threshold', type=float, default=0.4)          # :72-73
    ap.add_argument('--matching_threshold', type=float, default=0.5)
    ap.add_argument('--remove_difficult', type=int, default=0)
    ap.add_argument('--num_classes', type=int, default=21)               # :92
    ap.add_argument('--batch_size', type=int, default=1)                 # :93-94
    ap.add_argument('--max_num_batches', type=int, default=4)
    ap.add_argument('--model_name', default='ron_320_vgg')               # :116-117
    ap.add_argument('--checkpoint_path', default='',
                    help='TF V2 checkpoint prefix / directory, or .npz of TF variables (weights.save_npz); synthetic if empty')
    ap.add_argument('--checkpoint_model_scope', default=None)            # ron_eval.py:98-100
    ap.add_argument('--checkpoint_exclude_scopes', default=None)         # ron_eval.py:101-104
    ap.add_argument('--ignore_missing_vars', type=int, default=0)        # ron_eval.py:105-107
    ap.add_argument('--images', default='', help='.npy [N,320,320,3] pre-whitened float32; synthetic if empty')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--variant', default='reducedfc', help="what RONNet.net builds in the reference (nets/ron_vgg_320.py:144)")
    FLAGS = ap.parse_args(argv)

    # Get the RON network and its anchors.                                eval_ron_network.py:147-152
    ron_class = nets_factory.get_network(FLAGS.model_name)
    ron_params = ron_class.default_params._replace(num_classes=FLAGS.num_classes)
    ron_net = ron_class(ron_params, variant=FLAGS.variant, dtype=FLAGS.dtype, max_batch=FLAGS.batch_size)
    ron_shape = ron_net.params.img_shape
    ron_anchors = ron_net.anchors(ron_shape)
    if FLAGS.checkpoint_path:                                              # eval_ron_network.py:346-361
        ron_net.load_checkpoint(FLAGS.checkpoint_path, checkpoint_model_scope=FLAGS.checkpoint_model_scope,
                                checkpoint_exclude_scopes=FLAGS.checkpoint_exclude_scopes,
                                ignore_missing_vars=bool(FLAGS.ignore_missing_vars))
    else:
        ron_net.load_weights(W.synthetic_weights(FLAGS.variant, FLAGS.num_classes))
    n_total = FLAGS.batch_size * FLAGS.max_num_batches
    rs = np.random.RandomState(0)
    if FLAGS.images:
        data = np.load(FLAGS.images)
    else:       # decoded "JPEGs": uint8 RGB, PASCAL-like ragged sizes
        data = [rs.randint(0, 256, (int(rs.randint(200, 500)), int(rs.randint(200, 500)), 3)).astype(np.uint8) for _ in range(n_total)]
    # synthetic ground truth, padded to a fixed count per image like the reference's batching (:160-170)
    n_gt = 8
    g_labels = rs.randint(0, FLAGS.num_classes, (n_total, n_gt)).astype(np.int64)
    yx = rs.rand(n_total, n_gt, 2).astype(np.float32) * 0.6
    g_bboxes = np.concatenate([yx, yx + 0.1 + 0.3 * rs.rand(n_total, n_gt, 2).astype(np.float32)], -1)
    g_bboxes[g_labels == 0] = 0
    g_difficults = (rs.rand(n_total, n_gt) < 0.1).astype(np.int64)
    if FLAGS.remove_difficult:
        g_difficults[:] = 0
    labels = list(range(1, FLAGS.num_classes))
    tp_fp_metric = tfe_metrics.StreamingTpFp(labels)

    times, detections = [], []
    for i in range(0, min(len(data), n_total), FLAGS.batch_size):
        torch.cuda.synchronize()
        start = time.time()
        if FLAGS.images:
            b_image = torch.from_numpy(np.ascontiguousarray(data[i:i + FLAGS.batch_size])).to(ron_net.device)
        else:                                                                                                  # :153-158
            b_image = ssd_vgg_preprocessing.preprocess_for_eval_batch(data[i:i + FLAGS.batch_size], out_shape=ron_shape,
                                                                      resize=ssd_vgg_preprocessing.Resize.WARP_RESIZE,
                                                                      device=ron_net.device)
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
        sl = slice(i, i + FLAGS.batch_size)
        num_gbboxes, tp, fp = tfe_metrics.bboxes_matching_batch(rscores.keys(), rscores, rbboxes, g_labels[sl], g_bboxes[sl],
                                                                g_difficults[sl], matching_threshold=FLAGS.matching_threshold)  # :237-241
        tp_fp_metric.update(torch.stack([num_gbboxes[c] for c in labels], 1), torch.stack([tp[c] for c in labels], 1),
                            torch.stack([fp[c] for c in labels], 1), torch.stack([rscores[c] for c in labels], 1))  # :259-261
        torch.cuda.synchronize()
        times.append(time.time() - start)
        kept = sum(int((v > 0).sum().item()) for v in rscores.values())
        detections.append({c: (rscores[c].cpu().numpy(), rbboxes[c].cpu().numpy()) for c in labels})
        print('batch %d: %d detections over %d classes' % (i // FLAGS.batch_size, kept, len(rscores)))
    res = tfe_metrics.evaluate(tp_fp_metric)                                                                  # :289-335
    print('AP_VOC07/mAP %.6f  AP_VOC12/mAP %.6f  (synthetic weights and ground truth: plumbing check, not accuracy)'
          % (res['AP_VOC07/mAP'], res['AP_VOC12/mAP']))
    per_batch = sum(times[1:]) / max(len(times) - 1, 1)
    print('Time spent per BATCH: %.3f seconds.' % per_batch)
    ron_net.close()
    return {'AP_VOC07/mAP': res['AP_VOC07/mAP'], 'AP_VOC12/mAP': res['AP_VOC12/mAP'], 'seconds_per_batch': per_batch,
            'detections': detections}


if __name__ == '__main__':
    main()
