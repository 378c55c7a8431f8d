#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own numpy code.

Runs only in the build container (needs /root/reference).  It imports

  * ``nets/np_methods.py``  (numpy only)                      -> G2, G3, G4, G5 (SSD-512 pipeline)
  * ``nets/ron_vgg_320.py`` under a stubbed ``tensorflow``     -> G1 (anchor grids)
  * ``convert_pytorch_vgg.py`` (its torch ``VGG16`` / ``vgg``)  -> G8 (the conv backbone, every layer, 320^2 and 512^2)

and stores inputs (or the seed that regenerates them) together with the outputs the
reference produced.  Nothing of the reference's source text is stored: the .npz
files hold arrays only.  Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from oracle import np_post, synth  # noqa: E402  (only for softmax/gate, which are TF ops in the reference)


def load_np_methods():
    spec = importlib.util.spec_from_file_location('ref_np_methods', os.path.join(REF, 'nets', 'np_methods.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_ron():
    """Import nets/ron_vgg_320.py with tensorflow replaced by MagicMocks (anchor code is numpy)."""
    import importlib.abc
    import importlib.machinery

    class _Stub(mock.MagicMock):
        __path__ = []          # looks like a package, so sub-imports are attempted
        add_arg_scope = staticmethod(lambda f: f)   # decorators must stay pass-through

    class _TFFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path, target=None):
            if fullname == 'tensorflow' or fullname.startswith('tensorflow.'):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Stub(name=spec.name)

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _TFFinder())
    sys.path.insert(0, REF)
    from nets import ron_vgg_320  # noqa
    return ron_vgg_320


def g1_anchors(ron):
    net = ron.RONNet()
    layers = net.anchors((320, 320))
    out = {}
    for i, (y, x, h, w) in enumerate(layers):
        out['y%d' % i], out['x%d' % i], out['h%d' % i], out['w%d' % i] = y, x, h, w
    np.savez_compressed(os.path.join(HERE, 'g1_anchors_ron320.npz'), **out)
    return layers


def g5_anchors_ssd(ron_module):
    """SSD-512 anchors from the reference's ssd_anchors_all_layers (nets/ssd_vgg_512.py:286-358)."""
    from nets import ssd_vgg_512
    net = ssd_vgg_512.SSDNet()
    layers = net.anchors((512, 512))
    out = {}
    for i, (y, x, h, w) in enumerate(layers):
        out['y%d' % i], out['x%d' % i], out['h%d' % i], out['w%d' % i] = y, x, h, w
    np.savez_compressed(os.path.join(HERE, 'g5_anchors_ssd512.npz'), **out)
    return layers


G5_CASES = [
    # name, seed, bg, cls_scale, select_thr, nms_thr
    ('ssd_real_s0', 40, 8.0, 1.0, 0.01, 0.45),
    ('ssd_real_s1', 41, 7.0, 1.0, 0.01, 0.45),
    ('ssd_dense_s2', 42, 4.0, 1.0, 0.01, 0.45),
    ('ssd_thr50_s3', 43, 6.0, 3.0, 0.5, 0.40),
    ('ssd_empty_s4', 44, 30.0, 1.0, 0.01, 0.45),
]


def g5_pipeline_ssd(npm, anchors):
    """The reference's np_methods pipeline on the 24 564 SSD-512 anchors (7 scales, 4 / 6 anchors per cell, no
    objectness gate; SURVEY.md 8c "G5"): ssd_bboxes_select -> bboxes_clip -> bboxes_sort(400) -> bboxes_nms ->
    bboxes_resize (nets/np_methods.py:100-131, 137-150, 153-183, 229-242), on seeded head tensors."""
    out = {}
    names = []
    for name, seed, bg, scale, thr, nms in G5_CASES:
        cls, loc = synth.ssd_head_tensors(seed, batch=1, bg=bg, cls_scale=scale)
        pred = [np_post.softmax_last(x) for x in cls]
        rbbox_img = np.array([0., 0., 1., 1.], dtype=np.float32)
        c, s, b = npm.ssd_bboxes_select(pred, loc, anchors, select_threshold=thr, img_shape=(512, 512), num_classes=21, decode=True)
        n_cand = c.shape[0]
        b = npm.bboxes_clip(rbbox_img, b)
        c, s, b = npm.bboxes_sort(c, s, b, top_k=400)
        srt = (c.copy(), s.copy(), b.copy())
        assert len(np.unique(s)) == len(s), name          # the reference's argsort is unstable: tie-free cases only
        c, s, b = npm.bboxes_nms(c, s, b, nms_threshold=nms)
        b = npm.bboxes_resize(rbbox_img, b)
        names.append(name)
        out[name + '/params'] = np.array([seed, bg, scale, thr, nms], dtype=np.float64)
        out[name + '/n_cand'] = np.int64(n_cand)
        out[name + '/n_sorted'] = np.int64(srt[0].shape[0])
        out[name + '/classes'] = c.astype(np.int64)
        out[name + '/scores'] = s.astype(np.float32)
        out[name + '/bboxes'] = b.astype(np.float32).reshape(-1, 4)
        out[name + '/sorted_classes'] = srt[0].astype(np.int64)
        out[name + '/sorted_scores'] = srt[1].astype(np.float32)
        print('%-13s cand=%6d sorted=%3d kept=%3d' % (name, n_cand, srt[0].shape[0], c.shape[0]))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'g5_pipeline_ssd512.npz'), **out)


def g6_voc_ap():
    """AP of seeded precision/recall curves through the reference's numpy voc_ap (datasets/voc_eval.py:130-162)."""
    sys.modules.setdefault('cv2', mock.MagicMock(name='cv2'))
    from datasets import voc_eval
    out = {}
    rs = np.random.RandomState(33)
    for k in range(6):
        n = [5, 40, 200, 1000, 3, 64][k]
        n_gt = [4, 30, 120, 400, 10, 64][k]
        tp = rs.rand(n) < [0.7, 0.5, 0.4, 0.3, 1.0, 0.0][k]
        fp = ~tp
        ctp, cfp = np.cumsum(tp.astype(np.float64)), np.cumsum(fp.astype(np.float64))
        rec = ctp / n_gt
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out['rec%d' % k], out['prec%d' % k] = rec, prec
        out['ap07_%d' % k] = np.float64(voc_eval.DetectorEvalPascal.voc_ap(None, rec, prec, use_07_metric=True))
        out['ap12_%d' % k] = np.float64(voc_eval.DetectorEvalPascal.voc_ap(None, rec, prec, use_07_metric=False))
    np.savez_compressed(os.path.join(HERE, 'g6_voc_ap.npz'), **out)


def g7_voc_eval():
    """The reference's numpy VOC evaluation (datasets/voc_eval.py:164-295) run on a small synthetic VOC tree written to a temp
    directory: XML annotations + per-class detection files in, (recall, precision, AP07, AP12) per class out.  The arrays the
    files held (as parsed back by the reference) are stored as the inputs."""
    import tempfile
    sys.modules.setdefault('cv2', mock.MagicMock(name='cv2'))
    from datasets import voc_eval
    rs = np.random.RandomState(77)
    classes = ['aeroplane', 'bicycle', 'bird']
    n_img = 12
    out = {}
    with tempfile.TemporaryDirectory() as root:
        voc = os.path.join(root, 'VOC2007')
        for d in ('Annotations', 'ImageSets/Main', 'JPEGImages'):
            os.makedirs(os.path.join(voc, d))
        names = ['%06d' % (i + 1) for i in range(n_img)]
        with open(os.path.join(voc, 'ImageSets', 'Main', 'test.txt'), 'w') as f:
            f.write('\n'.join(names) + '\n')
        gt = []                                         # (image, class, difficult, xmin, ymin, xmax, ymax) as written to the XML
        for i, name in enumerate(names):
            objs = []
            for _ in range(rs.randint(0, 5)):
                c = rs.randint(0, 3)
                x1, y1 = rs.randint(1, 300), rs.randint(1, 200)
                w, h = rs.randint(20, 150), rs.randint(20, 150)
                diff = int(rs.rand() < 0.25)
                objs.append((c, diff, x1, y1, x1 + w, y1 + h))
                gt.append((i, c, diff, x1, y1, x1 + w, y1 + h))
            xml = '<annotation>' + ''.join(
                '<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>'
                '<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>'
                % (classes[c], d, a, b, cc, dd) for (c, d, a, b, cc, dd) in objs) + '</annotation>'
            with open(os.path.join(voc, 'Annotations', name + '.xml'), 'w') as f:
                f.write(xml)
        out['gt'] = np.array(gt, np.int64).reshape(-1, 7)
        ev = voc_eval.DetectorEvalPascal(root, root, set_type='test', output_dir=os.path.join(root, 'out_{}'))
        for ci, cname in enumerate(classes):
            # detections: jittered ground truth of this class (some duplicated), boxes of other classes, random boxes
            dets = []
            for (i, c, d, a, b, cc, dd) in gt:
                reps = rs.randint(0, 3) if c == ci else (1 if rs.rand() < 0.3 else 0)
                for _ in range(reps):
                    j = rs.randint(-25, 26, 4) if rs.rand() < 0.3 else rs.randint(-6, 7, 4)
                    dets.append((i, a + j[0], b + j[1], cc + j[2], dd + j[3]))
            for _ in range(10):
                i = rs.randint(0, n_img)
                x1, y1 = rs.randint(1, 300), rs.randint(1, 200)
                dets.append((i, x1, y1, x1 + rs.randint(10, 120), y1 + rs.randint(10, 120)))
            scores = rs.permutation(np.arange(1, 1000))[:len(dets)] / 1000.0        # unique at the 3 decimals the file keeps
            detfile = os.path.join(root, 'det_test_%s.txt' % cname)
            with open(detfile, 'w') as f:
                for (i, a, b, cc, dd), sc in zip(dets, scores):
                    f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(names[i], sc, a, b, cc, dd))
            cache = os.path.join(root, 'cache_%d' % ci)
            for use07 in (True, False):
                rec, prec, ap = ev.voc_eval(os.path.join(root, 'det_test_{}.txt'), cname, cache, ovthresh=0.5, use_07_metric=use07)
                out['rec_%d' % ci], out['prec_%d' % ci] = np.asarray(rec, np.float64), np.asarray(prec, np.float64)
                out['ap%s_%d' % ('07' if use07 else '12', ci)] = np.float64(ap)
            out['det_%d' % ci] = np.array([(i, a, b, cc, dd) for (i, a, b, cc, dd) in dets], np.float64).reshape(-1, 5)
            out['score_%d' % ci] = np.array([float('%.3f' % sc) for sc in scores], np.float64)
        # an empty detection file: the reference returns -1 for everything
        with open(os.path.join(root, 'det_test_empty.txt'), 'w') as f:
            pass
    np.savez_compressed(os.path.join(HERE, 'g7_voc_eval.npz'), **out)


def g2_decode(npm, anchors):
    out = {'seed': np.int64(11)}
    rs = np.random.RandomState(11)
    for i, a in enumerate(anchors):
        hh, ww = a[0].shape[:2]
        loc = rs.randn(1, hh, ww, 10, 4).astype(np.float32)
        out['loc%d' % i] = loc
        out['dec%d' % i] = npm.ssd_bboxes_decode(loc, a)
    np.savez_compressed(os.path.join(HERE, 'g2_decode.npz'), **out)


def run_reference_pipeline(npm, predictions, localisations, anchors, select_threshold, top_k, nms_threshold,
                           bbox_img=(0., 0., 1., 1.)):
    """notebooks/ssd_notebook.ipynb cell 8, batch 1."""
    rbbox_img = np.asarray(bbox_img, dtype=np.float32)
    c, s, b = npm.ssd_bboxes_select(predictions, localisations, anchors, select_threshold=select_threshold,
                                    img_shape=(320, 320), num_classes=21, decode=True)
    n_cand = c.shape[0]
    sel = (c.copy(), s.copy(), b.copy())
    b = npm.bboxes_clip(rbbox_img, b)
    c, s, b = npm.bboxes_sort(c, s, b, top_k=top_k)
    srt = (c.copy(), s.copy(), b.copy())
    c, s, b = npm.bboxes_nms(c, s, b, nms_threshold=nms_threshold)
    b = npm.bboxes_resize(rbbox_img, b)
    return dict(sel=sel, srt=srt, out=(c, s, b), n_cand=n_cand)


G3_CASES = [
    # name, seed, bg, ob, cls_scale, select_thr, nms_thr
    ('real_s0', 0, 8.0, -4.0, 1.0, 0.01, 0.45),
    ('real_s1', 1, 8.0, -4.0, 1.0, 0.01, 0.45),
    ('real_s2', 2, 8.0, -4.0, 1.0, 0.01, 0.40),
    ('real_s3', 3, 8.0, -3.0, 1.0, 0.01, 0.45),
    ('mid_s4', 4, 7.0, -3.0, 1.0, 0.01, 0.45),
    ('dense_s5', 5, 4.0, -2.0, 1.0, 0.01, 0.45),
    ('thr50_s6', 6, 8.0, -2.0, 3.0, 0.5, 0.45),
    ('thr50_s7', 7, 6.0, -1.0, 3.0, 0.5, 0.40),
    ('empty_s8', 8, 30.0, -30.0, 1.0, 0.01, 0.45),
    # select_threshold 0: the "score > no-label" branch of ssd_bboxes_select_layer (np_methods.py:82-89): one candidate per
    # anchor, arg-max over all classes, kept when the class is not background
    ('argmax_dense_s9', 9, 2.0, 1.0, 1.0, 0.0, 0.45),
    ('argmax_sparse_s10', 10, 4.5, -2.0, 1.0, 0.0, 0.45),
    ('argmax_empty_s11', 11, 30.0, -30.0, 1.0, 0.0, 0.45),
]


def g3_pipeline(npm, anchors):
    out = {}
    names = []
    for name, seed, bg, ob, scale, thr, nms in G3_CASES:
        cls, obj, loc = synth.head_tensors(seed, batch=1, bg=bg, ob=ob, cls_scale=scale)
        pred = [np_post.softmax_last(x) for x in cls]
        objp = [np_post.objectness_from_logits(x) for x in obj]
        gated = np_post.objectness_gate(pred, objp, 0.03)
        r = run_reference_pipeline(npm, gated, loc, anchors, thr, 400, nms)
        sc_sorted = r['srt'][1]
        # the reference's argsort is unstable: only tie-free cases are pinned row by row
        assert len(np.unique(r['sel'][1])) == len(r['sel'][1]) or len(np.unique(sc_sorted)) == len(sc_sorted), name
        names.append(name)
        out[name + '/params'] = np.array([seed, bg, ob, scale, thr, nms], dtype=np.float64)
        out[name + '/n_cand'] = np.int64(r['n_cand'])
        out[name + '/n_sorted'] = np.int64(r['srt'][0].shape[0])
        out[name + '/classes'] = r['out'][0].astype(np.int64)
        out[name + '/scores'] = r['out'][1].astype(np.float32)
        out[name + '/bboxes'] = r['out'][2].astype(np.float32).reshape(-1, 4)
        out[name + '/sorted_classes'] = r['srt'][0].astype(np.int64)
        out[name + '/sorted_scores'] = r['srt'][1].astype(np.float32)
        print('%-10s cand=%6d sorted=%3d kept=%3d' % (name, r['n_cand'], r['srt'][0].shape[0], r['out'][0].shape[0]))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'g3_pipeline.npz'), **out)


def g4_edge(npm):
    """Hand-built inputs to the individual np_methods steps (sort / clip / nms / resize)."""
    out = {}
    f = np.float32
    # (a) ties: np.argsort(-scores) is unstable, so rows inside a run of equal scores come out in an
    #     unspecified order (tests compare such runs as sets); the NMS result of this case does not
    #     depend on that order.  Identical boxes of different class survive.
    classes = np.array([3, 3, 5, 3, 5, 7, 7, 7], dtype=np.int64)
    scores = np.array([.5, .9, .5, .5, .9, .25, .25, .75], dtype=f)
    boxes = np.array([[.1, .1, .5, .5], [.1, .1, .5, .5], [.1, .1, .5, .5], [.12, .1, .5, .52],
                      [.6, .6, .9, .9], [.0, .0, .2, .2], [.0, .0, .2, .2], [.01, .0, .2, .21]], dtype=f)
    c, s, b = npm.bboxes_sort(classes, scores, boxes, top_k=400)
    out['ties/in_classes'], out['ties/in_scores'], out['ties/in_bboxes'] = classes, scores, boxes
    out['ties/sorted_classes'], out['ties/sorted_scores'], out['ties/sorted_bboxes'] = c, s, b
    c2, s2, b2 = npm.bboxes_nms(c, s, b, nms_threshold=0.45)
    out['ties/nms_classes'], out['ties/nms_scores'], out['ties/nms_bboxes'] = c2, s2, b2
    # (b) zero-area boxes: IoU = 0/0 = NaN -> "NaN < thr" is False -> suppressed when same class
    classes = np.array([1, 1, 1, 2, 1], dtype=np.int64)
    scores = np.array([.9, .8, .7, .6, .5], dtype=f)
    boxes = np.array([[.3, .3, .3, .3], [.3, .3, .3, .3], [.2, .2, .6, .6], [.3, .3, .3, .3], [.2, .2, .6, .6]], dtype=f)
    with np.errstate(all='ignore'):
        c2, s2, b2 = npm.bboxes_nms(classes, scores, boxes, nms_threshold=0.45)
    out['zero/in_classes'], out['zero/in_scores'], out['zero/in_bboxes'] = classes, scores, boxes
    out['zero/nms_classes'], out['zero/nms_scores'], out['zero/nms_bboxes'] = c2, s2, b2
    # (c) clip: boxes outside / inverted after clip (np version has no repair)
    boxes = np.array([[-.2, -.1, .5, .6], [.4, .5, 1.3, 1.2], [1.1, 1.2, 1.5, 1.6], [-.5, -.5, -.1, -.2], [.2, .2, .8, .8]], dtype=f)
    out['clip/in_bboxes'] = boxes
    out['clip/out_bboxes'] = npm.bboxes_clip(np.array([0., 0., 1., 1.], dtype=f), boxes)
    out['clip/ref2'] = np.array([.1, .2, .7, .9], dtype=f)
    out['clip/out_bboxes2'] = npm.bboxes_clip(out['clip/ref2'], boxes)
    out['resize/out_bboxes2'] = npm.bboxes_resize(out['clip/ref2'], out['clip/out_bboxes2'])
    # (d) inverted boxes through nms (negative extents -> negative areas)
    classes = np.array([4, 4, 4], dtype=np.int64)
    scores = np.array([.9, .8, .7], dtype=f)
    boxes = np.array([[.5, .5, .2, .2], [.5, .5, .2, .2], [.1, .1, .6, .6]], dtype=f)
    with np.errstate(all='ignore'):
        c2, s2, b2 = npm.bboxes_nms(classes, scores, boxes, nms_threshold=0.45)
    out['inv/in_classes'], out['inv/in_scores'], out['inv/in_bboxes'] = classes, scores, boxes
    out['inv/nms_classes'], out['inv/nms_scores'], out['inv/nms_bboxes'] = c2, s2, b2
    # (e) IoU exactly at the threshold: [0,0,1,1] vs [0,0,1,.45] -> IoU = .45 (f32) ; "<" keeps only below
    classes = np.array([9, 9, 9], dtype=np.int64)
    scores = np.array([.9, .8, .7], dtype=f)
    boxes = np.array([[0, 0, 1, 1], [0, 0, 1, .45], [0, 0, 1, .44]], dtype=f)
    c2, s2, b2 = npm.bboxes_nms(classes, scores, boxes, nms_threshold=0.45)
    out['thr/in_classes'], out['thr/in_scores'], out['thr/in_bboxes'] = classes, scores, boxes
    out['thr/nms_classes'], out['thr/nms_scores'], out['thr/nms_bboxes'] = c2, s2, b2
    out['thr/iou'] = npm.bboxes_jaccard(boxes[0], boxes[1:])
    # (f) select at exactly the threshold (strict >) and 400/401 candidate cut
    rs = np.random.RandomState(21)
    pred = np.zeros((1, 1, 1, 500, 21), dtype=f)
    sc = np.sort(rs.uniform(.02, .99, 401).astype(f))[::-1]
    assert len(np.unique(sc)) == 401
    order = rs.permutation(401)
    pred[0, 0, 0, order, 1 + (order % 20)] = sc
    pred[0, 0, 0, 450, 5] = f(0.01)          # exactly thr -> not selected
    pred[0, 0, 0, 451, 5] = np.nextafter(f(0.01), f(1))  # just above -> selected (402 candidates)
    loc = rs.uniform(0, 1, (1, 1, 1, 500, 4)).astype(f)
    loc[..., 2:] += loc[..., :2]
    c, s, b = npm.ssd_bboxes_select_layer(pred, loc, None, select_threshold=0.01, decode=False)
    out['cut/pred'], out['cut/boxes'] = pred, loc
    out['cut/sel_classes'], out['cut/sel_scores'], out['cut/sel_bboxes'] = c, s, b
    c, s, b = npm.bboxes_sort(c, s, b, top_k=400)
    out['cut/sorted_classes'], out['cut/sorted_scores'], out['cut/sorted_bboxes'] = c, s, b
    np.savez_compressed(os.path.join(HERE, 'g4_edge.npz'), **out)


def load_ref_torch_vgg():
    """The reference's only EXECUTABLE conv stack: ``VGG16`` and ``vgg(cfg, i)`` of convert_pytorch_vgg.py:13-58 (plain torch:
    13 convs, pools M M C M, pool5 3x3 s1, conv6 3x3 dilation 6, conv7 1x1 = nets/ssd_vgg_512.py:364-400 up to block7 and
    conv1_1 .. conv5_3 of nets/ron_vgg_320.py:454-475).  The file imports keras / mmdnn / pytorch2keras at module level for its
    converter functions (not installed, not needed by the two definitions): those three are stubbed in sys.modules and the
    module body is executed up to its ``__main__`` guard, in memory - nothing of its text is stored."""
    for name in ('keras', 'mmdnn', 'mmdnn.conversion', 'mmdnn.conversion.keras', 'mmdnn.conversion.keras.keras2_parser',
                 'pytorch2keras', 'pytorch2keras.converter'):
        sys.modules.setdefault(name, mock.MagicMock(name=name))
    with open(os.path.join(REF, 'convert_pytorch_vgg.py')) as f:
        text = f.read()
    text = text[:text.index("if __name__ == '__main__':")]
    mod = types.ModuleType('ref_convert_pytorch_vgg')
    mod.__dict__['os'] = os            # VGG16.load_weights (unused here) expects it
    exec(compile(text, os.path.join(REF, 'convert_pytorch_vgg.py'), 'exec'), mod.__dict__)
    return mod


def g8_vgg_backbone():
    """Golden G8: the reference's torch VGG16 run on seeded weights and seeded 320^2 / 512^2 images, every one of its 35 modules
    captured.  Stored per tensor (NCHW -> NHWC): the values at synth.g8_sample_index rows x columns (all channels) and float64
    sum / sum of squares over the WHOLE tensor.  Weights and images are regenerated from the seeds (oracle/synth.py)."""
    import torch
    ref = load_ref_torch_vgg()
    seed_w = 80
    model = ref.VGG16(ref.vgg(list(synth.VGG_CFG), 3))
    assert len(model.vgg) == 35
    convs = [m for m in model.vgg if isinstance(m, torch.nn.Conv2d)]
    wts = synth.vgg_backbone_weights_oihw(seed_w)
    assert len(convs) == len(wts) == 15
    with torch.no_grad():
        for m, (w, b) in zip(convs, wts):
            assert tuple(m.weight.shape) == w.shape
            m.weight.copy_(torch.from_numpy(w))
            m.bias.copy_(torch.from_numpy(b))
    model.eval()
    # module outputs in order; a Conv2d's tensor is overwritten by the in-place ReLU behind it, so taps are taken after ReLU / pool
    taps = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: taps.append(out.detach().clone()))
             for m in model.vgg if not isinstance(m, torch.nn.Conv2d)]
    out = {'seed_weights': np.int64(seed_w)}
    for size, seed_x in ((320, 81), (512, 82)):
        img = synth.vgg_backbone_image(seed_x, size)
        del taps[:]
        with torch.no_grad():
            y = model(torch.from_numpy(img).permute(0, 3, 1, 2).contiguous())
        assert len(taps) == len(synth.VGG_TAPS) == 20 and y.shape[1] == 1024
        out['seed_image_%d' % size] = np.int64(seed_x)
        for name, t in zip(synth.VGG_TAPS, taps):
            a = t.permute(0, 2, 3, 1).contiguous().numpy()
            iy, ix = synth.g8_sample_index(a.shape[1]), synth.g8_sample_index(a.shape[2])
            out['%d/%s/shape' % (size, name)] = np.array(a.shape, dtype=np.int64)
            out['%d/%s/sample' % (size, name)] = a[:, iy][:, :, ix].copy()
            out['%d/%s/sum' % (size, name)] = np.array([a.sum(dtype=np.float64), (a.astype(np.float64) ** 2).sum()])
    for h in hooks:
        h.remove()
    # ---- second part: the same reference function with batch_norm=True (convert_pytorch_vgg.py:47-48: Conv2d -> BatchNorm2d ->
    # ReLU), in eval mode on seeded running statistics: the only executable "conv + inference BatchNorm + ReLU" of the reference
    # (eps 1e-5 = torch's default = ron_arg_scope's batch_norm_epsilon, nets/ron_vgg_320.py:601).  Small: stored in full.
    layers = ref.vgg(list(synth.VGG_BN_CFG), 3, batch_norm=True)[:-5]          # drop the pool5 / conv6 / conv7 tail vgg() appends
    seed_bn = 83
    params = synth.vgg_bn_params(seed_bn)
    convs = [m for m in layers if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in layers if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(convs) == len(bns) == len(params) == 2
    with torch.no_grad():
        for cv, bn, (w, gamma, beta, mean, var) in zip(convs, bns, params):
            cv.weight.copy_(torch.from_numpy(w)); cv.bias.zero_()
            bn.weight.copy_(torch.from_numpy(gamma)); bn.bias.copy_(torch.from_numpy(beta))
            bn.running_mean.copy_(torch.from_numpy(mean)); bn.running_var.copy_(torch.from_numpy(var))
            assert bn.eps == 1e-5
    net = torch.nn.Sequential(*layers).eval()
    img = synth.vgg_backbone_image(84, 24)
    acts, x = [], torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for m in net:
            x = m(x)
            if isinstance(m, (torch.nn.ReLU, torch.nn.MaxPool2d)):
                acts.append(x.clone().permute(0, 2, 3, 1).contiguous().numpy())
    assert len(acts) == 3
    out['bn/seed_params'], out['bn/seed_image'] = np.int64(seed_bn), np.int64(84)
    out['bn/conv_bn_relu_1'], out['bn/pool'], out['bn/conv_bn_relu_2'] = acts
    np.savez_compressed(os.path.join(HERE, 'g8_vgg_backbone.npz'), **out)


def main():
    npm = load_np_methods()
    ron = load_ref_ron()
    anchors = g1_anchors(ron)
    ssd_anchors = g5_anchors_ssd(ron)
    g6_voc_ap()
    g7_voc_eval()
    g2_decode(npm, anchors)
    g3_pipeline(npm, anchors)
    g5_pipeline_ssd(npm, ssd_anchors)
    g4_edge(npm)
    g8_vgg_backbone()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith('.npz'):
            print(fn, os.path.getsize(os.path.join(HERE, fn)), 'bytes')


if __name__ == '__main__':
    main()
