"""GPU: ron_bboxes_matching vs the oracle restatement of tfe.bboxes_matching_batch — tp / fp / counts bit-exact."""
import numpy as np
import pytest
import torch

from oracle import eval_metrics as em

pytestmark = pytest.mark.gpu


def _case(rs, n, nl, k, g, jitter=0.05):
    """Ground truth + detections derived from it (so overlaps around the 0.5 threshold and duplicates occur)."""
    gl = rs.randint(0, nl + 1, (n, g)).astype(np.int64)
    yx = rs.rand(n, g, 2).astype(np.float32) * 0.6
    hw = (rs.rand(n, g, 2).astype(np.float32) * 0.3 + 0.05)
    gb = np.concatenate([yx, yx + hw], -1).astype(np.float32)
    gb[gl == 0] = 0
    gd = (rs.rand(n, g) < 0.2).astype(np.int64)
    sc = np.sort(rs.rand(n, nl, k).astype(np.float32), -1)[..., ::-1].copy()
    bb = np.zeros((n, nl, k, 4), np.float32)
    for i in range(n):
        for l in range(nl):
            for j in range(k):
                src = gb[i, rs.randint(g)]
                bb[i, l, j] = src + rs.randn(4).astype(np.float32) * jitter * rs.rand()
    pad = max(1, k // 5)
    sc[..., -pad:] = 0
    bb[..., -pad:, :] = 0
    return sc, bb, gl, gb, gd


@pytest.mark.parametrize('n,nl,k,g', [(2, 20, 200, 12), (3, 5, 64, 1), (1, 3, 33, 64), (2, 4, 50, 100), (1, 2, 40, 256)])
def test_matching_vs_oracle(n, nl, k, g):
    from ron_tensorflow_amd import metrics
    rs = np.random.RandomState(n * 1000 + g)
    sc, bb, gl, gb, gd = _case(rs, n, nl, k, g)
    dev = torch.device('cuda:0')
    ngb, tp, fp = metrics.bboxes_matching(torch.from_numpy(sc).to(dev), torch.from_numpy(bb).to(dev), gl, gb, gd)
    ngb, tp, fp = ngb.cpu().numpy(), tp.cpu().numpy(), fp.cpu().numpy()
    labels = list(range(1, nl + 1))
    o_n, o_tp, o_fp = em.bboxes_matching_batch(labels, {c: sc[:, c - 1] for c in labels}, {c: bb[:, c - 1] for c in labels}, gl, gb, gd)
    assert tp.any() and fp.any()
    for c in labels:
        assert np.array_equal(ngb[:, c - 1], o_n[c])
        assert np.array_equal(tp[:, c - 1], o_tp[c]), 'tp class %d' % c
        assert np.array_equal(fp[:, c - 1], o_fp[c]), 'fp class %d' % c


def test_matching_dict_api_and_map():
    """Reference call sequence eval_ron_network.py:237-335 on synthetic detections: dict API -> streaming -> mAP."""
    from ron_tensorflow_amd import metrics
    rs = np.random.RandomState(11)
    n, nl, k, g = 4, 6, 80, 9
    dev = torch.device('cuda:0')
    labels = list(range(1, nl + 1))
    st = metrics.StreamingTpFp(labels)
    o_acc = {c: [0, [], [], []] for c in labels}
    for _ in range(2):
        sc, bb, gl, gb, gd = _case(rs, n, nl, k, g)
        d_s = {c: torch.from_numpy(sc[:, c - 1]).to(dev) for c in labels}
        d_b = {c: torch.from_numpy(bb[:, c - 1]).to(dev) for c in labels}
        d_n, d_tp, d_fp = metrics.bboxes_matching_batch(labels, d_s, d_b, gl, gb, gd)
        st.update(torch.stack([d_n[c] for c in labels], 1), torch.stack([d_tp[c] for c in labels], 1),
                  torch.stack([d_fp[c] for c in labels], 1), torch.stack([d_s[c] for c in labels], 1))
        o_n, o_tp, o_fp = em.bboxes_matching_batch(labels, {c: sc[:, c - 1] for c in labels}, {c: bb[:, c - 1] for c in labels}, gl, gb, gd)
        for c in labels:
            t, f, s = em.streaming_filter(o_tp[c], o_fp[c], sc[:, c - 1])
            o_acc[c][0] += int(o_n[c].sum())
            o_acc[c][1].append(t); o_acc[c][2].append(f); o_acc[c][3].append(s)
    res = metrics.evaluate(st)
    aps = []
    for c in labels:
        t, f, s = (np.concatenate(x) for x in o_acc[c][1:])
        prec, rec = em.precision_recall(o_acc[c][0], t, f, s)
        assert res['AP_VOC12/%d' % c] == em.average_precision_voc12(prec, rec)
        assert res['AP_VOC07/%d' % c] == em.average_precision_voc07(prec, rec)
        aps.append(res['AP_VOC07/%d' % c])
    assert abs(res['AP_VOC07/mAP'] - float(np.mean(aps))) < 1e-15
    assert 0.0 < res['AP_VOC07/mAP'] <= 1.0


def test_matching_rejects_too_many_gt():
    from ron_tensorflow_amd import metrics
    dev = torch.device('cuda:0')
    with pytest.raises(RuntimeError):
        metrics.bboxes_matching(torch.zeros((1, 1, 4), device=dev), torch.zeros((1, 1, 4, 4), device=dev),
                                np.zeros((1, 300), np.int64), np.zeros((1, 300, 4), np.float32), np.zeros((1, 300), np.int64))


def test_eval_driver_main(tmp_path, capsys):
    """The drop-in caller itself (ron_tensorflow_amd/eval_ron_network.py::main, the call sequence of the reference's
    eval_ron_network.py:137-366), in-process: once on synthetic weights, once restoring the same weights from a TF V2 checkpoint
    bundle written by checkpoint.write_checkpoint.  Both print the reference's summary lines and give identical detections."""
    import re
    from ron_tensorflow_amd import checkpoint, eval_ron_network
    from ron_tensorflow_amd import weights as W
    common = ['--batch_size', '1', '--max_num_batches', '3', '--dtype', 'fp32']
    first = eval_ron_network.main(common)
    out1 = capsys.readouterr().out
    prefix = str(tmp_path / 'model.ckpt-120000')
    checkpoint.write_checkpoint(prefix, W.synthetic_weights('reducedfc', 21))
    second = eval_ron_network.main(common + ['--checkpoint_path', str(tmp_path)])      # a directory: latest_checkpoint, like tf_utils.py
    out2 = capsys.readouterr().out
    for out in (out1, out2):
        assert re.search(r'AP_VOC07/mAP \d\.\d{6}  AP_VOC12/mAP \d\.\d{6}', out), out
        assert re.search(r'Time spent per BATCH: \d+\.\d{3} seconds\.', out), out
        assert len(re.findall(r'^batch \d+: \d+ detections over 20 classes$', out, re.M)) == 3
    assert first['AP_VOC07/mAP'] == second['AP_VOC07/mAP'] and first['AP_VOC12/mAP'] == second['AP_VOC12/mAP']
    assert len(first['detections']) == len(second['detections']) == 3
    kept = 0
    for a, b in zip(first['detections'], second['detections']):
        assert sorted(a) == sorted(b) == list(range(1, 21))
        for c in a:
            assert np.array_equal(a[c][0], b[c][0]) and np.array_equal(a[c][1], b[c][1]), c
            kept += int((a[c][0] > 0).sum())
    assert kept > 0
