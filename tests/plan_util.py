"""Dependencies of the RON graph from conv5_1 on and the check a launch plan must pass - shared by the CPU test that reads the plan tables
out of csrc/graph.cpp and the GPU test that reads the plans back from the library."""


def head_dependencies():
    """reads / writes of every op from conv5_1 on, by name (csrc/graph.cpp; nets/ron_vgg_320.py:418-432, :454-506).  The reference map
    of a scale has two versions: '<scale>.left' (the left conv's half) and the finished map."""
    dep = {'conv5_1': (['pool4'], ['conv5_1']), 'conv5_2': (['conv5_1'], ['conv5_2']), 'conv5_3': (['conv5_2'], ['conv5_3']),
           'pool5': (['conv5_3'], ['pool5']), 'fc6': (['pool5'], ['fc6']), 'fc7': (['fc6'], ['fc7'])}
    scales, left_src = ['block7', 'block6', 'block5', 'block4'], ['fc7', 'fc6', 'conv5_3', 'conv4_3']
    for i, (sc, src) in enumerate(zip(scales, left_src)):
        if i == 0:
            dep[sc + '_conv_left'] = ([src], [sc + '.ref'])
        else:
            dep[sc + '_conv_left'] = ([src], [sc + '.left'])
            dep[sc + '_deconv_right'] = ([scales[i - 1] + '.ref', sc + '.left'], [sc + '.ref'])
        dep[sc + '_trio3'] = ([sc + '.ref'], [sc + '.hcat'])
        dep[sc + '_objectness_score'] = ([sc + '.hcat'], [sc + '.obj'])
        dep[sc + '_loc_pred'] = ([sc + '.hcat'], [sc + '.loc'])
        dep[sc + '_inception2'] = ([sc + '.hcat'], [sc + '.inc2'])
        dep[sc + '_cls_pred'] = ([sc + '.inc2'], [sc + '.cls'])
    return dep


def check_launches(launches, dep):
    """launches: [[op names of one launch]] in order.  Every op exactly once; a launch only reads what EARLIER launches wrote; members of
    a grouped launch (they run concurrently) neither read nor write what another member writes."""
    seen = [m for l in launches for m in l]
    assert sorted(seen) == sorted(dep), sorted(set(dep) ^ set(seen))
    written = {'pool4', 'conv4_3'}
    for l in launches:
        writes = [x for m in l for x in dep[m][1]]
        assert len(set(writes)) == len(writes), l
        for m in l:
            for r in dep[m][0]:
                assert r in written, '%s reads %s before it is written (launch %s)' % (m, r, l)
            others = [x for o in l if o != m for x in dep[o][1]]
            assert not (set(dep[m][0]) & set(others)), '%s reads what a member of its own launch writes: %s' % (m, l)
        written |= set(writes)
