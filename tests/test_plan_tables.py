"""CPU: the head launch plans of csrc/graph.cpp (plan_groups: ron_order / ron_mid / ron_levels) read out of the SOURCE and checked against
the graph's true dependencies (tests/plan_util.py).  The plans are tables of op names; a table edit that lets a launch read what a
later launch - or a member of its own grouped launch - writes would pass most numerical tests most of the time.  The GPU suite checks
the same on the plans the library reports (tests/test_gpu_forward.py::test_launch_plans_respect_dependencies)."""
import os
import re

import pytest

from plan_util import check_launches, head_dependencies

SRC = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ron_tensorflow_amd', 'csrc', 'graph.cpp')).read()


def plan_table(name):
    a = SRC.index('const std::vector<Slot> %s = {' % name)
    body = SRC[a:SRC.index('\n  };', a)]
    body = re.sub(r'//[^\n]*', '', body).replace('\n', ' ')
    return [re.findall(r'"([^"]+)"', m.group(2)) for m in re.finditer(r'\{\s*(-1|MIX|G256|T64|kCfgPatch64)\s*,\s*\{([^}]*)\}\s*\}', body)]


@pytest.mark.parametrize('name', ['ron_order', 'ron_mid', 'ron_levels'])
def test_plan_table_respects_dependencies(name):
    launches = plan_table(name)
    assert len(launches) >= 7
    named = {m for l in launches for m in l}
    backbone = [[p] for p in ('conv5_1', 'conv5_2', 'conv5_3', 'pool5', 'fc6', 'fc7') if p not in named]    # ops a table does not name keep their place
    check_launches(backbone + launches, head_dependencies())


def test_the_check_sees_a_broken_plan():
    dep = head_dependencies()
    good = [[p] for p in ('conv5_1', 'conv5_2', 'conv5_3', 'pool5', 'fc6', 'fc7')] + plan_table('ron_levels')
    check_launches(good, dep)
    bad = [l for l in good]
    i = next(k for k, l in enumerate(bad) if 'block6_deconv_right' in l)
    j = next(k for k, l in enumerate(bad) if 'block6_conv_left' in l)
    bad[i], bad[j] = bad[j], bad[i]                       # the transposed conv before the left conv it adds to
    with pytest.raises(AssertionError):
        check_launches(bad, dep)
    merged = [list(l) for l in good]
    i = next(k for k, l in enumerate(merged) if 'block7_trio3' in l)
    j = next(k for k, l in enumerate(merged) if 'block7_objectness_score' in l)
    merged[i] = merged[i] + merged[j]                     # trio3 and what reads its output in ONE launch
    del merged[j]
    with pytest.raises(AssertionError, match='reads .* before it is written|member of its own launch'):
        check_launches(merged, dep)
