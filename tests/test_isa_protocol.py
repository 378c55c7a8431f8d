"""CPU test: the counted `s_waitcnt vmcnt(N)` protocols of the LDS-DMA conv kernels, checked in the ISA hipcc emits.

The dominant kernel (csrc/conv_mfma.hip, conv_igemm_kernel and its grouped form) and the halo-patch kernel (csrc/conv_patch.hip) wait
for "all but the newest N LDS-DMA instructions"; N is only right when the compiler emits the instructions one for one, placeholders
with zero-record descriptors included.  Round 3 lost two placeholders of the patch kernel's prologue to dead-store merging (a real
race, visible only as run-to-run differences).  tools/check_dma_counts.py cross-compiles both files for gfx950 (no GPU) and checks,
per instantiation, the instructions per K step, the loop's vmcnt immediate and the prologue's count; here it runs as a test, so a
compiler / ROCm change that alters the counts fails the suite."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_dma_counts as cdc  # noqa: E402


@pytest.fixture(scope='module')
def checked():
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not available')
    with ThreadPoolExecutor(3) as ex:                       # three hipcc -S runs side by side (~35 s each)
        return dict(zip(cdc.EXPECTED, ex.map(cdc.check_file, cdc.EXPECTED)))


@pytest.mark.parametrize('source', sorted(cdc.EXPECTED))
def test_every_instantiation_issues_the_group_its_wait_counts(checked, source):
    res = checked[source]
    assert len(res) == cdc.EXPECTED[source], '%s: %d kernels found, the build holds %d' % (source, len(res), cdc.EXPECTED[source])
    bad = ['%s: %s' % (d, '; '.join(p)) for d, p in res if p]
    assert not bad, '\n'.join(bad)


def test_the_checker_sees_a_dropped_placeholder():
    """The check itself: a K loop with one LDS-DMA fewer than its wait counts, and a prologue short of one, are reported."""
    name = '_ZN3ron6detail17conv_igemm_kernelINS0_11TraitsBF16SELi128ELi128ELi2ELi2ELi3ELi1ELb0EEEvNS0_8ConvArgsE'   # S = 3: LPT = 8
    dma = '\tbuffer_load_dwordx4 v1, s[0:3], s4 offen lds'
    good = [dma] * 16 + ['.LBB0_1:', '\t;;#ASMSTART', '\ts_waitcnt vmcnt(8) lgkmcnt(0)', '\t;;#ASMEND', '\ts_barrier'] + [dma] * 8 + ['\ts_cbranch_scc1 .LBB0_1', '\ts_endpgm']
    assert cdc.check_igemm(name, good) == []
    short_loop = [dma] * 16 + good[16:21] + [dma] * 7 + good[-2:]          # prologue intact (16), one of the loop's eight gone
    assert any('per K step' in p for p in cdc.check_igemm(name, short_loop))
    short_prologue = [dma] * 15 + good[16:]
    assert any('prologue' in p for p in cdc.check_igemm(name, short_prologue))
    wrong_wait = [l.replace('vmcnt(8)', 'vmcnt(9)') for l in good]
    assert any('vmcnt(9)' in p for p in cdc.check_igemm(name, wrong_wait))
