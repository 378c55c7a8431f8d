"""CPU: the host planners of libron_hip under AddressSanitizer + UBSan (`make -C ron_tensorflow_amd/csrc asan`).

graph.cpp is 1.3 k lines of index tables (tensors, ops, grouped launch plans), conv_mfma.hip's host half picks tiles, split-K
factors and tile orders from a schedule model: an index that runs off a table there corrupts memory quietly.  The library has
a dry-run mode for exactly this (RON_PLAN_ONLY=1, csrc/common.h: every host-side decision is made, no HIP call); the `asan` target
compiles every source host-only with -fsanitize=address,undefined and runs tools/plan_sweep.cpp, which builds contexts over
variants x head plans x batch sizes, plans every batch 1..max_batch in each, and walks ron_detect / ron_clone up to the launches.
Here the quick ladder (three batch sizes per plan, ~1 minute with the build); `make asan` without arguments runs all 115 contexts."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ron_tensorflow_amd', 'csrc')


@pytest.mark.timeout(900)
def test_host_planners_are_clean_under_asan_and_ubsan():
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not available')
    p = subprocess.run(['make', '-C', CSRC, 'asan', 'ASAN_ARGS=--quick'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=850)
    out = p.stdout.decode(errors='replace')
    tail = out[-3000:]
    assert p.returncode == 0, tail
    assert 'ERROR: AddressSanitizer' not in out and 'runtime error:' not in out and 'LeakSanitizer' not in out, tail
    assert 'contexts planned, no sanitizer report' in out, tail


def test_dry_run_refuses_to_run_without_its_switch():
    exe = os.path.join(CSRC, 'build_asan', 'plan_sweep')
    if not os.path.exists(exe):
        pytest.skip('build_asan/plan_sweep not built (the test above builds it)')
    env = {k: v for k, v in os.environ.items() if k != 'RON_PLAN_ONLY'}
    p = subprocess.run([exe, '--quick'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 2 and b'RON_PLAN_ONLY' in p.stderr
