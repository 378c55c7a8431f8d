"""The assembly K loop of the four-wave tiles is generated: `ron_tensorflow_amd/csrc/kloop4w.inc` is committed, the build does not run
the generator.  The committed file must be what `tools/gen_kloop4w.py` writes, and the schedule must keep the properties the loop's
correctness rests on (the generator asserts them while it runs; restated here so that a change to either side shows up on the CPU)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, 'ron_tensorflow_amd', 'csrc', 'kloop4w.inc')


def _macros(text):
    out = {}
    for m in re.finditer(r'#define (\w+) \\\n((?:  .*\n)+)', text):
        out[m.group(1)] = [l.strip().rstrip('\\').strip().strip('"').replace('\\n', '') for l in m.group(2).splitlines()]
    return out


def test_committed_file_is_the_generators_output(tmp_path):
    out = tmp_path / 'kloop4w.inc'
    env = dict(os.environ, KLOOP_OUT=str(out))
    env.pop('KLOOP_OPTS', None)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_kloop4w.py')], check=True, env=env, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(INC).read()


def test_loop_properties():
    mac = _macros(open(INC).read())
    for name, nb, mfmas in (('RON_KLOOP4W_BF16', 8, 128), ('RON_KLOOP4W_F16', 8, 128), ('RON_KLOOP4W_F16X3', 8, 192),
                            ('RON_KLOOP4W_N128_BF16', 4, 64), ('RON_KLOOP4W_N128_F16', 4, 64), ('RON_KLOOP4W_N128_F16X3', 4, 96)):
        lines = mac[name]
        lo = next(i for i, l in enumerate(lines) if l.startswith('.Lk4w_loop_'))
        hi = next(i for i, l in enumerate(lines) if l.startswith('s_cbranch_scc0 .Lk4w_loop_'))
        body, pro = lines[lo + 1:hi], lines[:lo]
        pieces = 8 + nb
        assert sum(1 for l in body if l.startswith('v_mfma')) == mfmas, name
        # one K tile of LDS-DMA per step, two in the prologue; three barriers per step
        assert sum(1 for l in body if l.startswith('buffer_load_dwordx4') and l.endswith('lds')) == pieces, name
        assert sum(1 for l in pro if l.startswith('buffer_load_dwordx4') and l.endswith('lds')) == 2 * pieces, name
        assert sum(1 for l in body if l == 's_barrier') == 3, name
        # every LDS-DMA has an instruction between it and the M0 write in front of it
        for seq in (pro, body):
            for a, b in zip(seq, seq[1:]):
                assert not (b.startswith('buffer_load') and 'm0' in a.split(',')[0]), (name, a, b)
        # the counted wait leaves exactly the pieces issued earlier in the same step in flight; the prologue one K tile
        vm = [l for l in body if l.startswith('s_waitcnt vmcnt')]
        assert len(vm) == 1, name
        before = sum(1 for l in body[:body.index(vm[0])] if l.startswith('buffer_load'))
        assert int(re.search(r'vmcnt\((\d+)\)', vm[0]).group(1)) == before, name
        assert 's_waitcnt vmcnt(%d)' % pieces in pro, name
        # every accumulation register is cleared before the loop and written by exactly the MFMAs of a step
        nacc = 8 * nb * 4
        assert sum(1 for l in pro if l.startswith('v_accvgpr_write_b32')) == nacc, name
        dst = set()
        for l in body:
            if l.startswith('v_mfma'):
                m = re.match(r'\S+ a\[(\d+):(\d+)\]', l)
                dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
        assert dst == set(range(nacc)), name
        # the loop touches nothing outside its declared clobbers / inputs
        clob = mac_clobbers = open(INC).read().split('#define RON_KLOOP4W_CLOBBERS')[1]
        used_v = set(int(x) for l in lines for x in re.findall(r'\bv(\d+)\b', l))
        for l in lines:
            for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', l):
                used_v.update(range(int(a), int(b) + 1))
        declared = set(int(x) for x in re.findall(r'"v(\d+)"', clob)) | set(range(100, 121))
        assert used_v <= declared, (name, sorted(used_v - declared))
        used_s = set(int(x) for l in lines for x in re.findall(r'\bs(\d+)\b', l))
        for l in lines:
            for a, b in re.findall(r'\bs\[(\d+):(\d+)\]', l):
                used_s.update(range(int(a), int(b) + 1))
        declared_s = set(int(x) for x in re.findall(r'"s(\d+)"', clob)) | set(range(36, 45))
        assert used_s <= declared_s, (name, sorted(used_s - declared_s))
        # ... M0 included: the loop rewrites it for every LDS-DMA piece
        assert any(re.search(r'\bm0\b', l) and l.split()[0] in ('s_mov_b32', 's_add_u32') for l in lines), name
        assert '"m0"' in clob, 'the loop writes M0: it must be among the clobbers'
