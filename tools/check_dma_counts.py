#!/usr/bin/env python3
"""Static check of the counted-wait protocols of the conv kernels in the ISA hipcc actually emits (no GPU needed).

Both LDS-DMA kernels wait with `s_waitcnt vmcnt(N)` for "everything but the newest N LDS-DMA instructions"; that is only right if
every K step -- and the prologue -- issues exactly the group of instructions the source describes:

  conv_mfma.hip   conv_igemm_kernel / conv_igemm_group_kernel <Tr, BM, BN, WM, WN, S, SPREAD[, TI]>:
                  LPT = (BM + BN) / (WM * WN * 8) instructions per step (zero-record placeholder descriptors past the last tile
                  included), (S - 1) * LPT in the prologue, loop wait vmcnt((S - 2) * LPT);
  conv_patch.hip  conv3x3_patch_kernel / conv3x3_patch_pair_kernel <Tr, BN, WN, SB, TPS>: 3 * B_IT + 3 instructions per step in the
                  row-step form (TPS = 3), B_IT + 1 in the one-tap form, loop wait vmcnt((SB - 2) * group + pieces behind the
                  weights); the prologue is waited for as a whole (vmcnt(0)).

Round 3 found the compiler merging identical placeholder instructions of the patch kernel's prologue (dead stores to it), which
left the first step's wait two short and showed only as run-to-run differences.  This tool compiles the two files to assembly
(about a minute) and checks every instantiation; tests/test_isa_protocol.py runs it in the CPU suite, so a ROCm bump that changes
the emitted counts fails a test, not a soak run.
    python tools/check_dma_counts.py
    python tools/check_dma_counts.py --asm <device .s kept by hipcc -save-temps> conv_mfma4w.hip     (what `make` runs on the four-wave tiles:
        their epilogue reads accumulation registers the K loop's assembly only declares as clobbers, so "compiler code touches no
        accumulation register behind the loop" is a property every build has to re-establish, not just the test suite)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ron_tensorflow_amd', 'csrc')
DMA = re.compile(r'buffer_load_dword\w* .*\blds\b')
KERNEL = re.compile(r'^(_ZN3ron6detail\d+(?:conv3x3_patch\w*kernel|conv_igemm\w*kernel)\w+):')


def compile_asm(source):
    """Device-only assembly of ron_tensorflow_amd/csrc/<source> with the flags of the Makefile."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I.', '--cuda-device-only', '-S', '-o', out, source]
        subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        with open(out) as f:
            return f.read()


def kernels(asm):
    """{mangled name: [lines]} of every conv kernel in the assembly text."""
    out, name = {}, None
    for line in asm.splitlines():
        m = KERNEL.match(line)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            if line.startswith('.Lfunc_end'):       # not the first s_endpgm: a kernel may hold early exits
                name = None
            else:
                out[name].append(line)
    return out


def loops(lines):
    """[(index of the K loop's counted wait, its vmcnt immediate, LDS-DMA instructions in the loop body, LDS-DMA instructions between
    the previous loop (or the kernel's start) and this wait)]: a K loop = `s_waitcnt vmcnt(N) lgkmcnt(0)` followed by s_barrier
    within a few (scalar) instructions, up to the branch back to a label at or above it."""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    # the kernels' own counted wait is inline assembly (;;#ASMSTART in front of it): a compiler-generated `s_waitcnt vmcnt(0) lgkmcnt(0)`
    # before the s_barrier of a __syncthreads() is not a K-loop top
    tops = [i for i, l in enumerate(lines) if re.search(r's_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)', l)
            and i > 0 and '#ASMSTART' in lines[i - 1] and any('s_barrier' in x for x in lines[i + 1:i + 12])]
    out, prev_end = [], 0
    for top in tops:
        n = int(re.search(r'vmcnt\((\d+)\)', lines[top]).group(1))
        count, end = 0, None
        for i in range(top + 1, len(lines)):
            if DMA.search(lines[i]):
                count += 1
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', lines[i])
            if m and labels.get(m.group(1), 1 << 30) <= top:
                end = i
                break
        before = sum(1 for l in lines[prev_end:top] if DMA.search(l))
        out.append((top, n, count if end is not None else None, before))
        prev_end = end if end is not None else top
    return out


def template_ints(name):
    return [int(x) for x in re.findall(r'ELi(\d+)', name)]


def check_patch(name, lines):
    bn, wn, sb, tps = template_ints(name)[:4]
    b_it = bn // 64
    group = 3 * b_it + 3 if tps == 3 else b_it + 1
    behind = 3 if tps == 3 else 1
    problems = []
    found = loops(lines)
    if not found:
        return ['no K-loop wait found']
    for top, n, count, _ in found:          # a pair kernel holds the loop twice (one tile function per convolution)
        if count is None:
            problems.append('loop end not found after line %d' % top)
            continue
        if count != group:
            problems.append('%d LDS-DMA instructions per step, the protocol counts %d' % (count, group))
        if n != (sb - 2) * group + behind:
            problems.append('vmcnt(%d), expected %d' % (n, (sb - 2) * group + behind))
    return problems


def check_asm4w(name, lines):
    """The four-wave assembly K loop (kloop4w.inc, tools/gen_kloop4w.py): two whole tiles (2 x 16 pieces; 256 x 128: 2 x 12) in the
    prologue and `vmcnt(pieces)` before the first fragment reads; 16 (12) pieces per K step; the loop's one `vmcnt(N)` has exactly N of them in front of
    it in the loop body (the previous step's tile has then landed), the rest behind."""
    pieces = 8 + template_ints(name)[1] // 32          # per wave and K step: 8 of A + BN / 32 of B (256 x 256: 16, 256 x 128: 12)
    top = [i for i, l in enumerate(lines) if re.match(r'^\.Lk4w_loop_\d+:', l)]
    if len(top) != 1:
        return ['%d assembly K loops found, expected 1' % len(top)]
    top = top[0]
    start = max(i for i in range(top) if '#ASMSTART' in lines[i])
    end = [i for i in range(top, len(lines)) if re.search(r's_cbranch_scc0 \.Lk4w_loop_\d+', lines[i])]
    if len(end) != 1:
        return ['loop end not found']
    end = end[0]
    problems = []
    pro = sum(1 for l in lines[start:top] if DMA.search(l))
    if pro != 2 * pieces:
        problems.append('%d LDS-DMA instructions in the prologue, expected %d' % (pro, 2 * pieces))
    if not any(re.search(r's_waitcnt vmcnt\(%d\)$' % pieces, l.strip()) for l in lines[start:top]):
        problems.append('no vmcnt(%d) in the prologue' % pieces)
    body = lines[top:end]
    waits = [(i, int(m.group(1))) for i, l in enumerate(body) for m in [re.search(r's_waitcnt vmcnt\((\d+)\)', l)] if m]
    n_dma = sum(1 for l in body if DMA.search(l))
    if n_dma != pieces:
        problems.append('%d LDS-DMA instructions per K step, expected %d' % (n_dma, pieces))
    if len(waits) != 1:
        problems.append('%d vmcnt waits in the loop, expected 1' % len(waits))
    else:
        before = sum(1 for l in body[:waits[0][0]] if DMA.search(l))
        if before != waits[0][1]:
            problems.append('vmcnt(%d) with %d LDS-DMA instructions of the step in front of it' % (waits[0][1], before))
    if sum(1 for l in body if 's_barrier' in l) != 3:
        problems.append('%d barriers per K step, expected 3' % sum(1 for l in body if 's_barrier' in l))
    total = sum(1 for l in lines if DMA.search(l))
    if total != pro + n_dma:
        problems.append('%d LDS-DMA instructions outside the assembly loop' % (total - pro - n_dma))
    # The loop leaves its sums in a[0:255] and declares them clobbered; the epilogue reads them in asm statements of its own.  Between
    # the loop and the end of the kernel the COMPILER's code must not touch an accumulation register (it would only as spill space).
    loop_end = next(i for i in range(end, len(lines)) if '#ASMEND' in lines[i])
    inside, touched, reads = False, [], 0
    for l in lines[loop_end + 1:]:
        if '#ASMSTART' in l:
            inside = True
        elif '#ASMEND' in l:
            inside = False
        elif inside:
            reads += len(re.findall(r'v_accvgpr_read_b32 v\d+, a\d+', l))
        elif re.search(r'\bv_accvgpr|\ba\[?\d+[\]:,\s]', l.split(';')[0]):
            touched.append(l.strip())
    if touched:
        problems.append('compiler code touches accumulation registers behind the loop: %s' % touched[:3])
    if reads == 0:
        problems.append('no accumulator read-out behind the loop')
    return problems


def is_asm4w(name):
    return template_ints(name)[:5] in ([256, 256, 2, 2, 2], [256, 128, 2, 2, 2]) and ('TraitsBF16S' in name or 'TraitsF16S' in name or 'TraitsF16X3S' in name) and 'mixed' not in name


def check_igemm(name, lines):
    if is_asm4w(name):
        return check_asm4w(name, lines)
    if 'group_mixed_kernel' in name:
        # conv_igemm_group_mixed_kernel<Tr>: the 128 x 64 and the 128 x 128 tile function (4 waves, S = 2, early issue), one branch each
        forms = [(128, 64, 2, 2, 2), (128, 128, 2, 2, 2)]
    else:
        bm, bn, wm, wn, s, spread = template_ints(name)[:6]
        forms = [(bm, bn, wm, wn, s)]
    found = loops(lines)
    if len(found) != len(forms):
        return ['%d K loops found, expected %d' % (len(found), len(forms))]
    problems, accounted = [], 0
    want = sorted((bm + bn) // (wm * wn * 8) for bm, bn, wm, wn, s in forms)
    got = sorted(c for _, _, c, _ in found if c is not None)
    if len(forms) > 1 and got != want:
        return ['K loops with %s LDS-DMA instructions per step, expected %s' % (got, want)]
    for top, n, count, before in found:
        if count is None:
            return ['loop end not found after line %d' % top]
        s = forms[0][4]
        lpt = count if len(forms) > 1 else want[0]         # mixed kernel: which tile this loop belongs to is read off its own count
        if count != lpt:
            problems.append('%d LDS-DMA instructions per K step, the protocol counts LPT = %d' % (count, lpt))
        if n != (s - 2) * lpt:
            problems.append('vmcnt(%d), expected (S - 2) * LPT = %d' % (n, (s - 2) * lpt))
        if before != (s - 1) * lpt:
            problems.append('%d LDS-DMA instructions in the prologue, the first wait assumes (S - 1) * LPT = %d' % (before, (s - 1) * lpt))
        accounted += before + count
    total = sum(1 for l in lines if DMA.search(l))
    if total != accounted:
        problems.append('%d LDS-DMA instructions outside the prologues and the K loops' % (total - accounted))
    return problems


def describe(name):
    tr = re.search(r'Traits(\w+?)SE', name).group(1)
    kind = re.search(r'\d+(conv\w+?kernel)I', name).group(1)
    extra = ', TI' if 'ELb1E' in name else ''
    return '%-5s %-26s <%s%s>' % (tr, kind, ', '.join(str(i) for i in template_ints(name)), extra)


def check_file(source, asm_path=None):
    """[(description, [problems])] for every conv kernel instantiation of csrc/<source> (compiled here, or the device assembly the
    build kept: `asm_path`)."""
    if asm_path is not None:
        with open(asm_path) as f:
            ks = kernels(f.read())
    else:
        ks = kernels(compile_asm(source))
    out = []
    for name, lines in sorted(ks.items()):
        fn = check_patch if 'conv3x3_patch' in name else check_igemm
        out.append((describe(name), fn(name, lines)))
    return out


# instantiations the Makefile's build holds (dtypes x tile forms); a different count means the check no longer sees all of them
# (+ the four-wave assembly forms: the 256 x 256 tile and its grouped kernel, the 256 x 128 tile: bf16, f16, f16x3)
EXPECTED = {'conv_patch.hip': 4 * (3 + 3), 'conv_mfma.hip': 4 * (4 + 3 + 1), 'conv_mfma4w.hip': 3 * 2 + 3}


def main():
    bad = 0
    asm = None
    sources = ('conv_patch.hip', 'conv_mfma.hip', 'conv_mfma4w.hip')
    if len(sys.argv) == 4 and sys.argv[1] == '--asm':
        # the Makefile's build step: check the device assembly of the object being built (hipcc -save-temps), one source
        asm, sources = sys.argv[2], (sys.argv[3],)
    for source in sources:
        res = check_file(source, asm)
        for desc, problems in res:
            print('%s: %s' % (desc, 'ok' if not problems else '; '.join(problems)))
            bad += len(problems)
        print('%s: %d kernels checked (expected %d)' % (source, len(res), EXPECTED[source]))
        bad += len(res) != EXPECTED[source]
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
