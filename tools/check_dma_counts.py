#!/usr/bin/env python3
"""Static check of the halo-patch kernel's counted-wait protocol (csrc/conv_patch.hip) in the ISA hipcc actually emits.

The kernel's K loop waits with `s_waitcnt vmcnt(N)` for "everything but the newest N LDS-DMA instructions"; that is only right if
every step issues exactly the group the source describes (3 B_IT + 3 instructions in the row-step form, B_IT + 1 in the one-tap
form).  Round 3 found the compiler merging identical placeholder instructions of the PROLOGUE (dead stores to it), which left
the first step's wait two short and showed as run-to-run differences; the prologue now waits for everything (vmcnt(0)) and
counts nothing, but the loop still relies on its placeholder pieces (run-time destinations: not mergeable today) being emitted one
for one.  This tool compiles conv_patch.hip to assembly (about a minute, no GPU needed) and checks,
for every instantiation: LDS-DMA instructions per loop iteration == the group size, and the loop's vmcnt immediate ==
(SB - 2) * group + pieces-behind-the-weights.  Run it after touching conv_patch.hip or changing ROCm:
    python tools/check_dma_counts.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ron_tensorflow_amd', 'csrc')


def kernels(asm):
    """{mangled name: [lines]} of every conv3x3_patch kernel in the assembly text."""
    out, name = {}, None
    for line in asm.splitlines():
        m = re.match(r'^(_ZN3ron6detail\d+conv3x3_patch\w*kernel\w+):', line)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            out[name].append(line)
            if 's_endpgm' in line:
                name = None
    return out


def check(name, lines):
    ints = [int(x) for x in re.findall(r'ELi(\d+)', name)]
    pair = 'pair_kernel' in name
    bn, wn, sb = ints[0], ints[1], ints[2]
    tps = ints[3] if pair else ints[4]
    b_it = bn // 64
    group = 3 * b_it + 3 if tps == 3 else b_it + 1
    behind = 3 if tps == 3 else 1
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    tops = [i for i, l in enumerate(lines) if re.search(r's_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)', l)
            and any('s_barrier' in x for x in lines[i + 1:i + 3])]
    problems = []
    if not tops:
        return ['no K-loop wait found']
    for top in tops:                      # a pair kernel holds the loop twice (one tile function per convolution)
        n = int(re.search(r'vmcnt\((\d+)\)', lines[top]).group(1))
        count, end = 0, None
        for i in range(top + 1, len(lines)):
            if re.search(r'buffer_load_dword\w* .* lds', lines[i]):
                count += 1
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', lines[i])
            if m and labels.get(m.group(1), 1 << 30) <= top:
                end = i
                break
        if end is None:
            problems.append('loop end not found after line %d' % top)
            continue
        if count != group:
            problems.append('%d LDS-DMA instructions per step, the protocol counts %d' % (count, group))
        if n != (sb - 2) * group + behind:
            problems.append('vmcnt(%d), expected %d' % (n, (sb - 2) * group + behind))
    return problems


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'conv_patch.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I.', '--cuda-device-only', '-S', '-o', out, 'conv_patch.hip']
        subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    ks = kernels(asm)
    bad = 0
    for name, lines in sorted(ks.items()):
        problems = check(name, lines)
        ints = re.findall(r'ELi(\d+)', name)
        tr = re.search(r'Traits(\w+?)SE', name).group(1)
        print('%-5s %-28s <%s>: %s' % (tr, 'patch_pair_kernel' if 'pair' in name else 'patch_kernel', ', '.join(ints), 'ok' if not problems else '; '.join(problems)))
        bad += len(problems)
    print('%d kernels checked' % len(ks))
    sys.exit(1 if bad or not ks else 0)


if __name__ == '__main__':
    main()
