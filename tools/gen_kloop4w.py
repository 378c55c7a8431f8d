#!/usr/bin/env python3
"""Generates ron_tensorflow_amd/csrc/kloop4w.inc: the K loop of the 256 x 256 tile on FOUR waves (one per SIMD, 128 x 128 per wave,
256 accumulator + 256 vector registers) as one inline-asm block per dtype.

  python3 tools/gen_kloop4w.py            (rewrites the .inc; the file is committed, the build does not run this script)

Why assembly: the loop's structure is the point.  A K step (64 bf16 of K, 128 MFMA 16x16x32 per wave) keeps TWO tiles of LDS-DMA in
flight with only two LDS stages, because a stage is released as soon as its fragments are in registers:
    MFMA   0.. 63  k-half 0 (fragments read at the end of the previous step)
           0.. 14  read k-half 1 of B            -> wait, barrier 1 -> B stage free  -> DMA of tile t+2's B pieces starts
          24.. 42  read k-half 1 of A            -> wait, barrier 2 -> A stage free  -> DMA of tile t+2's A pieces starts
    MFMA  64..127  k-half 1
          91       vmcnt: tile t+1 landed (13 younger pieces stay in flight), barrier 3
          93..123  read k-half 0 of tile t+1
A piece is issued 0.75-1.55 K steps before it is needed (the 8-wave loop of conv_mfma.hip: 0.1-1.0).  The order of the events is the
one hipBLASLt's hand-written 256x256x64 kernel uses on this chip (profiles/r05/vendor_loop_vs_conv_igemm.md); registers, LDS
layout (128-byte rows, XOR-swizzled 16-byte slots), the row-gather A operand and the per-step offset table are this kernel's own.

Interface (physical registers; conv_mfma.hip, conv_igemm_tile, binds them):
  in : s[36:37] activations base, s[38:39] packed weights base, s40 / s41 their sizes in bytes (buffer range), s42 K steps (may be 0),
       s43 / s44 LDS byte address this wave's first DMA piece of stage 0 goes to (A / B),
       v[100:107] / v[108:115] per-lane source byte offsets of the wave's 8 A / 8 B pieces, v116 / v117 LDS byte address the lane's
       first A fragment is read from in stage 0 for k-half 0 / 1, v118 / v119 the same for B, v120 LDS byte address of the step
       table: entry q = {soffset of the A pieces, soffset of the B pieces} of K step q, steps + 2 entries
  out: a[0:255] accumulators, block (i, j) of the wave's 8 x 8 grid of 16 x 16 tiles in a[4 * (8 * i + j) : +3]
  clobbers: s[48:64], v[84:99], v[121:255], scc, M0
"""
import os

OPTS = set(x for x in os.environ.get('KLOOP_OPTS', '').split(',') if x)

STAGE = 32768          # bytes of one operand's stage: 256 rows x 128 B
PIECE_STEP = 4096      # LDS bytes between a wave's consecutive DMA pieces (4 waves x 1 KB)
FRAG_STEP = 2048       # LDS bytes between consecutive 16-row fragments

FA = [[128 + 4 * i for i in range(8)], [192 + 4 * i for i in range(8)]]     # FA[half][i]: first VGPR of A fragment i
FB = [[160 + 4 * j for j in range(8)], [224 + 4 * j for j in range(8)]]


def vr(first, n=4):
    return 'v[%d:%d]' % (first, first + n - 1)


def gen(mfma, split=False):
    L = []
    e = L.append
    # ---- set-up
    e('s_mov_b32 s48, s36')
    e('s_and_b32 s49, s37, 0xffff')
    e('s_mov_b32 s50, s40')
    e('s_mov_b32 s51, 0x00020000')
    e('s_mov_b32 s52, s38')
    e('s_and_b32 s53, s39, 0xffff')
    e('s_mov_b32 s54, s41')
    e('s_mov_b32 s55, 0x00020000')
    e('s_cmp_eq_u32 s42, 0')
    e('s_cbranch_scc1 .Lk4w_zero_%=')
    for k in range(4):
        e('v_mov_b32 v%d, v%d' % (88 + k, 116 + k))                    # v88 rdA0, v89 rdA1, v90 rdB0, v91 rdB1
        e('v_add_u32 v%d, 0x%x, v%d' % (92 + k, STAGE, 88 + k))
        e('v_xor_b32 v%d, v%d, v%d' % (92 + k, 92 + k, 88 + k))        # toggles between the two stages whatever the base
    e('s_add_u32 s62, s43, 0x%x' % STAGE)
    e('s_xor_b32 s62, s62, s43')
    e('s_add_u32 s63, s44, 0x%x' % STAGE)
    e('s_xor_b32 s63, s63, s44')
    e('s_mov_b32 s59, s43')
    e('s_mov_b32 s60, s44')
    e('v_mov_b32 v84, v120')
    e('ds_read_b64 v[86:87], v84')
    e('ds_read_b64 v[96:97], v84 offset:8')
    e('v_add_u32 v84, 16, v84')
    e('s_waitcnt lgkmcnt(0)')
    e('v_readfirstlane_b32 s56, v86')
    e('v_readfirstlane_b32 s57, v87')
    e('s_mov_b32 m0, s60')
    e('s_nop 4')

    def tile_dma(first):
        for it in range(8):
            e('buffer_load_dwordx4 v%d, s[52:55], s57 offen lds' % (108 + it))
            if it < 7:
                e('s_add_u32 m0, m0, 0x%x' % PIECE_STEP)
                e('s_nop 0')
        e('s_mov_b32 m0, s59')
        e('s_nop 0')
        for it in range(8):
            e('buffer_load_dwordx4 v%d, s[48:51], s56 offen lds' % (100 + it))
            if it < 7:
                e('s_add_u32 m0, m0, 0x%x' % PIECE_STEP)
                e('s_nop 0')
    tile_dma(True)
    # tile 1 -> stage 1 (a zero-record descriptor when there is no tile 1: nothing moves, the vmcnt bookkeeping stays uniform)
    e('v_readfirstlane_b32 s56, v96')
    e('v_readfirstlane_b32 s57, v97')
    e('s_cmp_gt_u32 s42, 1')
    e('s_cselect_b32 s50, s40, 0')
    e('s_cselect_b32 s54, s41, 0')
    e('s_xor_b32 s59, s59, s62')
    e('s_xor_b32 s60, s60, s63')
    e('s_mov_b32 m0, s60')
    e('s_nop 4')
    tile_dma(False)
    e('s_xor_b32 s59, s59, s62')
    e('s_xor_b32 s60, s60, s63')
    e('s_mov_b32 s58, 2')
    e('s_mov_b32 s61, s42')
    # the accumulators are cleared while the first two tiles are on their way (1 k cycles of v_accvgpr_write)
    for r in range(256):
        e('v_accvgpr_write_b32 a%d, 0' % r)
    e('s_waitcnt vmcnt(16)')
    e('s_barrier')
    for j in range(8):
        e('ds_read_b128 %s, v90 offset:%d' % (vr(FB[0][j]), j * FRAG_STEP))
    for i in range(7 if split else 8):          # split precision: the last hi fragment of A is read inside the step (see below)
        e('ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][i]), i * FRAG_STEP))
    e('s_waitcnt lgkmcnt(0)')
    # ---- the K step
    NM = 192 if split else 128
    ev = {m: [] for m in range(NM)}             # instructions issued right after MFMA m

    def at(m, s):
        ev[m].append(s)
    # B second fragment set (k-half 1 / lo plane) + the table entry of tile t+2
    for j in range(8):
        at(2 * j, 'ds_read_b128 %s, v91 offset:%d' % (vr(FB[1][j]), j * FRAG_STEP))
    at(1, 'ds_read_b64 v[86:87], v84')
    at(3, 'v_add_u32 v84, 8, v84')
    if split:
        at(3, 'ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][7]), 7 * FRAG_STEP))
    at(5, 's_cmp_lt_u32 s58, s42')
    at(5, 's_cselect_b32 s50, s40, 0')
    at(7, 's_cselect_b32 s54, s41, 0')
    at(7, 's_add_u32 s58, s58, 1')
    if 'endwait' in OPTS and not split:
        at(7, 's_waitcnt lgkmcnt(11)')           # 7 reads of the previous step + 5 of this one issued: the oldest (row tile 1's A) is back
        at(10, 's_waitcnt lgkmcnt(5)')
    else:
        at(10, 's_waitcnt lgkmcnt(%d)' % (6 if split else 5))      # the table entry is back (LDS operations return in order)
    at(10, 'v_readfirstlane_b32 s56, v86')
    at(10, 'v_readfirstlane_b32 s57, v87')
    at(19, 's_mov_b32 m0, s60')
    at(20, 's_waitcnt lgkmcnt(0)')
    at(21, 's_barrier')
    bpos = [22, 25, 28, 31, 34, 52, 55, 58]
    for it, m in enumerate(bpos):
        at(m, 'buffer_load_dwordx4 v%d, s[52:55], s57 offen lds' % (108 + it))
        at(m + 1, 's_add_u32 m0, m0, 0x%x' % PIECE_STEP if it < 7 else 's_mov_b32 m0, s59')
    apos_rd = [24, 27, 30, 33, 36, 38, 40, 42]
    for i, m in enumerate(apos_rd):
        at(m, 'ds_read_b128 %s, v89 offset:%d' % (vr(FA[1][i]), i * FRAG_STEP))
    at(50, 's_waitcnt lgkmcnt(0)')
    at(51, 's_barrier')
    if not split:
        apos = [61, 64, 85, 87, 89, 96, 100, 124]
        if 'earlyA' in OPTS:
            apos = [61, 64, 67, 70, 73, 76, 79, 82]
        t_xor, t_wait = 83, 91
        bpos_rd = [93, 94, 95, 97, 98, 102, 103, 104]
        apos_rd0 = [105, 106, 109, 112, 114, 117, 120, 123]
        end_wait = 's_waitcnt lgkmcnt(0)'
        if 'endwait' in OPTS:
            # only the first A fragment of the next step is needed by its first 8 MFMAs: the other seven reads may still be in flight at
            # the branch; MFMA 8 (row tile 1) is covered by a counted wait in front of it, the rest by the step's own waits
            end_wait = 's_waitcnt lgkmcnt(7)'
    else:
        # Three products per 16 x 16 x 32 block, in the order of the eight-wave loop (bit-identical sums): MFMA 0..63 hi * hi,
        # 64..127 lo(A) * hi(B), 128..191 hi(A) * lo(B).  hi(B) is dead after 127 and hi(A) of row tile i after 128 + 8 i + 7: the next
        # tile's hi fragments go straight into those registers (the last one, row tile 7, at the start of the next step).
        apos = [61, 64, 67, 70, 73, 76, 79, 82]
        t_xor, t_wait = 120, 123
        bpos_rd = [128, 130, 132, 134, 136, 138, 140, 142]
        apos_rd0 = [137, 145, 153, 161, 169, 177, 185]
        end_wait = 's_waitcnt lgkmcnt(1)'           # everything but the newest read (hi(A) of row tile 6, needed at MFMA 48)
    for it, m in enumerate(apos):
        at(m, 'buffer_load_dwordx4 v%d, s[48:51], s56 offen lds' % (100 + it))
        if it < 7:
            at(m + 1, 's_add_u32 m0, m0, 0x%x' % PIECE_STEP)
    for k in range(4):
        at(t_xor, 'v_xor_b32 v%d, v%d, v%d' % (88 + k, 92 + k, 88 + k))
    n_before = sum(1 for m in range(t_wait) for s_ in ev[m] if s_.startswith('buffer_load'))
    at(t_wait, 's_waitcnt vmcnt(%d)' % n_before)
    at(t_wait + 1, 's_barrier')
    for j, m in enumerate(bpos_rd):
        at(m, 'ds_read_b128 %s, v90 offset:%d' % (vr(FB[0][j]), j * FRAG_STEP))
    for i, m in enumerate(apos_rd0):
        at(m, 'ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][i]), i * FRAG_STEP))
    at(NM - 3, 's_xor_b32 s59, s59, s62')
    at(NM - 3, 's_xor_b32 s60, s60, s63')
    at(NM - 2, 's_sub_u32 s61, s61, 1')
    at(NM - 2, 's_cmp_eq_u32 s61, 0')
    at(NM - 2, end_wait)
    n_dma = sum(1 for m in ev for s_ in ev[m] if s_.startswith('buffer_load'))
    assert n_dma == 16
    assert n_before in (13, 16), n_before
    e('.Lk4w_loop_%=:')
    for m in range(NM):
        ph, i, j = m // 64, (m % 64) // 8, m % 8
        acc = 4 * (8 * i + j)
        if split:
            a, b = (FA[0][i], FB[0][j]) if ph == 0 else ((FA[1][i], FB[0][j]) if ph == 1 else (FA[0][i], FB[1][j]))
        else:
            a, b = FA[ph][i], FB[ph][j]
        e('%s a[%d:%d], %s, %s, a[%d:%d]' % (mfma, acc, acc + 3, vr(a), vr(b), acc, acc + 3))
        for s_ in ev[m]:
            e(s_)
    e('s_cbranch_scc0 .Lk4w_loop_%=')
    e('s_waitcnt vmcnt(0)')
    e('s_branch .Lk4w_end_%=')
    e('.Lk4w_zero_%=:')                          # no K steps (an empty split-K slice): zeros
    for r in range(256):
        e('v_accvgpr_write_b32 a%d, 0' % r)
    e('.Lk4w_end_%=:')
    e('s_nop 15')
    e('s_nop 15')
    return L


def main():
    out = os.environ.get('KLOOP_OUT') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ron_tensorflow_amd', 'csrc', 'kloop4w.inc')
    with open(out, 'w') as f:
        f.write('// GENERATED by tools/gen_kloop4w.py - do not edit.  The K loop of the 256 x 256 tile on four waves as inline assembly;\n')
        f.write('// interface, schedule and rationale: the generator\'s docstring.\n')
        for name, mfma, split in (('BF16', 'v_mfma_f32_16x16x32_bf16', False), ('F16', 'v_mfma_f32_16x16x32_f16', False),
                                  ('F16X3', 'v_mfma_f32_16x16x32_f16', True)):
            f.write('#define RON_KLOOP4W_%s \\\n' % name)
            lines = gen(mfma, split)
            for k, s in enumerate(lines):
                f.write('  "%s\\n"%s\n' % (s, ' \\' if k + 1 < len(lines) else ''))
            f.write('\n')
        # accumulator read-out, one output row of the lane at a time (AccAgpr4w::row in conv_mfma.hip): case i * 4 + e
        f.write('#define RON_ACC4W_CASES \\\n')
        for i in range(8):
            for e_ in range(4):
                txt = ''.join('v_accvgpr_read_b32 %%%d, a%d\\n' % (j, 32 * i + 4 * j + e_) for j in range(8))
                outs = ', '.join('"=v"(v[%d])' % j for j in range(8))
                last = (i == 7 and e_ == 3)
                f.write('  case %d: asm volatile("%s" : %s : "{a[%d:%d]}"(c%d)); break;%s\n' %
                        (i * 4 + e_, txt, outs, 32 * i, 32 * i + 31, i, '' if last else ' \\'))
        f.write('\n')
        clob = ['"s%d"' % r for r in range(48, 65)] + ['"v%d"' % r for r in list(range(84, 100)) + list(range(121, 256))]
        f.write('#define RON_KLOOP4W_CLOBBERS "memory", "scc", %s\n' % ', '.join(clob))
    print('wrote', out)


if __name__ == '__main__':
    main()
