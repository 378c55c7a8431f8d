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
  out: a[0:255] accumulators (declared as clobbers: see main()), block (i, j) of the wave's 8 x 8 grid of 16 x 16 tiles in
       a[4 * (8 * i + j) : +3] (256 x 128 tile: a[4 * (4 * i + j) : +3])
  clobbers: s[48:64], v[84:99], v[121:255], scc, M0
"""
import os

OPTS = set(x for x in os.environ.get('KLOOP_OPTS', '').split(',') if x)

STAGE_A = 32768        # bytes of an A stage: 256 rows x 128 B
PIECE_STEP = 4096      # LDS bytes between a wave's consecutive DMA pieces (4 waves x 1 KB)
FRAG_STEP = 2048       # LDS bytes between consecutive 16-row fragments


def vr(first, n=4):
    return 'v[%d:%d]' % (first, first + n - 1)


def schedule(NB, split):
    """Where (after which MFMA of the step) every event of a K step goes.  NB = 16-column tiles per wave: 8 (256 x 256 tile) or 4
    (256 x 128).  MFMA m of a phase is (row tile i, column tile j) = divmod(m % (8 * NB), NB)."""
    if NB == 8 and not split:
        d = dict(NM=128, b1_rd=[0, 2, 4, 6, 8, 10, 12, 14], valid=(5, 7), tab_wait=(10, 5), m0b=19, w1=20,
                 bdma=[22, 25, 28, 31, 34, 52, 55, 58], a1_rd=[24, 27, 30, 33, 36, 38, 40, 42], w2=50,
                 adma=[61, 64, 85, 87, 89, 96, 100, 124], xor=83, vm=91, b0_rd=[93, 94, 95, 97, 98, 102, 103, 104],
                 a0_rd=[105, 106, 109, 112, 114, 117, 120, 123], end_wait='s_waitcnt lgkmcnt(0)')
        if 'earlyA' in OPTS:
            d['adma'] = [61, 64, 67, 70, 73, 76, 79, 82]
        if 'endwait' in OPTS:
            # only the first A fragment of the next step is needed by its first 8 MFMAs: the other seven reads may still be in flight at
            # the branch; MFMA 8 (row tile 1) is covered by a counted wait in front of it, the rest by the step's own waits
            d['end_wait'] = 's_waitcnt lgkmcnt(7)'
            d['extra_wait'] = (7, 11)
        return d
    if NB == 8 and split:
        # Three products per 16 x 16 x 32 block, in the order of the eight-wave loop (bit-identical sums): MFMA 0..63 hi * hi,
        # 64..127 lo(A) * hi(B), 128..191 hi(A) * lo(B).  hi(B) is dead after 127 and hi(A) of row tile i after 128 + 8 i + 7: the next
        # tile's hi fragments go straight into those registers (the last one, row tile 7, at the start of the next step).
        return dict(NM=192, b1_rd=[0, 2, 4, 6, 8, 10, 12, 14], valid=(5, 7), tab_wait=(10, 6), m0b=19, w1=20,
                    bdma=[22, 25, 28, 31, 34, 52, 55, 58], a1_rd=[24, 27, 30, 33, 36, 38, 40, 42], w2=50,
                    adma=[61, 64, 67, 70, 73, 76, 79, 82], xor=120, vm=123, b0_rd=[128, 130, 132, 134, 136, 138, 140, 142],
                    a0_rd=[137, 145, 153, 161, 169, 177, 185],
                    end_wait='s_waitcnt lgkmcnt(1)')       # everything but the newest read (hi(A) of row tile 6, needed at MFMA 48)
    if NB == 4 and not split:
        # 64 MFMAs per step carry the same 24 fragment reads and 12 DMA pieces
        return dict(NM=64, b1_rd=[0, 2, 4, 6], valid=(3, 5), tab_wait=(7, 3), m0b=8, w1=9,
                    bdma=[11, 13, 15, 17], a1_rd=[12, 14, 16, 18, 20, 21, 22, 23], w2=27,
                    adma=[29, 31, 33, 35, 37, 39, 41, 43], xor=44, vm=45, b0_rd=[47, 48, 49, 50],
                    a0_rd=[51, 52, 53, 54, 55, 56, 57, 58], end_wait='s_waitcnt lgkmcnt(0)')
    return dict(NM=96, b1_rd=[0, 2, 4, 6], valid=(3, 5), tab_wait=(7, 4), m0b=8, w1=9,
                bdma=[11, 13, 15, 17], a1_rd=[12, 14, 16, 18, 20, 22, 24, 26], w2=29,
                adma=[31, 33, 35, 37, 39, 41, 43, 45], xor=58, vm=60, b0_rd=[64, 65, 66, 67],
                a0_rd=[69, 73, 77, 81, 85, 89, 93], end_wait='s_waitcnt lgkmcnt(1)')


def gen(mfma, split=False, NB=8):
    sc = schedule(NB, split)
    NM = sc['NM']
    STAGE_B = NB * 2 * 16 * 128                   # the B stage: 2 waves across x NB column tiles x 16 rows x 128 B
    NBP = NB                                      # B pieces per wave and K step (A: 8)
    if NB == 8:
        FA = [[128 + 4 * i for i in range(8)], [192 + 4 * i for i in range(8)]]     # FA[half][i]: first VGPR of A fragment i
        FB = [[160 + 4 * j for j in range(8)], [224 + 4 * j for j in range(8)]]
    else:
        FA = [[128 + 4 * i for i in range(8)], [160 + 4 * i for i in range(8)]]
        FB = [[192 + 4 * j for j in range(4)], [208 + 4 * j for j in range(4)]]
    NACC = 8 * NB * 4
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
        e('v_add_u32 v%d, 0x%x, v%d' % (92 + k, STAGE_A if k < 2 else STAGE_B, 88 + k))
        e('v_xor_b32 v%d, v%d, v%d' % (92 + k, 92 + k, 88 + k))        # toggles between the two stages whatever the base
    e('s_add_u32 s62, s43, 0x%x' % STAGE_A)
    e('s_xor_b32 s62, s62, s43')
    e('s_add_u32 s63, s44, 0x%x' % STAGE_B)
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

    def tile_dma():
        for it in range(NBP):
            e('buffer_load_dwordx4 v%d, s[52:55], s57 offen lds' % (108 + it))
            if it < NBP - 1:
                e('s_add_u32 m0, m0, 0x%x' % PIECE_STEP)
                e('s_nop 0')
        e('s_mov_b32 m0, s59')
        e('s_nop 0')
        for it in range(8):
            e('buffer_load_dwordx4 v%d, s[48:51], s56 offen lds' % (100 + it))
            if it < 7:
                e('s_add_u32 m0, m0, 0x%x' % PIECE_STEP)
                e('s_nop 0')
    tile_dma()
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
    tile_dma()
    e('s_xor_b32 s59, s59, s62')
    e('s_xor_b32 s60, s60, s63')
    e('s_mov_b32 s58, 2')
    e('s_mov_b32 s61, s42')
    # the accumulators are cleared while the first two tiles are on their way (1 k cycles of v_accvgpr_write)
    for r in range(NACC):
        e('v_accvgpr_write_b32 a%d, 0' % r)
    e('s_waitcnt vmcnt(%d)' % (8 + NBP))
    e('s_barrier')
    for j in range(NB):
        e('ds_read_b128 %s, v90 offset:%d' % (vr(FB[0][j]), j * FRAG_STEP))
    for i in range(7 if split else 8):          # split precision: the last hi fragment of A is read inside the step (see below)
        e('ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][i]), i * FRAG_STEP))
    e('s_waitcnt lgkmcnt(0)')
    # ---- the K step
    ev = {m: [] for m in range(NM)}             # instructions issued right after MFMA m

    def at(m, s):
        ev[m].append(s)
    # B second fragment set (k-half 1 / lo plane) + the table entry of tile t+2
    for j in range(NB):
        at(sc['b1_rd'][j], 'ds_read_b128 %s, v91 offset:%d' % (vr(FB[1][j]), j * FRAG_STEP))
    at(1, 'ds_read_b64 v[86:87], v84')
    at(3, 'v_add_u32 v84, 8, v84')
    if split:
        at(3, 'ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][7]), 7 * FRAG_STEP))
    v0, v1 = sc['valid']
    at(v0, 's_cmp_lt_u32 s58, s42')
    at(v0, 's_cselect_b32 s50, s40, 0')
    at(v1, 's_cselect_b32 s54, s41, 0')
    at(v1, 's_add_u32 s58, s58, 1')
    if 'extra_wait' in sc:
        at(sc['extra_wait'][0], 's_waitcnt lgkmcnt(%d)' % sc['extra_wait'][1])
    at(sc['tab_wait'][0], 's_waitcnt lgkmcnt(%d)' % sc['tab_wait'][1])      # the table entry is back (LDS operations return in order)
    at(sc['tab_wait'][0], 'v_readfirstlane_b32 s56, v86')
    at(sc['tab_wait'][0], 'v_readfirstlane_b32 s57, v87')
    at(sc['m0b'], 's_mov_b32 m0, s60')
    at(sc['w1'], 's_waitcnt lgkmcnt(0)')
    at(sc['w1'] + 1, 's_barrier')
    for it, m in enumerate(sc['bdma']):
        at(m, 'buffer_load_dwordx4 v%d, s[52:55], s57 offen lds' % (108 + it))
        at(m + 1, 's_add_u32 m0, m0, 0x%x' % PIECE_STEP if it < NBP - 1 else 's_mov_b32 m0, s59')
    for i, m in enumerate(sc['a1_rd']):
        at(m, 'ds_read_b128 %s, v89 offset:%d' % (vr(FA[1][i]), i * FRAG_STEP))
    at(sc['w2'], 's_waitcnt lgkmcnt(0)')
    at(sc['w2'] + 1, 's_barrier')
    for it, m in enumerate(sc['adma']):
        at(m, 'buffer_load_dwordx4 v%d, s[48:51], s56 offen lds' % (100 + it))
        if it < 7:
            at(m + 1, 's_add_u32 m0, m0, 0x%x' % PIECE_STEP)
    for k in range(4):
        at(sc['xor'], 'v_xor_b32 v%d, v%d, v%d' % (88 + k, 92 + k, 88 + k))
    n_before = sum(1 for m in range(sc['vm']) for s_ in ev[m] if s_.startswith('buffer_load'))
    at(sc['vm'], 's_waitcnt vmcnt(%d)' % n_before)
    at(sc['vm'] + 1, 's_barrier')
    for j, m in enumerate(sc['b0_rd']):
        at(m, 'ds_read_b128 %s, v90 offset:%d' % (vr(FB[0][j]), j * FRAG_STEP))
    for i, m in enumerate(sc['a0_rd']):
        at(m, 'ds_read_b128 %s, v88 offset:%d' % (vr(FA[0][i]), i * FRAG_STEP))
    at(NM - 3, 's_xor_b32 s59, s59, s62')
    at(NM - 3, 's_xor_b32 s60, s60, s63')
    at(NM - 2, 's_sub_u32 s61, s61, 1')
    at(NM - 2, 's_cmp_eq_u32 s61, 0')
    at(NM - 2, sc['end_wait'])
    n_dma = sum(1 for m in ev for s_ in ev[m] if s_.startswith('buffer_load'))
    assert n_dma == 8 + NBP
    assert n_before in (13, 8 + NBP), n_before
    # every LDS-DMA instruction has at least one instruction between it and the M0 write in front of it
    flat = [s_ for m in range(NM) for s_ in (['MFMA'] + ev[m])]
    for k, s_ in enumerate(flat):
        assert not (s_.startswith('buffer_load') and 'm0' in flat[k - 1]), (k, flat[k - 1], s_)
    e('.Lk4w_loop_%=:')
    per = 8 * NB
    for m in range(NM):
        ph, i, j = m // per, (m % per) // NB, m % NB
        acc = 4 * (NB * i + j)
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
    for r in range(NACC):
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
        for NB, tag in ((8, ''), (4, 'N128_')):
            for name, mfma, split in (('BF16', 'v_mfma_f32_16x16x32_bf16', False), ('F16', 'v_mfma_f32_16x16x32_f16', False),
                                      ('F16X3', 'v_mfma_f32_16x16x32_f16', True)):
                f.write('#define RON_KLOOP4W_%s%s \\\n' % (tag, name))
                lines = gen(mfma, split, NB)
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
                f.write('  case %d: asm volatile("%s" : %s); break;%s\n' % (i * 4 + e_, txt, outs, '' if last else ' \\'))
        f.write('\n')
        # the 256 x 128 tile: block (i, j), j < 4, in a[4 * (4 * i + j) : +3]; two row tiles per 32-register tuple
        f.write('#define RON_ACC4W_N128_CASES \\\n')
        for i in range(8):
            for e_ in range(4):
                txt = ''.join('v_accvgpr_read_b32 %%%d, a%d\\n' % (j, 16 * i + 4 * j + e_) for j in range(4))
                outs = ', '.join('"=v"(v[%d])' % j for j in range(4))
                last = (i == 7 and e_ == 3)
                f.write('  case %d: asm volatile("%s" : %s); break;%s\n' % (i * 4 + e_, txt, outs, '' if last else ' \\'))
        f.write('\n')
        # the accumulators are clobbers, not outputs: as eight 1024-bit physical-register outputs that stay live through the epilogue's
        # few hundred basic blocks they cost hipcc 14 s of "Live Variable Analysis" per kernel (9 kernels); the epilogue reads them with
        # v_accvgpr_read in asm statements of its own, and tools/check_dma_counts.py verifies in the ISA that the compiler itself
        # touches no accumulation register between the loop and the last of those reads
        clob = ['"s%d"' % r for r in range(48, 65)] + ['"v%d"' % r for r in list(range(84, 100)) + list(range(121, 256))]
        clob_a = ['"a%d"' % r for r in range(256)]
        # m0: every LDS-DMA piece's LDS destination goes through it (s_mov_b32 m0 / s_add_u32 m0), and the compiler may hold a value of
        # its own there across an asm statement that does not name it
        f.write('#define RON_KLOOP4W_CLOBBERS "memory", "scc", "m0", %s, %s\n' % (', '.join(clob), ', '.join(clob_a)))
    print('wrote', out)


if __name__ == '__main__':
    main()
