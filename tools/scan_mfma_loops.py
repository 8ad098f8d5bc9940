#!/usr/bin/env python3
"""Static check of the ring GEMM kernels' ISA.  Their main loops are inline asm whose MFMAs the compiler cannot see, so any register move
it inserts between them that READS an accumulator register (a live-range split, a permutation between the loop's and the tail's register
assignment) reads results that may still be in the matrix pipe -- the hardware does not interlock these reads and LLVM pads them only
behind MFMAs it emitted itself.  Seen twice on gemm_ring_mx_kernel (gemm.hip): silently wrong sums in one tile.

For every gemm_ring*_kernel in the assembly: the accumulator registers are the destinations of its v_mfma instructions; between the
first and the last MFMA (in layout order) no v_mov / v_accvgpr_* / v_pk_mov / v_swap may have one of them as a source.  Exception: the
zero fill that the compiler lays out inside that span (v_mov vX, 0 followed by copies of vX).  Exits 1 on a finding.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o gemm.s atspeed_amd/csrc/gemm.hip && tools/scan_mfma_loops.py gemm.s"""
import re, sys


def regs_of(op):
    """'v[2:5]' -> {('v',2),..}; 'a17' -> {('a',17)}; anything else -> empty"""
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', op)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([va])(\d+)', op)
    return {(m.group(1), int(m.group(2)))} if m else set()


def scan(path, verbose=True):
    s = open(path).read()
    findings = 0
    for m in re.finditer(r'^(_Z\w*gemm_ring\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        name, lines = m.group(1), m.group(2).splitlines()
        mf = [i for i, l in enumerate(lines) if re.match(r'\s+v_mfma', l)]
        if not mf:
            continue
        acc = set()
        for i in mf:
            acc |= regs_of(lines[i].split()[1].rstrip(','))
        zero_src = set()                                        # registers holding the zero of the accumulator fill
        bad = []
        for i in range(mf[0], mf[-1]):
            t = lines[i].strip()
            mm = re.match(r'(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32|v_pk_mov_b32|v_swap_b32)\s+(\S+),\s*(\S+)', t)
            if not mm:
                continue
            dst, src = regs_of(mm.group(2)), regs_of(mm.group(3).rstrip(','))
            if not src:                                          # immediate
                if mm.group(3).rstrip(',') in ('0', '0x0'):
                    zero_src |= dst
                continue
            if src <= zero_src:                                  # copy of the fill value
                zero_src |= dst
                continue
            if src & acc:
                bad.append((i, t))
        short = re.sub(r'EEvPKv.*', '', name)
        short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', short)
        if verbose:
            print(f"{short:42s} {len(mf):4d} MFMAs, {len(acc):3d} accumulator registers, {len(bad):3d} moves reading them between the MFMAs {bad[:2] if bad else ''}")
        findings += len(bad)
    return findings


if __name__ == "__main__":
    sys.exit(1 if scan(sys.argv[1]) else 0)
