#!/usr/bin/env python3
"""Static check of the ring GEMM kernels' ISA (gemm.hip: gemm_ring_kernel, gemm_ring_mx_kernel).

Their main loops are volatile inline asm: MFMAs the compiler cannot see, and ASYNCHRONOUS ds_read / LDS-DMA statements whose results land
long after the statement.  Wherever the register allocator is free to choose it is also free to put a COPY or a SPILL between two of those
statements, and such a copy reads (a) an accumulator whose MFMA may still be in the matrix pipe -- the hardware does not interlock these
reads, and LLVM's hazard recogniser pads them only behind MFMAs it emitted itself -- or (b) a fragment register whose LDS data has not
landed.  Both were seen on gemm_ring_mx_kernel during round 2 (a live-range split inside the loop; a permutation of accumulator tuples
between the loop's and the tail's register assignment; scratch spills of fragments in the tail): silently wrong sums in one tile,
partly run-to-run different, in one instantiation and only for some K.

Rule checked, per kernel, between its first and its last MFMA in layout order:
  * no scratch_* / buffer spill instruction at all;
  * no v_mov / v_accvgpr_* / v_pk_mov / v_swap whose source is a vector register -- except the accumulators' zero fill, which the
    compiler lays out inside that span as runs of >= 8 consecutive moves from ONE source register (a splat), and moves of literals.
Exits 1 on a finding.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o gemm.s atspeed_amd/csrc/gemm.hip && tools/scan_mfma_loops.py gemm.s"""
import re, sys

MOVE = re.compile(r'\s+(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32|v_pk_mov_b32|v_swap_b32)\s+(\S+),\s*(\S+)')


def scan(path, verbose=True):
    s = open(path).read()
    findings, kernels = 0, 0
    for m in re.finditer(r'^(_Z\w*gemm_ring\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        name, lines = m.group(1), m.group(2).splitlines()
        mf = [i for i, l in enumerate(lines) if re.match(r'\s+v_mfma', l)]
        if not mf:
            continue
        kernels += 1
        moves = []                                              # (line, text, source operand) of register-to-register moves in the span
        bad = []
        for i in range(mf[0], mf[-1]):
            t = lines[i]
            if re.match(r'\s+(scratch_|buffer_(load|store)_dword.*offen)', t):
                bad.append((i, t.strip()[:60]))
                continue
            mm = MOVE.match(t)
            if mm and re.match(r'[va](\d+|\[)', mm.group(3).rstrip(',')):
                moves.append((i, t.strip(), mm.group(3).rstrip(',')))
        # runs of consecutive moves from one source register = the zero fill
        k = 0
        while k < len(moves):
            e = k
            while e + 1 < len(moves) and moves[e + 1][2] == moves[k][2] and moves[e + 1][0] - moves[e][0] <= 2:
                e += 1
            if e - k + 1 < 8:
                bad.extend((i, t[:60]) for i, t, _ in moves[k:e + 1])
            k = e + 1
        short = re.sub(r'EEvPKv.*', '', name)
        short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', short)
        if verbose:
            print(f"{short:40s} {len(mf):4d} MFMAs, {len(moves):4d} register moves between them (zero fill included), {len(bad):3d} findings {bad[:2] if bad else ''}")
        findings += len(bad)
    return findings, kernels


def scan_dma(path, verbose=True, want=r'.'):
    """Second rule, for EVERY kernel that issues LDS-DMA (`global_load_lds_*`) from inline asm and waits for it with hand-counted
    `s_waitcnt vmcnt(N)` -- the ring GEMMs, gemm_wdma_kernel, the attention rings of attn.hip.  On gfx9 every vector-memory operation counts
    in vmcnt.  Loads retire in order among themselves, so a load the compiler issues itself only makes a counted wait stricter; STORES (and
    scratch spills, which are stores + loads) retire out of order with loads, so one of them inside the counted window makes the count
    wrong: tiles read before they land, silently wrong sums.  Checked per kernel, in layout order:
      * between the first LDS-DMA and the last MFMA (prologue + main loop; the epilogue's stores come after): no scratch_* instruction, no
        buffer_ / flat_ / global_ STORE or atomic;
      * the kernel descriptor's .amdhsa_private_segment_fixed_size is 0, or every spill sits in front of the first LDS-DMA and every reload
        behind the last MFMA (nothing of it in flight while a count is relied on);
      * `strict` kernels (the GEMMs, whose loops contain nothing but asm): no compiler VMEM load in that window either.
    -> (findings, kernels)"""
    s = open(path).read()
    priv = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(?:[^\n]*\n)*?\s+\.amdhsa_private_segment_fixed_size (\d+)', s))
    findings = kernels = 0
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        name, lines = m.group(1), m.group(2).splitlines()
        dma = [i for i, l in enumerate(lines) if re.match(r'\s+global_load_lds', l)]
        mf = [i for i, l in enumerate(lines) if re.match(r'\s+v_mfma', l)]
        if not dma or not mf or not re.search(want, name):
            continue
        kernels += 1
        strict = "gemm_" in name
        bad = []
        for i in range(dma[0], mf[-1]):
            t = lines[i]
            if re.match(r'\s+scratch_', t) or re.match(r'\s+(buffer|flat|global)_(store|atomic)', t):
                bad.append((i, t.strip()[:60]))
            elif strict and re.match(r'\s+(buffer_|flat_|global_(?!load_lds))', t):
                bad.append((i, t.strip()[:60]))
        note = ""
        if priv.get(name, 0) != 0:
            # scratch is tolerated only as "spilled before the ring starts, reloaded after its last MFMA" (the stream-K RoPE instantiation keeps a
            # few epilogue values that way at 256 VGPRs): every scratch store in front of the first LDS-DMA, every reload behind the last MFMA
            st = [i for i, l in enumerate(lines) if re.match(r'\s+scratch_store', l)]
            ld = [i for i, l in enumerate(lines) if re.match(r'\s+scratch_load', l)]
            if (st and max(st) > dma[0]) or (ld and min(ld) < mf[-1]) or not (st or ld):
                bad.append((-1, f"private_segment_fixed_size {priv[name]} with scratch traffic inside the ring"))
            else:
                note = f" (scratch {priv[name]} B: {len(st)} spills before the ring, {len(ld)} reloads after it)"
        short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', re.sub(r'EEvPK.*', '', name))
        if verbose:
            print(f"{short:44s} {len(dma):4d} LDS-DMA, {len(mf):4d} MFMAs, {len(bad):3d} VMEM / scratch findings in the counted window {bad[:2] if bad else ''}{note}")
        findings += len(bad)
    return findings, kernels


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--dma-only":           # attn.hip: no asm MFMAs, only the counted DMA windows
        f2, k2 = scan_dma(sys.argv[1])
        print(f"{k2} LDS-DMA kernels, {f2} findings")
        sys.exit(1 if f2 or not k2 else 0)
    f, k = scan(sys.argv[1])
    f2, k2 = scan_dma(sys.argv[1])
    print(f"{k} ring kernels, {f} findings; {k2} LDS-DMA kernels, {f2} findings")
    sys.exit(1 if f or f2 or not k or not k2 else 0)
