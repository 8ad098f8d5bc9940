"""The ring GEMM kernels (gemm.hip) issue their MFMAs, fragment reads and LDS-DMA as volatile inline asm.  A register copy or a spill that
the compiler puts between those statements reads an accumulator still in the matrix pipe or a fragment whose LDS data has not landed:
silently wrong sums (seen in round 2 on gemm_ring_mx_kernel; tools/scan_mfma_loops.py has the story).  This test compiles gemm.hip for
gfx950 (device pass only, no GPU needed) and checks the ISA of every ring kernel for such instructions."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_register_moves_or_spills_between_the_ring_kernels_asm_mfmas(tmp_path):
    import scan_mfma_loops
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # `make` runs this scan on the ISA of the very compile that produced gemm.o (atspeed_amd/csrc/Makefile) and fails the build on a finding;
    # when that by-product is there and current, check it rather than compiling again
    built = os.path.join(ROOT, "atspeed_amd", "csrc", ".isa", "gemm-hip-amdgcn-amd-amdhsa-gfx950.s")
    src = os.path.join(ROOT, "atspeed_amd", "csrc", "gemm.hip")
    if os.path.exists(built) and os.path.getmtime(built) >= os.path.getmtime(src):
        findings, kernels = scan_mfma_loops.scan(built, verbose=False)
        assert kernels >= 20 and findings == 0, (kernels, findings)
        f2, k2 = scan_mfma_loops.scan_dma(built, verbose=False)
        assert k2 >= 40 and f2 == 0, (k2, f2)
        return
    out = tmp_path / "gemm.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "atspeed_amd", "csrc", "gemm.hip")], check=True, cwd=os.path.join(ROOT, "atspeed_amd", "csrc"),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    findings, kernels = scan_mfma_loops.scan(str(out), verbose=False)
    assert kernels >= 20, f"only {kernels} ring kernels found in the assembly: the scan is looking at the wrong thing"
    assert findings == 0, "register moves / spills between asm MFMAs: run tools/scan_mfma_loops.py on the assembly for the list"
    f2, k2 = scan_mfma_loops.scan_dma(str(out), verbose=False)
    assert k2 >= 40 and f2 == 0, f"{f2} compiler VMEM / scratch instructions inside hand-counted LDS-DMA windows of {k2} kernels"


def test_no_compiler_stores_inside_the_attention_kernels_counted_dma_windows(tmp_path):
    """attn.hip's K / V rings (tree_attn32_kernel, tree_attn_mfma_kernel<.., 4>) wait for LDS-DMA with hand-counted vmcnt: same rule."""
    import scan_mfma_loops
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    built = os.path.join(ROOT, "atspeed_amd", "csrc", ".isa", "attn-hip-amdgcn-amd-amdhsa-gfx950.s")
    src = os.path.join(ROOT, "atspeed_amd", "csrc", "attn.hip")
    if not (os.path.exists(built) and os.path.getmtime(built) >= os.path.getmtime(src)):
        built = str(tmp_path / "attn.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", built, src], check=True,
                       cwd=os.path.join(ROOT, "atspeed_amd", "csrc"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    findings, kernels = scan_mfma_loops.scan_dma(built, verbose=False)
    assert kernels >= 8 and findings == 0, (kernels, findings)
