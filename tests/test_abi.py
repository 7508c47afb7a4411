"""The C-ABI shared library loads and exports every symbol include/im360_kernels.h declares (no compute)."""
import ctypes
import os
import re

from imagine360_amd import kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "im360_kernels.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(im360_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"))
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_library_exports_nothing_the_header_does_not_declare():
    """VERDICT r4 (b) nit: im360_set_error was exported but undeclared -- internal helpers are hidden now."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so")],
                         capture_output=True, text=True, check=True).stdout
    # EVERY defined dynamic symbol (seven mangled C++ internals used to leak beside the C ABI; the library is
    # now linked with csrc/exports.map)
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split()})
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_binding_rejects_a_library_of_another_abi_version(monkeypatch):
    """the ctypes binding refuses a library whose im360_abi_version() is not the one it was written for."""
    import pytest
    monkeypatch.setattr(kernels, "_lib", None)
    monkeypatch.setattr(kernels, "ABI_VERSION", 1)
    with pytest.raises(RuntimeError, match="C-ABI version"):
        kernels.lib()
    monkeypatch.setattr(kernels, "ABI_VERSION", 5)
    assert kernels.lib().im360_abi_version() == 5


def test_python_binding_covers_header():
    assert set(declared_symbols()) == set(kernels.exported_symbols())
    assert kernels.lib().im360_abi_version() == kernels.ABI_VERSION == 5
    assert kernels.lib().im360_last_error() is not None


# kernels that exist only as measured-and-rejected A/B variants or cold fallbacks may spill; everything a default launch can
# reach may not (a 688-byte private array in the 256 x 64 conv tile made cfg1's convolutions 8x slower for a round, unnoticed)
_SPILL_ALLOWED = ("temporal_attn_lds_kernel",                       # scalar fallback (knob tattn_scalar / > 64 frames)
                  "Li2ELi2ELi3ELi5E",                               # 192 x 320 four-wave tile (knob conv_big 4)
                  "Li64ELi4ELi1ELi2ELi2ELi0E",                      # 256 x 64 tile with 64-channel K steps (knob conv_small 1)
                  "gemm_g4_kernelIDF16bLi2E", "gemm_g4_kernelIDF16_Li2E")      # four-wave tile with the plain epilogue (knob conv_ring 12, A/B only): hipcc moves accumulators through scratch in the epilogue


# minimum waves per SIMD the registers of a default-path kernel must allow (gfx950: 512 registers per lane and SIMD, allocated
# in granules of 8; VGPRs + AGPRs).  A kernel that silently grows past its step loses a third or half of its latency hiding --
# hipcc only warns ("failed to meet occupancy target").  First match wins.
_MIN_WAVES = (("attn_pipe_kernel", 2),
              ("xattn_resident_kernel", 2),
              (r"attn_fwd_kernelIDF16[b_]Li\d+ELi1E", 1),       # one-wave workgroups (<= 32 query rows): 16 staging chunks per lane
              (r"attn_fwd_kernel.*Lb1EEEv", 3),                  # W3: one query block per wave at three waves per SIMD
              (r"attn_fwd_kernelIDF16[b_]Li64ELi[24]ELi1ELb0ELb0ELb0ELb0ELb0ELb1E", 3),      # single-tile variant
              ("attn_fwd_kernel", 2),
              (r"conv_igemm_kernel.*Li2ELi2ELi3ELi5E", 1),        # (ablation builds only) four-wave 192 x 320 tile: the whole register file
              ("conv_igemm_kernel", 2), ("conv_ring_kernel", 2), ("conv_halo_kernel", 2),
              ("gemm_g4b_kernel", 2),                             # round 6: 256 x 128 tile, two workgroups per CU (A/B variant)
              ("gemm_g4_kernel", 1),                              # round 6: 256 x 256 tile on one wave per SIMD: the whole register file
              ("temporal_attn_mfma", 2))


def test_default_path_kernels_keep_their_occupancy():
    """Register budget of every MFMA kernel in the library against the occupancy it was designed for."""
    import glob
    import shutil
    import subprocess
    import tempfile
    tools = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{tools}/llvm-objdump") and os.path.exists(f"{tools}/llvm-readelf")):
        import pytest
        pytest.skip("ROCm LLVM binutils not installed")
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"), so)
        subprocess.run([f"{tools}/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp, check=True)
        checked, bad = 0, []
        for f in sorted(glob.glob(so + ".*gfx950")):
            notes = subprocess.run([f"{tools}/llvm-readelf", "--notes", f], capture_output=True, text=True, check=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
                name, body = m.group(1), m.group(2)
                regs = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
                ag = re.search(r"\.agpr_count:\s+(\d+)", body)
                regs += int(ag.group(1)) if ag else 0
                waves = min(8, 512 // max(8, (regs + 7) // 8 * 8))
                for pat, want in _MIN_WAVES:
                    if re.search(pat, name):
                        checked += 1
                        if waves < want:
                            bad.append((name, regs, waves, want))
                        break
        assert checked >= 40, checked
        assert not bad, bad
    finally:
        shutil.rmtree(tmp)


def test_gemm_and_conv_k_loops_have_no_scratch_reloads():
    """Round 4: a scratch reload is a VMEM load, and its vmcnt wait -- vmcnt retires in order -- also waits for the LDS-DMA stage
    requested in front of it (cost a tile-boundary rewrite 12 % and the persistent convolution 5 % before it was found).  The
    kernels of the default path may spill a few registers, but never between the first and the last MFMA of their K loop."""
    import glob
    import shutil
    import subprocess
    import tempfile
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{tools}/llvm-objdump"):
        import pytest
        pytest.skip("ROCm LLVM binutils not installed")
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"), so)
        subprocess.run([f"{tools}/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp, check=True)
        checked, bad = 0, []
        for f in sorted(glob.glob(so + ".*gfx950")):
            syms = subprocess.run([f"{tools}/llvm-objdump", "-t", f], capture_output=True, text=True, check=True).stdout
            if "conv_ring_kernel" not in syms and "conv_igemm_kernel" not in syms:
                continue
            asm = subprocess.run([f"{tools}/llvm-objdump", "-d", f], capture_output=True, text=True, check=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(_ZN5im360\d+conv_(?:ring|igemm)_kernel\w+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.S | re.M):
                name, body = m.group(1), m.group(2).split("\n")
                # default-path kernels: the staggered GEMM loop (MODE 3) and the two-stage 256 x 320 convolution tile
                if not (re.search(r"conv_ring_kernelIDF16[b_]Li\dELi\dELb1ELb1ELi3E", name) or re.search(r"conv_igemm_kernelIDF16[b_]Li64ELi4ELi2ELi2ELi5E", name)):
                    continue
                mf = [i for i, l in enumerate(body) if "v_mfma" in l]
                assert mf, name
                checked += 1
                inside = [l.strip() for l in body[mf[0]:mf[-1]] if "scratch_load" in l or "scratch_store" in l]
                if inside:
                    bad.append((name, len(inside)))
        assert checked >= 20, checked
        assert not bad, bad
    finally:
        shutil.rmtree(tmp)


def test_default_path_kernels_do_not_spill():
    """Per-kernel metadata of the gfx950 code objects inside the library (llvm-objdump --offloading + llvm-readelf --notes):
    at most 16 spilled VGPRs / 64 bytes of scratch outside the listed A/B variants."""
    import glob
    import shutil
    import subprocess
    import tempfile
    tools = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{tools}/llvm-objdump") and os.path.exists(f"{tools}/llvm-readelf")):
        import pytest
        pytest.skip("ROCm LLVM binutils not installed")
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"), so)
        subprocess.run([f"{tools}/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp, check=True)
        seen, bad = 0, []
        for f in sorted(glob.glob(so + ".*gfx950")):
            notes = subprocess.run([f"{tools}/llvm-readelf", "--notes", f], capture_output=True, text=True, check=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
                name, body = m.group(1), m.group(2)
                spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
                seen += 1
                if (spill > 16 or scratch > 64) and not any(a in name for a in _SPILL_ALLOWED):
                    bad.append((name, spill, scratch))
        assert seen >= 100, seen
        assert not bad, bad
    finally:
        shutil.rmtree(tmp)


def test_conv_k_split_planner_rule():
    """im360_conv_ksplit_plan is host code: the shipped rule (knob conv_ksplit = 1) splits 3 x 3 launches BELOW 512 large tiles only (measured: the
    level-3 convolutions of cfg2 gain, the 640-tile level-2 ones lose -- profiles/r06_conv_ksplit_ab.log), never the upsample / wrap-addressed
    ones, and never launches that write GroupNorm statistics from fewer than 512 tiles."""
    lib = kernels.lib()
    plan = lambda N, H, W, Cin, Cout, taps=9, up=0, wrap=0, gn=0: lib.im360_conv_ksplit_plan(N, H, W, Cin, Cout, taps, up, wrap, gn)
    assert plan(640, 4, 4, 1280, 1280) == 3            # cfg2 perspective level 3: 160 tiles -> three parts (480 workgroups)
    assert plan(640, 8, 8, 1280, 1280) == 1            # level 2: 640 tiles = 2.5 rounds: unsplit
    assert plan(640, 16, 16, 640, 640) == 1            # level 1: 1280 tiles
    assert plan(32, 8, 16, 1280, 1280) in (2, 3, 4)    # panorama level 3 (pre-padded window, no wrap flag): 64 tiles
    assert plan(32, 8, 20, 1280, 1280, wrap=1) == 1    # wrap addressing: left alone by the rule
    assert plan(32, 16, 36, 1280, 1280) == 1           # 288 tiles: above the rule's liveness bound (248 tiles = 31 waiting owners per XCD)
    assert plan(640, 4, 4, 1280, 1280, up=1) == 1 and plan(640, 4, 4, 1280, 1280, taps=1) == 1
    assert plan(640, 4, 4, 1280, 1280, gn=1) == 1      # statistics epilogue needs >= 512 tiles
    assert plan(640, 4, 4, 1280, 1000) == 1 and plan(640, 4, 4, 96, 1280) == 1      # not the 256 x 320 tile / not whole 64-channel chunks
