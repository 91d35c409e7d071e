"""The build gate (huffman-gpu_amd/build_check.py): the scanner finds the instruction sequence that faulted on MI355X in
round 2, accepts the same code once the SGPR is far enough from the load, and the library that ships passes it."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "huffman-gpu_amd")
spec = importlib.util.spec_from_file_location("hvlc_build_check", os.path.join(PKG, "build_check.py"))
bc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bc)

# What llvm-objdump printed for the faulting build (commit fd389d9, k_encode's polling loop).
FAULTING = """
0000000000006900 <_ZN4hvlc8k_encodeENS_12EncodeParamsE>:
\ts_and_saveexec_b64 s[52:53], s[16:17]                      // 00000000F4EC: BEB42010
\ts_cbranch_execz 11                                         // 00000000F4F0: BF88000B
\tv_readlane_b32 s60, v79, 0                                 // 00000000F4F4: D289003C 0001014F
\tv_readlane_b32 s62, v79, 2                                 // 00000000F4FC: D289003E 0001054F
\tv_readlane_b32 s63, v79, 3                                 // 00000000F504: D289003F 0001074F
\tv_readlane_b32 s61, v79, 1                                 // 00000000F50C: D289003D 0001034F
\ts_nop 3                                                    // 00000000F514: BF800003
\tglobal_load_dwordx2 v[14:15], v2, s[62:63] sc1             // 00000000F518: DE548000 0E3E0002
\ts_or_b64 exec, exec, s[52:53]                              // 00000000F520: 87FE347E
"""


def test_scanner_finds_the_faulting_sequence():
    found = bc.scan(FAULTING, 12)
    assert [(f[2], f[3]) for f in found] == [("s62", 6), ("s63", 5)]
    assert all("global_load_dwordx2" in f[1] and "v_readlane_b32" in f[4] for f in found)


def test_scanner_counts_nops_and_salu_rewrites():
    padded = FAULTING.replace("\ts_nop 3 ", "\ts_nop 7\n\ts_nop 7 ")
    assert [(f[2], f[3]) for f in bc.scan(padded, 12)] == []
    assert len(bc.scan(padded, 20)) == 2
    # an SALU instruction that rewrites the register is interlocked in hardware: not a finding
    salu = FAULTING.replace("\ts_nop 3 ", "\ts_mov_b64 s[62:63], s[4:5] ")
    assert bc.scan(salu, 12) == []
    # vcc written by an e32 compare, then used as a scalar offset
    vcc = "0 <k>:\n\tv_cmp_eq_u32_e32 vcc, 1, v0\n\tbuffer_load_dword v1, off, s[0:3], vcc_lo\n"
    assert [(f[2], f[3]) for f in bc.scan(vcc, 12)] == [("s106", 0)]


def test_shipped_library_passes_the_gate():
    lib = os.path.join(PKG, "libhvlc.so")
    if not os.path.exists(lib):
        pytest.skip("libhvlc.so not built")
    if not os.path.exists(os.path.join(bc.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    assert bc.check(lib) == []
