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


# ---------------------------------------------------------------------------------------------------------
# hazard_pass.py: the copies it inserts into the device assembly
# ---------------------------------------------------------------------------------------------------------
spec2 = importlib.util.spec_from_file_location("hvlc_hazard_pass", os.path.join(PKG, "hazard_pass.py"))
hp = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(hp)

KERNEL = """\t.text
\t.type\tk,@function
k:                                      ; @k
; %bb.0:
{body}\ts_endpgm
.Lfunc_end0:
\t.size\tk, .Lfunc_end0-k
"""


def _patched(body, window=48):
    text, rep = hp.patch_asm(KERNEL.format(body=body), window)
    return [l.strip().split(";")[0].strip() for l in text.splitlines() if l.startswith("\t") and not l.startswith("\t.")], rep["k"]


def test_pass_puts_salu_copies_in_front_of_the_faulting_load():
    body = ("\tv_readlane_b32 s62, v79, 2\n\tv_readlane_b32 s63, v79, 3\n\tv_readlane_b32 s61, v79, 1\n\ts_nop 3\n"
            "\tglobal_load_dwordx2 v[14:15], v2, s[62:63] sc1\n")
    ins, n = _patched(body)
    assert n == 2
    i = ins.index("global_load_dwordx2 v[14:15], v2, s[62:63] sc1")
    assert ins[i - 2:i] == ["s_mov_b32 s62, s62", "s_mov_b32 s63, s63"]
    # the straight-line scanner accepts the result: the SALU copies supersede the VALU writes
    dis = "0 <k>:\n" + "\n".join("\t" + x for x in ins) + "\n"
    assert bc.scan(dis, 12) == []


def test_pass_follows_branches_and_loops():
    # the VALU write sits behind the load in program order: only the loop's back edge brings it in front
    body = ("\ts_mov_b64 s[4:5], s[0:1]\n.LBB0_1:\n\tglobal_load_dword v1, v0, s[4:5]\n\tv_readfirstlane_b32 s4, v1\n"
            "\ts_cbranch_scc1 .LBB0_1\n")
    ins, n = _patched(body)
    assert n == 1 and ins[ins.index("global_load_dword v1, v0, s[4:5]") - 1] == "s_mov_b32 s4, s4"
    # taken branch over many instructions: the write is old on the fall-through path only
    pad = "".join("\tv_add_u32_e32 v2, v2, v2\n" for _ in range(60))
    body = ("\tv_readfirstlane_b32 s6, v1\n\ts_cbranch_scc1 .LBB0_2\n" + pad + ".LBB0_2:\n\tglobal_store_dword v0, v1, s[6:7]\n")
    ins, n = _patched(body)
    assert n == 1 and ins[ins.index("global_store_dword v0, v1, s[6:7]") - 1] == "s_mov_b32 s6, s6"


def test_pass_leaves_safe_code_alone():
    pad = "".join("\tv_add_u32_e32 v2, v2, v2\n" for _ in range(60))
    # written by SALU; written by VALU long ago; VALU-written but rewritten by SALU; SMEM address (interlocked)
    body = ("\ts_mov_b64 s[4:5], s[0:1]\n\tglobal_load_dword v1, v0, s[4:5]\n"
            "\tv_readfirstlane_b32 s8, v1\n" + pad + "\tglobal_load_dword v1, v0, s[8:9]\n"
            "\tv_readfirstlane_b32 s10, v1\n\ts_add_u32 s10, s10, 4\n\tglobal_load_dword v1, v0, s[10:11]\n"
            "\tv_readfirstlane_b32 s12, v1\n\ts_load_dword s14, s[12:13], 0x0\n")
    ins, n = _patched(body)
    assert n == 0
    # a shorter window trusts more
    body = "\tv_readfirstlane_b32 s8, v1\n" + "".join("\ts_nop 0\n" for _ in range(20)) + "\tglobal_load_dword v1, v0, s[8:9]\n"
    assert _patched(body, 48)[1] == 1 and _patched(body, 16)[1] == 0


def test_pass_is_complete_and_idempotent_on_the_real_device_code(tmp_path):
    """The device code of libhvlc as the compiler writes it today: after one pass a second one finds nothing left to do
    (every VALU-written address register is covered on every path), and the pass only ADDS s_mov copies."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    src = os.path.join(PKG, "csrc", "hvlc_api.hip")
    asm = str(tmp_path / "dev.s")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-command-line-argument", "-DHVLC_HAZARD_PASS", "-S",
                    "--cuda-device-only", "-o", asm, src], check=True, cwd=os.path.dirname(src))
    text = open(asm).read()
    nops1, nops2 = {}, {}
    once, rep1 = hp.patch_asm(text, nops=nops1)
    twice, rep2 = hp.patch_asm(once, nops=nops2)
    assert any("k_encode" in k for k in rep1), "kernels not recognised in the assembly"
    assert sum(rep2.values()) == 0 and sum(nops2.values()) == 0 and twice == once
    assert sum(nops1.values()) > 0 and all("k_encode" in k for k, v in nops1.items() if v)    # the one kernel with s_wakeup
    before = set(text.splitlines())
    added = [l for l in once.splitlines() if l not in before]
    # new text: the SALU copies, and the first `s_nop 0` of every split s_nop (it carries a comment; the others read like
    # the compiler's own)
    assert len(added) == sum(rep1.values()) + sum(nops1.values())
    assert all("hazard pass" in l and l.strip().startswith(("s_mov_b32", "s_nop 0")) for l in added)


def test_pass_against_exhaustive_path_search_on_random_programs():
    """Random control-flow graphs over four SGPRs: after the pass, an exhaustive search over every (block, register ages)
    state reachable on ANY path -- not the pass's own merged data flow -- finds no memory instruction whose address
    register a VALU instruction wrote fewer than `window` wait states before with no SALU write in between."""
    import random
    W = 6
    rng = random.Random(5)
    for trial in range(60):
        nb = rng.randint(2, 7)
        body = []
        for b in range(nb):
            body.append(f".LBB0_{b}:\n")
            for _ in range(rng.randint(0, 6)):
                k, r = rng.random(), rng.randrange(0, 8, 2)
                if k < 0.30:   body.append(f"\tv_readfirstlane_b32 s{r + rng.randint(0, 1)}, v1\n")
                elif k < 0.40: body.append(f"\ts_mov_b32 s{r + rng.randint(0, 1)}, s9\n")
                elif k < 0.60: body.append(f"\ts_nop {rng.randint(0, 3)}\n")
                elif k < 0.70: body.append("\tv_add_u32_e32 v2, v2, v2\n")
                else:          body.append(f"\tglobal_load_dword v1, v0, s[{r}:{r + 1}]\n")
            k = rng.random()
            if k < 0.45:   body.append(f"\ts_cbranch_scc1 .LBB0_{rng.randrange(nb)}\n")
            elif k < 0.60: body.append(f"\ts_branch .LBB0_{rng.randrange(nb)}\n")
        text, _ = hp.patch_asm(KERNEL.format(body="".join(body)), W)
        # ---- independent check: explicit states, every path
        lines = [l for l in text.splitlines() if l.strip() and not l.strip().startswith((";", ".t", ".s")) and l.strip() != "k:"]
        prog, labels = [], {}
        for l in lines:
            t = l.split(";")[0].strip()
            if t.endswith(":"):
                labels[t[:-1]] = len(prog)
            elif t and not t.startswith("."):
                prog.append(t)
        seen, todo = set(), [(0, (W,) * 8)]
        while todo:
            pc, age = todo.pop()
            while pc < len(prog):
                if (pc, age) in seen:
                    break
                seen.add((pc, age))
                mnem, _, ops = prog[pc].partition(" ")
                a = list(age)
                if mnem.startswith("global_"):
                    lo = int(ops.split("s[")[1].split(":")[0])
                    assert a[lo] >= W and a[lo + 1] >= W, (trial, pc, prog[pc], a, "\n".join(prog))
                ws = 1 + (int(ops) if mnem == "s_nop" else 0)
                a = [min(W, x + ws) for x in a]
                if mnem == "v_readfirstlane_b32":
                    a[int(ops.split(",")[0][1:])] = 0
                elif mnem == "s_mov_b32":
                    a[int(ops.split(",")[0][1:])] = W
                age = tuple(a)
                if mnem == "s_branch":
                    pc = labels[ops.strip()]; continue
                if mnem == "s_cbranch_scc1":
                    todo.append((labels[ops.strip()], age))
                if mnem == "s_endpgm":
                    break
                pc += 1


def test_pass_splits_multi_state_nops_only_where_waves_wake_each_other():
    """An s_wakeup from another wave of the workgroup ends an `s_nop N` in progress (tools/hazard_repro.hip): in a kernel
    that contains s_wakeup every `s_nop N` becomes N + 1 separate `s_nop 0`; other kernels are left alone; the gate finds
    what a build without the pass would ship."""
    body = "\tv_add_u32_dpp v1, v0, v0 row_shr:1\n\ts_nop 1\n\tv_add_u32_dpp v1, v1, v1 row_shr:2\n\ts_nop 3\n\ts_nop 0\n"
    ins, _ = _patched(body + "\ts_wakeup\n")
    assert ins.count("s_nop 0") == 2 + 4 + 1 and not any(i.startswith("s_nop") and i != "s_nop 0" for i in ins)
    nops = {}
    text, _ = hp.patch_asm(KERNEL.format(body=body + "\ts_wakeup\n"), 48, nops)
    assert nops == {"k": 2}
    ins, _ = _patched(body)                                   # no s_wakeup in the kernel: nothing to do
    assert ins.count("s_nop 1") == 1 and ins.count("s_nop 3") == 1 and ins.count("s_nop 0") == 1
    dis = "0 <k>:\n\ts_nop 1\n\ts_wakeup\n\ts_nop 0\n10 <quiet>:\n\ts_nop 3\n"
    assert bc.multi_state_nops(dis) == [("k", "s_nop 1")]


# ---- round 3: the gate is the only door and the conditional SALU forms -----------------------
import shutil
import subprocess


def test_a_conditional_move_does_not_count_as_the_salu_write():
    """s_cmov writes nothing when SCC = 0: the VALU-written value may still be the one the load reads."""
    seq = ("0 <k>:\n\tv_readlane_b32 s4, v1, 0\n\tv_readlane_b32 s5, v1, 1\n\ts_cmov_b64 s[4:5], s[8:9]\n"
           "\tglobal_load_dword v2, v0, s[4:5]\n")
    assert [f[2] for f in bc.scan(seq, 12)] == ["s4", "s5"]
    assert bc.scan(seq.replace("s_cmov_b64", "s_mov_b64"), 12) == []
    hp = _hazard_pass()
    asm = ("\tv_readlane_b32 s4, v1, 0\n\tv_readlane_b32 s5, v1, 1\n\ts_cmov_b64 s[4:5], s[8:9]\n"
           "\tglobal_load_dword v2, v0, s[4:5]\n\ts_endpgm\n").splitlines(keepends=True)
    out, n, _ = hp.patch_function(asm, hp.WINDOW)
    assert n == 2 and sum("hazard pass" in l for l in out) == 2
    out2, n2, _ = hp.patch_function([l.replace("s_cmov_b64", "s_mov_b64") for l in asm], hp.WINDOW)
    assert n2 == 0


def test_a_scalar_store_operand_is_not_a_destination():
    hp = _hazard_pass()
    asm = ("\tv_readlane_b32 s4, v1, 0\n\ts_store_dword s4, s[8:9], 0x0\n\tbuffer_load_dword v2, off, s[12:15], s4\n\ts_endpgm\n"
           ).splitlines(keepends=True)
    _, n, _ = hp.patch_function(asm, hp.WINDOW)
    assert n == 1                      # s4 is still "VALU-written" at the load: the store only read it


def _hazard_pass():
    spec = importlib.util.spec_from_file_location("hvlc_hazard_pass_t", os.path.join(PKG, "hazard_pass.py"))
    hp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hp)
    return hp


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_a_bare_hipcc_build_of_the_library_is_refused():
    """libhvlc.so's source compiles only through hazard_pass.py (which defines HVLC_HAZARD_PASS): the binary a plain
    `hipcc -shared csrc/hvlc_api.hip` would give is the one that faulted in round 2."""
    src = os.path.join(PKG, "csrc", "hvlc_api.hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "--cuda-host-only", "-fsyntax-only", src], capture_output=True, text=True)
    assert r.returncode != 0 and "hazard_pass.py" in r.stderr
    ok = subprocess.run(["hipcc", "--offload-arch=gfx950", "--cuda-host-only", "-fsyntax-only", "-DHVLC_HAZARD_PASS", src],
                        capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-2000:]


def test_the_makefiles_build_the_library_through_the_pass_and_the_gate():
    """The C hosts' Makefile depends on ../libhvlc.so, whose only rule runs hazard_pass.py and build_check.py."""
    r = subprocess.run(["make", "-n", "-B", "-C", PKG, "libhvlc.so"], capture_output=True, text=True)
    assert r.returncode == 0 and "hazard_pass.py" in r.stdout and "build_check.py" in r.stdout and "hipcc -shared" not in r.stdout
    host = open(os.path.join(PKG, "host", "Makefile")).read()
    assert "$(HERE)../libhvlc.so:" in host and "$(MAKE) -C $(HERE).. libhvlc.so" in host
