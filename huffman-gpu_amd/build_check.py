#!/usr/bin/env python3
"""Build gate of libhvlc.so (hvlc.build() runs it; `python huffman-gpu_amd/build_check.py [lib]` by hand).
Scans the gfx950 code object for the sequence that faulted in round 2:

    v_readlane_b32 s63, v79, 3      ; VALU writes an SGPR (here: the reload of a spilled pointer)
    v_readlane_b32 s61, v79, 1
    s_nop 3                         ; the 5 wait states the compiler inserts for "VALU writes SGPR -> VMEM reads it"
    global_load_dwordx2 v[14:15], v2, s[62:63] sc1

On MI355X the load was seen (rarely: once per ~10^8 executions, in the scouts' polling loop) to use the OLD value of
s63: "Memory access fault ... on address 0xffff1ca03000" = {new low half, stale high half = -1}.  The same code with
the s_nop lengthened never faulted (profiles/r02_hazard_ab.txt).  The root cause (tools/hazard_repro.hip): an s_wakeup
from another wave of the workgroup ends an s_nop in progress.  hazard_pass.py therefore splits the multi-state s_nop of
kernels that use s_wakeup and puts a SALU
copy of the register onto itself (interlocked in hardware) in front of every such memory instruction, on any path; this
script is the independent check of the binary that ships:

    python huffman-gpu_amd/build_check.py [libhvlc.so] [--min-wait-states 12]

exits 1 and lists every multi-state `s_nop` left in a kernel that uses s_wakeup (the root cause: a wake-up from another
wave ends an s_nop in progress) and every VMEM instruction that reads an SGPR some VALU instruction wrote fewer than --min-wait-states wait
states earlier with no SALU write of it in between (straight-line distance in the disassembly; a label in between does
not reset it, which is conservative for fall-through; paths through taken branches are hazard_pass.py's data-flow
analysis).  SMEM addresses are interlocked like SALU reads and not listed.  hvlc.build() refuses a library that fails;
tests/test_build_check.py pins the scanner and the pass on the sequence above."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def _llvm_bin():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hvlc_hazard_pass_for_check", os.path.join(HERE, "hazard_pass.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.llvm_bin()


LLVM = _llvm_bin()

SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]|\b(vcc)\b|\b(vcc_lo)\b|\b(vcc_hi)\b")
VMEM_PREFIX = ("global_", "flat_", "scratch_", "buffer_", "tbuffer_")
SMEM_PREFIX = ("s_load_", "s_buffer_load_", "s_store_", "s_atomic_", "s_scratch_", "s_dcache_")
CARRY_OUT = ("_co_", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")


def sregs(text):
    """SGPR numbers named in an operand string (vcc = 106, 107)."""
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        elif m.group(2):
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.update((106, 107))
        elif m.group(5):
            out.add(106)
        elif m.group(6):
            out.add(107)
    return out


def extract_code_object(lib, workdir):
    tmp_lib = os.path.join(workdir, "lib.so")
    with open(lib, "rb") as f, open(tmp_lib, "wb") as g:
        g.write(f.read())
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp_lib], check=True, stdout=subprocess.DEVNULL)
    cos = [os.path.join(workdir, n) for n in os.listdir(workdir) if "amdgcn" in n and "gfx950" in n]
    if not cos:
        raise RuntimeError("no gfx950 code object in " + lib)
    return cos[0]


def multi_state_nops(disassembly):
    """[(kernel, instruction)]: every `s_nop N`, N > 0, in a kernel that also executes s_wakeup (an s_wakeup from another
    wave of the workgroup ends an s_nop in progress: hazard_pass.py splits them)."""
    kernels, cur = {}, None
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [False, []])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins.startswith("s_wakeup"):
            cur[0] = True
        elif ins.startswith("s_nop") and int(ins.split()[1], 0) > 0:
            cur[1].append(ins)
    return [(k, i) for k, (wakes, nops) in kernels.items() if wakes for i in nops]


def scan(disassembly, min_ws):
    findings = []
    kernel = "?"
    last_write = {}          # sgpr -> (wait-state clock of the VALU write, text)
    clock = 0
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel, last_write, clock = m.group(1), {}, 0
            continue
        if not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        mnem, _, ops = ins.partition(" ")
        if mnem.startswith(VMEM_PREFIX):      # (SMEM addresses are interlocked in hardware like any SALU read)
            for r in sorted(sregs(ops)):
                if r in last_write and clock - last_write[r][0] - 1 < min_ws:
                    findings.append((kernel, ins, "s%d" % r, clock - last_write[r][0] - 1, last_write[r][1]))
        if mnem.startswith("v_"):
            parts = [p.strip() for p in ops.split(",")]
            written = sregs(parts[0]) if parts else set()
            if len(parts) > 1 and any(k in mnem for k in CARRY_OUT):
                written |= sregs(parts[1])
            if mnem.startswith("v_cmp") and "_e64" not in mnem and "_sdwa" not in mnem and not mnem.startswith("v_cmpx"):
                written |= {106, 107}                       # e32 compares write vcc implicitly
            for r in written:
                last_write[r] = (clock, ins)
        elif mnem.startswith("s_") and not mnem.startswith(SMEM_PREFIX):
            parts = [p.strip() for p in ops.split(",")]     # a SALU write replaces the VALU one (interlocked in hardware)
            # (not s_cmov*: it writes nothing when SCC = 0, the VALU value may still be the one that is read)
            if parts and mnem not in ("s_nop", "s_waitcnt", "s_sleep", "s_branch", "s_barrier", "s_cmov_b32", "s_cmov_b64", "s_cmovk_i32") \
               and not mnem.startswith("s_cbranch") and not mnem.startswith("s_cmp") and not mnem.startswith("s_bitcmp"):
                for r in sregs(parts[0]):
                    last_write.pop(r, None)
        clock += 1
        if mnem == "s_nop":
            clock += int(ops.strip() or "0", 0)
    return findings


def check(lib, min_ws=12):
    with tempfile.TemporaryDirectory() as d:
        co = extract_code_object(lib, d)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    return scan(dis, min_ws) + [(k, i, "-", 0, "a kernel that uses s_wakeup must not rely on a multi-state s_nop") for k, i in multi_state_nops(dis)]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("lib", nargs="?", default=os.path.join(HERE, "libhvlc.so"))
    ap.add_argument("--min-wait-states", type=int, default=12)
    a = ap.parse_args()
    found = check(a.lib, a.min_wait_states)
    for kernel, ins, reg, dist, writer in found:
        print(f"{kernel}: `{ins}`: {writer}" if reg == "-" else f"{kernel}: `{ins}` reads {reg} {dist} wait state(s) after `{writer}`")
    print(f"{a.lib}: {len(found)} finding(s) (multi-state s_nop in a kernel that uses s_wakeup; VALU-written SGPR -> memory "
          f"address closer than {a.min_wait_states} wait states)")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
