#!/usr/bin/env python3
"""ISA lint over EVERY kernel of libjrlqp_hip.so (gfx950): the VALU-written-SGPR hazards this code base has met on hardware.

  R1  lane select too close to its producer.  `v_readlane_b32 D, vS, sN` / `v_writelane_b32 vD, X, sN` whose lane-select SGPR
      sN was written by a VALU instruction (v_readfirstlane_b32, v_readlane_b32, a v_cmp / carry-out into SGPRs) FEWER THAN
      FIVE wait states earlier.  hipcc inserts the four its hazard table asks for; on MI355X four were measured not to be
      enough (9-18 of 61 parity tests failing, different problems on every run, only where exactly four
      instructions sat between the two).  A wait state is one issued instruction, `s_nop N` counts N + 1.
  R2  read-lane of a VGPR right behind the VALU instruction that wrote it (no wait state at all).  The compiler separates its
      own instructions; it cannot see into inline-assembly units (device_common.h fma_neg_bcast), and a unit issued right behind
      the producer of its source read a stale half (measured: factors off by 1e-8).

  R3  memory instruction addressed through an SGPR too close to the VALU write of that SGPR.  `global_load_dwordx2 v, vOFF, s[0:1]`
      (or any global_ / flat_ / scratch_ / buffer_ / s_load_ / s_buffer_ instruction naming an SGPR) whose SGPR was written by a VALU
      instruction -- in practice the v_readlane_b32 that reloads a spilled SGPR pair -- FEWER THAN EIGHT wait states earlier.  The
      hazard table asks for five and hipcc emits exactly five (`s_nop 4`); round 3 caught an embedded-size instantiation of the
      batched kernel (the ones with SGPR spills) faulting about once in 150 launches on `global_load_dwordx2 v[132:133], v123,
      s[0:1]` five wait states behind `v_readlane_b32 s1, v205, 31`, at addresses made of whatever s[0:1] held BEFORE the reload
      (rocgdb, precise memory mode: tools/gdb_fault.sh).

All are checked on the DISASSEMBLY of the built objects (every instruction, compiler-emitted and inline assembly alike).  Wait states
are counted along the listing AND along every branch edge: a first pass records, at each s_branch / s_cbranch, how many wait states
each pending VALU-written SGPR has behind it (the branch itself included); at the target label the youngest of "fall-through" and
"every branch that lands here" counts -- a consumer at a branch target is not credited with the instructions the taken branch skipped
(round 3 counted in listing order only).

usage: isa_lint.py [--min-states 5] [--objdir jrl-qp_amd/csrc] [--list] [objects...]      exit status 1 on any finding
       isa_lint.py --fix in.s out.s        the SAME rules applied to a `hipcc -S --offload-device-only` listing: every finding
                                           gets the missing wait states as an `s_nop` in front of the consumer
The build (jrl-qp_amd/csrc/Makefile) compiles every .hip file to device assembly, runs --fix over it, assembles and bundles
the result; __graft_entry__.build() then runs the lint over the disassembly of the built objects: a finding fails the build.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INSN = re.compile(r"^\s+([a-z][a-z0-9_]+)(?:\s+(.*?))?\s*(?://.*|;.*)?$")  # llvm-objdump ("// addr: bytes") and hipcc -S ("; comment") lines
FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
MEM_STATES = 8  # R3: wait states between a VALU write of an SGPR and a memory instruction addressed through it (hazard table: 5)
CARRY_OUT = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
             "v_div_scale_f32", "v_div_scale_f64", "v_mad_u64_u32", "v_mad_i64_i32")


def regs_of(tok, kind):
    """register numbers named by one operand token: 's12', 's[4:5]', 'v[82:83]', '|v[2:3]|', '-v5' ..."""
    tok = tok.strip().strip("|").lstrip("-").strip("|")
    m = re.fullmatch(kind + r"(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(kind + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def split_ops(s):
    return [t.strip() for t in s.split(",")] if s else []


def disassemble(obj, workdir):
    base = os.path.join(workdir, os.path.basename(obj))
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, base + ".fat"])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={base}.fat", f"--output={base}.co", "--unbundle"])
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--symbolize-operands", "--mcpu=gfx950", base + ".co"], check=True, capture_output=True,
                          text=True).stdout


LABEL = re.compile(r"^(?:<(L\d+)>|(\.L[\w$.]+)):")  # llvm-objdump --symbolize-operands / hipcc -S


def lint_listing(text, min_states, name, fix_out=None):
    """fix_out: a list that receives the lines of the listing with the missing wait states inserted (then nothing is a finding)."""
    # pass 1: the age of every pending VALU-written SGPR at each branch, per target label -- repeated with the ages of the previous round
    # imported at the labels until nothing changes, so that an age carried over one edge is carried over the next one too
    # (write -> branch A -> label A: branch B -> label B: consumer); pass 2: the lint proper
    ages = {}
    for _ in range(8):
        nxt = {}
        _lint_pass(text, min_states, name, None, nxt, ages, fix_out is not None)
        for key, regs in ages.items():  # ages only ever get younger: keep the youngest seen
            tgt = nxt.setdefault(key, {})
            for sreg, v in regs.items():
                if sreg not in tgt or v[0] < tgt[sreg][0]:
                    tgt[sreg] = v
        if nxt == ages:
            break
        ages = nxt
    return _lint_pass(text, min_states, name, fix_out, None, ages, fix_out is not None)


def _lint_pass(text, min_states, name, fix_out, collect, label_ages, s_listing):
    findings = []
    stats = dict(functions=0, instructions=0, lane_selects=0, valu_sgpr_selects=0, readlanes=0)
    func = "?"
    # per function: wait-state clock, last VALU write of each SGPR (clock value), previous instruction's VALU VGPR writes
    clock = 0
    sgpr_valu_write = {}
    prev_vgpr_writes = set()
    prev_text = ""
    for line in text.split("\n"):
        if fix_out is not None:
            fix_out.append(line)
        fm = FUNC.match(line)
        if fm is None and s_listing:
            lm = re.match(r"^([A-Za-z_$.][\w$.]*):", line)  # -S listing: a label; function symbols start a new scope
            if lm and not lm.group(1).startswith(".L"):
                fm = lm
        if fm and re.fullmatch(r"L\d+", fm.group(1)):
            line = "<" + fm.group(1) + ">:"  # llvm-objdump --symbolize-operands prints a branch target like a symbol: a label, not a function
            fm = None
        if fm:
            func = fm.group(1)
            stats["functions"] += 1
            clock = 0
            sgpr_valu_write = {}
            prev_vgpr_writes = set()
            continue
        lb = LABEL.match(line.strip())
        if lb:
            # a branch target: for every SGPR some branch arrives with, the youngest write counts
            key = (func, lb.group(1) or lb.group(2))
            if label_ages is not None:
                for sreg, (age, txt) in label_ages.get(key, {}).items():
                    w = clock - age
                    if sreg not in sgpr_valu_write or sgpr_valu_write[sreg][0] < w:
                        sgpr_valu_write[sreg] = (w, txt if txt.endswith("(via a branch)") else txt + " (via a branch)")
            # (prev_vgpr_writes stays: the fall-through predecessor still immediately precedes the label's first instruction -- rule R2)
            continue
        m = INSN.match(line)
        if not m or line.lstrip().startswith((".", ";", "//", "#")):
            continue
        mn, ops = m.group(1), split_ops(m.group(2) or "")
        stats["instructions"] += 1
        if mn == "s_nop":
            clock += int(ops[0], 0) + 1
            prev_vgpr_writes = set()
            prev_text = line.strip()
            continue
        is_valu = mn.startswith("v_")
        # ---- consumers
        if mn in ("v_readlane_b32", "v_writelane_b32") and len(ops) == 3:
            sel = regs_of(ops[2], "s")
            stats["lane_selects"] += 1
            if mn == "v_readlane_b32":
                stats["readlanes"] += 1
                src = regs_of(ops[1], "v")
                if src & prev_vgpr_writes:
                    if fix_out is not None:
                        fix_out.insert(len(fix_out) - 1, "\ts_nop 0 ; isa_lint R2")
                        clock += 1
                        stats["fixed"] = stats.get("fixed", 0) + 1
                    else:
                        findings.append((name, func, "R2", f"'{prev_text.split('//')[0].strip()}' feeds '{line.strip().split('//')[0].strip()}' with no wait state"))
            for s in sel:
                if s in sgpr_valu_write:
                    stats["valu_sgpr_selects"] += 1
                    states = clock - sgpr_valu_write[s][0]
                    if states < min_states:
                        if fix_out is not None:
                            fix_out.insert(len(fix_out) - 1, f"\ts_nop {min_states - states - 1} ; isa_lint R1")
                            clock += min_states - states
                            stats["fixed"] = stats.get("fixed", 0) + 1
                        else:
                            findings.append((name, func, "R1", f"{states} wait states between '{sgpr_valu_write[s][1]}' and '{line.strip().split('//')[0].strip()}'"))
        # R3: a memory instruction that names a VALU-written SGPR (address base / offset / resource)
        if mn.startswith(("global_", "flat_", "scratch_", "buffer_", "s_load_", "s_buffer_", "s_scratch_", "tbuffer_")):
            used = set()
            for t in ops[(1 if (mn.startswith("s_load_") or mn.startswith("s_buffer_")) else 0):]:
                used |= regs_of(t.split()[0] if t else t, "s")
            worst = None
            for sreg in used:
                if sreg in sgpr_valu_write:
                    states = clock - sgpr_valu_write[sreg][0]
                    if states < MEM_STATES and (worst is None or states < worst[0]):
                        worst = (states, sgpr_valu_write[sreg][1])
            if worst is not None:
                if fix_out is not None:
                    fix_out.insert(len(fix_out) - 1, f"\ts_nop {MEM_STATES - worst[0] - 1} ; isa_lint R3")
                    clock += MEM_STATES - worst[0]
                    stats["fixed"] = stats.get("fixed", 0) + 1
                else:
                    findings.append((name, func, "R3", f"{worst[0]} wait states between '{worst[1]}' and '{line.strip().split('//')[0].strip()}'"))
        if collect is not None and mn.startswith(("s_cbranch", "s_branch")) and ops:
            tgt = collect.setdefault((func, ops[0].strip()), {})
            for sreg, (w, txt) in sgpr_valu_write.items():
                age = clock + 1 - w  # (the branch itself is a wait state)
                if age < max(min_states, MEM_STATES) and (sreg not in tgt or age < tgt[sreg][0]):
                    tgt[sreg] = (age, txt)
        # ---- producers
        written_s = set()
        if is_valu and ops:
            written_s |= regs_of(ops[0], "s")
            if any(mn.startswith(c) for c in CARRY_OUT) and len(ops) > 1:
                written_s |= regs_of(ops[1], "s")
        clock += 1
        for s in written_s:
            sgpr_valu_write[s] = (clock, line.strip().split("//")[0].strip())
        if not is_valu:
            # an SALU / SMEM write of the same SGPR supersedes the VALU write (no lane-select hazard behind the scalar unit)
            if ops and (mn.startswith("s_") and not mn.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_sleep",
                                                                      "s_endpgm", "s_setprio", "s_wakeup", "s_nop", "s_bitcmp"))):
                for s in regs_of(ops[0], "s"):
                    sgpr_valu_write.pop(s, None)
        prev_vgpr_writes = regs_of(ops[0], "v") if (is_valu and ops and mn not in ("v_readlane_b32", "v_readfirstlane_b32")
                                                     and not mn.startswith("v_cmp")) else set()
        if mn == "v_writelane_b32" and ops:
            prev_vgpr_writes = regs_of(ops[0], "v")
        prev_text = line.strip()
    return findings, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-states", type=int, default=5)
    ap.add_argument("--objdir", default=os.path.join(ROOT, "jrl-qp_amd", "csrc"))
    ap.add_argument("--list", action="store_true", help="print every finding (default: the first 40)")
    ap.add_argument("--fix", nargs=2, metavar=("IN.s", "OUT.s"), help="insert the missing wait states into a hipcc -S listing")
    ap.add_argument("objects", nargs="*")
    a = ap.parse_args()
    if a.fix:
        out = []
        _, st = lint_listing(open(a.fix[0]).read(), a.min_states, os.path.basename(a.fix[0]), out)
        with open(a.fix[1], "w") as f:
            f.write("\n".join(out))
        print(f"isa_lint --fix {os.path.basename(a.fix[0])}: {st['instructions']} instructions, {st['lane_selects']} lane selects, "
              f"{st.get('fixed', 0)} wait-state insertions")
        return 0
    objs = a.objects or sorted(glob.glob(os.path.join(a.objdir, "*.o")))
    total = []
    with tempfile.TemporaryDirectory() as wd:
        for obj in objs:
            # host-only objects (capi.o, gi_dense_host.o) carry no device code: nothing to check
            sec = subprocess.run(["objdump", "-h", obj], capture_output=True, text=True).stdout
            if ".hip_fatbin" not in sec:
                continue
            try:
                text = disassemble(obj, wd)
            except subprocess.CalledProcessError:
                print(f"isa_lint: {os.path.basename(obj)}: no gfx950 code object (skipped)")
                continue
            f, st = lint_listing(text, a.min_states, os.path.basename(obj))
            print(f"isa_lint: {os.path.basename(obj):22s} {st['functions']:3d} kernels {st['instructions']:8d} instructions, "
                  f"{st['lane_selects']:6d} lane selects ({st['valu_sgpr_selects']} by a VALU-written SGPR): {len(f)} findings")
            total += f
    for obj, func, rule, msg in (total if a.list else total[:40]):
        print(f"  {rule} {obj} {func[:90]}: {msg}")
    if total:
        print(f"isa_lint: {len(total)} findings (min wait states {a.min_states})")
        return 1
    print("isa_lint: clean")
    return 0


if __name__ == "__main__":
    sys.exit(main())
