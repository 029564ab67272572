"""tools/isa_lint.py on small hand-written listings: each rule flags its hazard and the fix pass pads it away (CPU only).

R1  a lane select (v_readlane / v_writelane) by an SGPR that a VALU instruction wrote fewer than five wait states earlier
R2  a v_readlane of a VGPR right behind the VALU instruction that wrote it
R3  a memory instruction addressed through an SGPR that a VALU instruction wrote fewer than eight wait states earlier -- the
    hazard table asks for five, hipcc emits exactly five behind an SGPR-spill reload, and that faulted on hardware"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)


def listing(body):
    return "0000000000001000 <kernel_under_test>:\n" + "\n".join("\t" + l for l in body) + "\n"


def rules(text):
    findings, _ = lint.lint_listing(text, 5, "unit")
    return sorted(f[2] for f in findings)


def fixed(text):
    out = []
    lint.lint_listing(text, 5, "unit", out)
    return "\n".join(out)


def test_r3_five_wait_states_behind_a_spill_reload_are_flagged_and_padded():
    # the sequence rocgdb stopped on: reload of an address pair from spill lanes, s_nop 4, the load through it
    bad = listing(["v_readlane_b32 s0, v205, 30", "v_lshlrev_b32_e32 v123, 3, v0", "v_readlane_b32 s1, v205, 31", "s_nop 4",
                   "global_load_dwordx2 v[132:133], v123, s[0:1]", "s_endpgm"])
    assert rules(bad) == ["R3"]
    out = fixed(bad)
    assert "isa_lint R3" in out and rules(out.replace("; isa_lint R3", "")) == []
    # eight wait states are enough; an SALU write of the pair in between supersedes the VALU write
    ok8 = listing(["v_readlane_b32 s1, v205, 31", "s_nop 7", "global_load_dwordx2 v[2:3], v1, s[0:1]"])  # (s_nop N = N + 1 wait states)
    assert rules(ok8) == []
    salu = listing(["v_readlane_b32 s1, v205, 31", "s_mov_b64 s[0:1], s[4:5]", "global_load_dwordx2 v[2:3], v1, s[0:1]"])
    assert rules(salu) == []
    # scalar loads and stores through a VALU-written base count as well
    smem = listing(["v_readfirstlane_b32 s4, v7", "v_readfirstlane_b32 s5, v8", "s_load_dwordx2 s[8:9], s[4:5], 0x0"])
    assert rules(smem) == ["R3"]
    st = listing(["v_readlane_b32 s3, v9, 2", "s_nop 1", "global_store_dwordx2 v1, v[4:5], s[2:3]"])
    assert rules(st) == ["R3"]


def test_r1_lane_select_and_r2_back_to_back_readlane():
    r1 = listing(["v_readfirstlane_b32 s6, v3", "s_nop 3", "v_readlane_b32 s8, v10, s6"])  # four wait states: what hipcc emits
    assert rules(r1) == ["R1"]
    assert rules(fixed(r1).replace("; isa_lint R1", "")) == []
    ok = listing(["v_readfirstlane_b32 s6, v3", "s_nop 4", "v_readlane_b32 s8, v10, s6"])
    assert rules(ok) == []
    r2 = listing(["v_add_f64 v[10:11], v[2:3], v[4:5]", "v_readlane_b32 s8, v10, 3"])
    assert rules(r2) == ["R2"]
    assert rules(fixed(r2).replace("; isa_lint R2", "")) == []


def test_scopes_do_not_leak_between_functions():
    two = ("0000000000001000 <first>:\n\tv_readlane_b32 s1, v205, 31\n\ts_endpgm\n"
           "0000000000002000 <second>:\n\tglobal_load_dwordx2 v[2:3], v1, s[0:1]\n")
    assert rules(two) == []


def test_wait_states_are_counted_along_branch_edges():
    """A consumer at a branch target is not credited with the instructions the taken branch skipped (round 3's advisor): the producer
    sits right in front of an s_cbranch, eight fall-through instructions follow, the consumer is the first instruction behind the label."""
    skipped = ["v_mov_b32_e32 v20, 0"] * 8
    # hipcc -S form of a label
    s_form = "kernel_under_test:\n" + "\n".join("\t" + l for l in ["v_readfirstlane_b32 s6, v3", "s_cbranch_scc1 .LBB0_2"] + skipped) + \
             "\n.LBB0_2:\n\tv_readlane_b32 s8, v10, s6\n\ts_endpgm\n"
    out = []
    lint.lint_listing(s_form, 5, "unit", out)
    padded = "\n".join(out)
    assert "isa_lint R1" in padded  # one wait state (the branch) on the taken edge: four more inserted in front of the consumer
    assert "s_nop 3 ; isa_lint R1" in padded
    # llvm-objdump --symbolize-operands form
    d_form = listing(["v_readfirstlane_b32 s6, v3", "s_cbranch_scc1 L2"] + skipped) + "0000000000001100 <L2>:\n\tv_readlane_b32 s8, v10, s6\n\ts_endpgm\n"
    assert rules(d_form) == ["R1"]
    # a memory instruction at a loop head whose base pair is reloaded at the loop's end (backward branch): R3 along the back edge
    loop = "kernel_under_test:\n.LBB0_1:\n\tglobal_load_dwordx2 v[2:3], v1, s[0:1]\n" + "\n".join("\t" + l for l in skipped) + \
           "\n\tv_readlane_b32 s0, v205, 30\n\tv_readlane_b32 s1, v205, 31\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n"
    out = []
    lint.lint_listing(loop, 5, "unit", out)
    assert "isa_lint R3" in "\n".join(out)
    # enough wait states in front of the branch: clean either way
    ok = listing(["v_readfirstlane_b32 s6, v3", "s_nop 4", "s_cbranch_scc1 L2"] + skipped) + "0000000000001100 <L2>:\n\tv_readlane_b32 s8, v10, s6\n\ts_endpgm\n"
    assert rules(ok) == []


def test_r2_across_a_label_on_the_fall_through_path():
    """A loop head or branch target between the VALU write and the v_readlane does not add a wait state on the fall-through path
    (llvm-objdump --symbolize-operands prints every such target as a label)."""
    r2 = listing(["v_add_f64 v[10:11], v[2:3], v[4:5]", "<L7>:", "v_readlane_b32 s8, v10, 3", "s_cbranch_scc1 L7", "s_endpgm"])
    assert rules(r2) == ["R2"]
    assert rules(fixed(r2).replace("; isa_lint R2", "")) == []
    ok = listing(["v_add_f64 v[10:11], v[2:3], v[4:5]", "<L7>:", "s_nop 0", "v_readlane_b32 s8, v10, 3", "s_cbranch_scc1 L7", "s_endpgm"])
    assert rules(ok) == []


def test_r1_age_carried_over_two_branch_edges():
    """write -> branch A -> label A: branch B -> label B: consumer.  The SGPR's age is imported at label A and must still be counted at B."""
    chain = listing(["v_readfirstlane_b32 s6, v3", "s_branch L1", "s_nop 7", "<L1>:", "s_branch L2", "s_nop 7", "<L2>:", "v_readlane_b32 s8, v10, s6", "s_endpgm"])
    assert rules(chain) == ["R1"]  # three wait states: the two branches and nothing else
    assert rules(fixed(chain).replace("; isa_lint R1", "")) == []
