"""Static check of the inline-assembly units in a kernel's ISA listing: the instruction in front of an unguarded unit (one that
starts with v_readlane_b32, no s_nop) must not be a VALU write of a register the unit's v_readlanes read -- the hardware wants a
wait state there and the compiler cannot see into the unit.   usage: check_asm_units.py listing.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
bad = units = 0
for i, l in enumerate(lines):
    if "#ASMSTART" not in l:
        continue
    j = i + 1
    body = []
    while "#ASMEND" not in lines[j]:
        body.append(lines[j].strip()); j += 1
    if not body or not body[0].startswith("v_readlane_b32"):
        continue
    units += 1
    srcs = set(re.findall(r"v_readlane_b32 s\d+, (v\d+),", " ".join(body)))
    k = i - 1
    while k >= 0 and (not lines[k].strip() or lines[k].strip().startswith(";") or lines[k].strip().endswith(":")):
        k -= 1
    prev = lines[k].strip()
    m = re.match(r"(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))", prev)
    if m and not prev.startswith("v_readlane") :
        regs = {f"v{m.group(5)}"} if m.group(5) else {f"v{r}" for r in range(int(m.group(3)), int(m.group(4)) + 1)}
        if regs & srcs:
            bad += 1
            print(f"line {i}: '{prev}' feeds the unit: {body[0]}")
print(f"{units} unguarded units, {bad} behind a VALU write of their source")
sys.exit(1 if bad else 0)
