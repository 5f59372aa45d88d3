#!/usr/bin/env python3
"""Build step between the compiler and the assembler: packed fp32 instructions whose LOW result selects the HIGH register
of src1 (op_sel bit 1) are rewritten with src0 and src1 exchanged (add, mul and the product of fma commute), so that the
selection sits on src0.

Why (round 3): on the MI355X such an instruction - e.g.
    v_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]
which the SLP vectoriser emits when it pairs two scalar operations whose operands sit crosswise in a register pair -
intermittently returned a wrong low result in lanes 32-63 (about one tile in 750; one wrong row in 10 % of the launches of a
2400-node graph).  Found by editing that one instruction in the compiler's assembly of the node kernel and re-assembling:
the scalar v_add_f32, the plain packed form and the SAME selection on src0 (operands exchanged) gave 150 of 150 bit-equal
runs each; the original, the original with s_nop 7 before / after it and the original with another destination register
did not.  The compiler has no switch for this, hence this peephole; tools/check_isa.py verifies the shipped library.

usage: pkfix.py in.s out.s      (prints how many instructions it rewrote)
"""
import re
import sys

COMMUTATIVE = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_max_f32", "v_pk_min_f32"}
MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")
INSN = re.compile(r"^(\s*)(v_pk_\w+)(\s+)(.*?)(\s*(?:;.*)?)$")


def split_operands(text):
    """'v[2:3], v[4:5], s[6:7]' -> list (commas inside brackets do not split)"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def low_takes_src1_high(line):
    """(mnemonic, True) when the line is a packed fp32 instruction with op_sel bit 1 set"""
    m = INSN.match(line)
    if not m or not m.group(2).endswith("_f32"):
        return None, False
    mods = dict(MOD.findall(m.group(4)))
    sel = [int(x) for x in mods.get("op_sel", "0,0,0").split(",")]
    return m.group(2), len(sel) > 1 and sel[1] == 1


def fix_line(line):
    op, bad = low_takes_src1_high(line)
    if not bad:
        return line, False
    m = INSN.match(line)
    body = m.group(4)
    first_mod = MOD.search(body)
    operand_text, mod_text = body[:first_mod.start()].rstrip(), body[first_mod.start():]
    ops = split_operands(operand_text)
    sel = [int(x) for x in dict(MOD.findall(mod_text))["op_sel"].split(",")]
    if op not in COMMUTATIVE or sel[0] == 1 or len(ops) < 3:
        raise SystemExit(f"pkfix: cannot rewrite (not commutative, or src0 selects its high register too): {line.strip()}")
    ops[1], ops[2] = ops[2], ops[1]

    def swap01(mm):
        bits = mm.group(2).split(",")
        bits[0], bits[1] = bits[1], bits[0]
        return f"{mm.group(1)}:[{','.join(bits)}]"

    new = f"{m.group(1)}{op}{m.group(3)}{', '.join(ops)} {MOD.sub(swap01, mod_text)}{m.group(5)}"
    assert not low_takes_src1_high(new)[1]
    return new, True


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n = 0
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            line = line.rstrip("\n")
            new, changed = fix_line(line)
            n += changed
            g.write(new + "\n")
    print(f"pkfix: {src}: {n} packed fp32 instruction(s) rewritten (src0 <-> src1)")


if __name__ == "__main__":
    main()
