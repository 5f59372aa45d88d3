"""The build's peephole for an unsound packed fp32 instruction form (amber_amd/csrc/pkfix.py) and the static check of the
shipped library (tools/check_isa.py).  CPU only: the library is disassembled, nothing runs."""
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


pkfix = _load(os.path.join(REPO, "amber_amd", "csrc", "pkfix.py"), "pkfix")
check_isa = _load(os.path.join(REPO, "tools", "check_isa.py"), "check_isa")


@pytest.mark.parametrize("before, after", [
    # the instruction the node kernel failed on and its sound form
    ("\tv_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]",
     "\tv_pk_add_f32 v[2:3], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]"),
    # fma: the product's factors are exchanged, the addend and its bits stay
    ("\tv_pk_fma_f32 v[6:7], v[90:91], v[0:1], v[72:73] op_sel:[0,1,1]",
     "\tv_pk_fma_f32 v[6:7], v[0:1], v[90:91], v[72:73] op_sel:[1,0,1]"),
    ("\tv_pk_fma_f32 v[6:7], v[90:91], s[0:1], v[72:73] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0] ; note",
     "\tv_pk_fma_f32 v[6:7], s[0:1], v[90:91], v[72:73] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0] ; note"),
    ("\tv_pk_mul_f32 v[8:9], v[10:11], v[12:13] op_sel:[0,1]", "\tv_pk_mul_f32 v[8:9], v[12:13], v[10:11] op_sel:[1,0]"),
])
def test_rewrite_moves_the_selection_to_src0(before, after):
    new, changed = pkfix.fix_line(before)
    assert changed and new == after
    assert pkfix.fix_line(new) == (new, False)  # idempotent


@pytest.mark.parametrize("line", [
    "\tv_pk_add_f32 v[2:3], v[4:5], v[2:3]",
    "\tv_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]",   # src0: sound
    "\tv_pk_fma_f32 v[2:3], v[4:5], v[2:3], v[6:7] op_sel:[1,0,1]",         # src0 and the addend: sound
    "\tv_pk_mul_f32 v[2:3], v[4:5], v[2:3] op_sel_hi:[1,0]",                # a broadcast of src1's low register
    "\tv_pk_add_f16 v2, v4, v3 op_sel:[0,1]",                               # not fp32
    "\tv_add_f32_e32 v2, v4, v3",
    "\ts_waitcnt vmcnt(0)",
])
def test_rewrite_leaves_everything_else(line):
    assert pkfix.fix_line(line) == (line, False)


def test_rewrite_refuses_what_it_cannot_fix():
    with pytest.raises(SystemExit):
        pkfix.fix_line("\tv_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[1,1]")  # both low operands from high registers


def test_scan_flags_the_form():
    dis = "0000000000001000 <k>:\n\tv_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0] // 0001\n\tv_pk_mul_f32 v[2:3], v[4:5], v[2:3] op_sel:[1,0]\n"
    assert list(check_isa.scan(dis)) == ["k"] and len(check_isa.scan(dis)["k"]) == 1


def test_shipped_library_holds_no_unsound_packed_instruction():
    lib = os.path.join(REPO, "amber_amd", "csrc", "libamber_mpn.so")
    assert os.path.exists(lib), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    kernels, packed, bad = check_isa.check([lib])
    assert kernels > 100 and packed > 10000  # the disassembly really covered the kernels
    assert not bad, {k: v[:2] for k, v in bad.items()}
