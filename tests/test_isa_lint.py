"""tools/isa_lint.py: the two instruction forms that return wrong results on MI355X (profiles/r06_hazards.txt) are recognised in
ISA text, and NO kernel of the built library contains them.  CPU only: the objects are cross-compiled here."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def _rules(text):
  return [(r, t) for r, _, t in isa_lint.lint_lines(text.split("\n"))]


def test_packed_f32_op_sel_forms():
  bad = ["v_pk_mul_f32 v[154:155], v[206:207], v[166:167] op_sel:[0,1]",
         "v_pk_add_f32 v[2:3], v[2:3], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]",
         "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]",
         "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] neg_lo:[0,0,1]"]
  good = ["v_pk_mul_f32 v[0:1], v[2:3], v[4:5]",
          "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]",
          "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]",
          "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]",
          "v_pk_mul_f32 v[180:181], v[220:221], s[36:37] op_sel_hi:[1,0]",
          "v_pk_fma_f32 v[154:155], v[208:209], s[36:37], v[154:155] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]",
          "v_pk_add_u16 v0, v1, v2 op_sel:[0,1]"]   # (16-bit packed integer: not the measured family)
  for l in bad:
    assert _rules("\t" + l) == [("R1", l)], l
  for l in good:
    assert _rules("\t" + l) == [], l


def test_wide_buffer_store_with_register_soffset():
  store = "\tbuffer_store_dwordx4 v[14:17], v159, s[72:75], s88 offen"
  # the pair head_fused_kernel<zinbd> had (round 4's "garbage in the table")
  assert [r for r, _ in _rules(store + "\n\tv_pk_add_f32 v[14:15], v[18:19], v[50:51] neg_lo:[0,1] neg_hi:[0,1]")] == ["R2"]
  assert [r for r, _ in _rules(store + "\n\tv_mov_b32_e32 v17, v2")] == ["R2"]
  # any instruction in between is the one wait state the register-soffset form needs
  assert _rules(store + "\n\ts_waitcnt vmcnt(33)\n\tv_cvt_pk_bf16_f32 v14, v66, v67") == []
  assert _rules(store + "\n\ts_nop 0\n\tv_mov_b32_e32 v14, v2") == []
  # other registers, a compare into scalar registers, and the immediate-soffset form (the compiler pads that one itself)
  assert _rules(store + "\n\tv_mov_b32_e32 v18, v2") == []
  assert _rules(store + "\n\tv_cmp_gt_f32_e32 vcc, v14, v15") == []
  assert _rules("\tbuffer_store_dwordx4 v[14:17], v159, s[72:75], 0 offen\n\tv_mov_b32_e32 v14, v2") == []
  assert _rules("\tbuffer_store_dwordx2 v[14:15], v159, s[72:75], s88 offen\n\tv_mov_b32_e32 v14, v2") == []
  # objdump spelling: addresses, encodings and labels around the instructions
  dis = ("0000000000001200 <_ZN3smx5thingEv>:\n\tbuffer_store_dwordx4 v[18:21], v164, s[72:75], s94 offen // 000000001234: E07C1000 5E120EA4\n"
         "\tv_cvt_pk_bf16_f32 v18, v16, v17                            // 00000000123C: D2680012 00022310")
  assert isa_lint.lint_lines(dis.split("\n"))[0][:2] == ("R2", "_ZN3smx5thingEv")


def test_the_built_library_has_neither_form():
  objs = sorted(glob.glob(os.path.join(ROOT, "sisua_amd", "csrc", "*.o")))
  from sisua_amd import build as b
  if len(objs) < len(b.SOURCES):
    pytest.skip("objects not built here (the .so came with the snapshot); build() lints them where it compiles")
  found = isa_lint.lint_files(objs)
  assert found == [], found[:5]


def test_the_build_passes_the_flag_that_keeps_the_vectoriser_from_forming_packed_f32():
  from sisua_amd import build as b
  assert "-fno-slp-vectorize" in b.FLAGS
