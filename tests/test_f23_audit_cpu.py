"""The hand-scheduled transform-domain convolution against the compiler: `__graft_entry__.audit_f23()` compiles
csrc/sg3_modconv_f23.hip for gfx950 with -save-temps (hipcc cross-compiles without a GPU) and scans the assembly with
tools/audit_f23_asm.py.  The audit itself is tested on synthetic listings: each rule must fire on a violation."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT = os.path.join(ROOT, 'tools', 'audit_f23_asm.py')

HEAD = '_ZN3sg318modconv_f23_kernelILi7EEEvNS_9F23ParamsE:\n'
TAIL = '\ts_endpgm\n'


def _audit(text, tmp_path):
    f = tmp_path / 'k.s'
    f.write_text(HEAD + text + TAIL)
    r = subprocess.run([sys.executable, AUDIT, str(f)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_audit_rules_fire_on_synthetic_violations(tmp_path):
    clean = ('\t;;#ASMSTART\n\ts_nop 4\n\tbuffer_load_dwordx2 v[10:11], v1, s[4:7], s9 offen\n\t;;#ASMEND\n'
             '\tv_add_u32_e32 v2, v3, v4\n'
             '\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v5, v10\n')
    assert _audit(clean, tmp_path)[0] == 0
    # (1) an in-flight vector destination is read before its wait
    rc, out = _audit(clean.replace('v_add_u32_e32 v2, v3, v4', 'v_mov_b32_e32 v2, v11'), tmp_path)
    assert rc == 1 and 'touches v[11]' in out
    # (2) an in-flight scalar destination (s_buffer_load) is read before an asm lgkmcnt(0)
    sc = ('\t;;#ASMSTART\n\ts_buffer_load_dwordx8 s[36:43], s[64:67], s82\n\t;;#ASMEND\n\ts_mov_b32 s1, s40\n'
          '\t;;#ASMSTART\n\ts_waitcnt vmcnt(6) lgkmcnt(0)\n\t;;#ASMEND\n')
    rc, out = _audit(sc, tmp_path)
    assert rc == 1 and 'touches s[40]' in out
    assert _audit(sc.replace('s_mov_b32 s1, s40', 's_mov_b32 s1, s44'), tmp_path)[0] == 0
    # (3) the compiler touches accumulator registers itself
    rc, out = _audit('\tv_accvgpr_read_b32 v1, a3\n', tmp_path)
    assert rc == 1 and 'accumulator-register access' in out
    # (4) scratch
    rc, out = _audit('\tscratch_store_dword off, v1, s0\n', tmp_path)
    assert rc == 1 and 'scratch access' in out
    # (5) a compiler-counted wait while hand-issued loads (here: into accumulator registers) are in flight
    cw = ('\t;;#ASMSTART\n\tbuffer_load_dwordx4 a[0:3], v54, s[60:63], s48 offen\n\t;;#ASMEND\n'
          '\tbuffer_load_dword v7, v1, s[4:7], 0 offen\n\ts_waitcnt vmcnt(1)\n')
    rc, out = _audit(cw, tmp_path)
    assert rc == 1 and 'compiler-counted' in out
    assert _audit(cw.replace('vmcnt(1)', 'vmcnt(0)'), tmp_path)[0] == 0


def test_shipped_f23_kernel_passes_the_audit():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.audit_f23()            # raises on any finding
