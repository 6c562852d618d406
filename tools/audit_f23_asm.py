"""Audit of the hand-issued loads of modconv_f23_kernel in the compiler's output (hipcc -save-temps .s):
between an inline-asm buffer_load and the inline-asm `s_waitcnt vmcnt` that follows it in the text, no instruction may mention
the load's destination registers (vector registers of buffer_load, scalar registers of s_buffer_load until an asm lgkmcnt(0)) (hipcc treats an asm load's destination as written when the load is ISSUED, so a copy, spill or
read placed there moves stale data: cdna_hip_programming.md section 5.7 item 1).  Also reports scratch use, compiler-generated accumulator-register
accesses, and compiler-COUNTED vmcnt waits (N > 0) issued while hand-issued loads are still in flight (the compiler's count does not
include them, so such a wait can release a consumer early).
A linear scan: the pending set is dropped at an unconditional branch (the rotated tile loop places the tile's end in front of its
head; the requests issued before the loop are waited for at the head, not in the text that follows them).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Istylegan3-editing_amd/csrc -c stylegan3-editing_amd/csrc/sg3_modconv_f23.hip -save-temps -o /tmp/x.o
    python tools/audit_f23_asm.py sg3_modconv_f23-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

text = open(sys.argv[1]).read().split('\n')
kern = None
bad = 0
pending = {}          # vector register number -> line of the load
spending = {}         # scalar register number -> line of the scalar-cache load (s_buffer_load into SGPRs, round 4)
apending = 0          # line of the last hand-issued load into accumulator registers that no asm vmcnt wait has covered yet
in_asm = False
for ln, line in enumerate(text, 1):
    m = re.match(r'^(_ZN3sg318modconv_f23_kernel\w+):', line)
    if m:
        kern, pending, spending, apending = m.group(1), {}, {}, 0
    if kern is None:
        continue
    if 's_endpgm' in line:
        kern = None
        continue
    if '#ASMSTART' in line:
        in_asm = True
        continue
    if '#ASMEND' in line:
        in_asm = False
        continue
    code = line.split(';')[0]
    if in_asm and 'buffer_load' in code and 's_buffer_load' not in code:
        d = re.search(r'buffer_load_\w+\s+v\[(\d+):(\d+)\]|buffer_load_\w+\s+v(\d+)', code)
        if d is None:
            apending = ln                                  # destination in the accumulator file: not compiler-visible, but IN FLIGHT
            continue
        lo, hi = (int(d.group(1)), int(d.group(2))) if d.group(1) else (int(d.group(3)), int(d.group(3)))
        for r in range(lo, hi + 1):
            pending[r] = ln
        continue
    if in_asm and 's_buffer_load' in code:
        d = re.search(r's_buffer_load_\w+\s+s\[(\d+):(\d+)\]|s_buffer_load_\w+\s+s(\d+)', code)
        lo, hi = (int(d.group(1)), int(d.group(2))) if d.group(1) else (int(d.group(3)), int(d.group(3)))
        for r in range(lo, hi + 1):
            spending[r] = ln
        continue
    if in_asm and 's_waitcnt' in code:
        if 'vmcnt' in code:
            pending, apending = {}, 0
        if 'lgkmcnt(0)' in code:
            spending = {}
        continue
    if not in_asm and re.match(r'\s*s_branch\b', code):
        pending, spending, apending = {}, {}, 0            # the text behind an unconditional branch is entered from elsewhere (rotated tile loop)
        continue
    if not in_asm and (pending or apending):
        # the compiler counts only the loads IT issued: a counted wait of its own (vmcnt(N), N > 0) while hand-issued loads are in
        # flight behind it lets N of ITS loads... in fact N loads of the whole queue stay out: its consumer may read a register that
        # has not landed.  (vmcnt(0) over-waits, which is only slow.)
        w = re.search(r's_waitcnt.*vmcnt\((\d+)\)', code)
        if w and int(w.group(1)) > 0:
            print(f'{kern}: line {ln}: compiler-counted {code.strip()} while hand-issued loads (line {apending or min(pending.values())}) are in flight'); bad += 1
    if not in_asm and 'v_accvgpr' in code:
        print(f'{kern}: compiler-generated accumulator-register access at line {ln}: {code.strip()}'); bad += 1
    if 'scratch_' in code:
        print(f'{kern}: scratch access at line {ln}: {code.strip()}'); bad += 1
    if not code.strip() or code.strip().startswith('.'):
        continue
    if spending:
        sregs = set()
        for a, b in re.findall(r'\bs\[(\d+):(\d+)\]', code):
            sregs.update(range(int(a), int(b) + 1))
        sregs.update(int(r) for r in re.findall(r'\bs(\d+)\b', code))
        hit = sorted(r for r in sregs if r in spending)
        if hit:
            print(f'{kern}: line {ln} touches s{hit} loaded at line {spending[hit[0]]} before its wait: {code.strip()}'); bad += 1
    if not pending:
        continue
    regs = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', code):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r'\bv(\d+)\b', code))
    hit = sorted(r for r in regs if r in pending)
    if hit:
        print(f'{kern}: line {ln} touches v{hit} loaded at line {pending[hit[0]]} before its wait: {code.strip()}'); bad += 1
print('audit:', 'CLEAN' if bad == 0 else f'{bad} finding(s)')
sys.exit(1 if bad else 0)
