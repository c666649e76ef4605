"""CPU: the two places where kernels carry hand-written instructions (inline asm) rely on wait
states that hipcc's hazard recogniser does not see inside an asm statement.  This test pins them
on the ISA the installed hipcc actually emits (cross-compiled to gfx950 assembly, no GPU needed):

* csrc/match_knn2v2.hip `med3_after`: an asm v_med3_i32 reads a raw MFMA accumulator; it is only
  safe because a COMPILER-generated VALU instruction (which gets its MFMA -> VALU wait states from
  the compiler) has read the same register first.
* csrc/match_knn2sym.hip `half_wave_min16<true>`: asm v_min_i32_dpp reads registers through DPP;
  a DPP read needs two wait states after a VALU write of that register, provided by the `s_nop 1`
  that opens every asm block and by the order of the instructions inside it.

A compiler upgrade that schedules differently fails here instead of producing wrong matches."""
import os
import re
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'imageanalysis_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def _regs(text):
    out = []
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _functions(path):
    """{kernel name: [(mnemonic, operand text, in_asm)]} of a .s file"""
    funcs, cur, in_asm = {}, None, False
    for line in open(path):
        line = line.rstrip()
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            cur = None
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        s = s.split(';')[0].strip()
        parts = s.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else '', in_asm))
    return funcs


def _assembly(src):
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    out = os.path.join(tempfile.mkdtemp(prefix='iamx_isa_'), 'k.s')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                           '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, os.path.join(CSRC, src),
                           '-o', out], stderr=subprocess.DEVNULL)
    return _functions(out)


def test_asm_med3_never_is_the_first_reader_of_an_mfma_result():
    funcs = _assembly('match_knn2v2.hip')
    checked = 0
    for name, ins in funcs.items():
        if 'knn2v2_kernel' not in name:
            continue
        pending = set()                      # VGPRs whose last writer is an MFMA nobody (safe) has read
        for _round in range(2):              # second pass: state carried around the loop back edges
            for mn, ops, in_asm in ins:
                regs = _regs(ops)
                if mn.startswith('v_mfma'):
                    dst = _regs(ops.split(',')[0])
                    pending.difference_update(regs)          # (MFMA -> MFMA accumulate chains are interlocked)
                    pending.update(dst)
                elif in_asm:
                    if mn == 'v_med3_i32':
                        checked += 1
                    bad = pending.intersection(regs[1:] if regs else [])
                    assert not bad, (name, mn, ops, sorted(bad))
                else:
                    pending.difference_update(regs)
    assert checked > 100                     # the kernels do contain the asm form


def test_asm_dpp_reads_have_their_wait_states():
    funcs = _assembly('match_knn2sym.hip')
    blocks = 0
    for name, ins in funcs.items():
        if 'knn2sym_kernel' not in name:
            continue
        k = 0
        while k < len(ins):
            if not ins[k][2]:
                k += 1
                continue
            block = []
            while k < len(ins) and ins[k][2]:
                block.append(ins[k])
                k += 1
            if not any(mn.endswith('_dpp') for mn, _o, _a in block):
                continue
            blocks += 1
            # two wait states in front of the block: whatever the compiler put before it
            assert block[0][0] == 's_nop' and int(block[0][1]) >= 1, (name, block[0])
            written = []                     # destination of the previous instructions of the block
            for mn, ops, _a in block[1:]:
                assert mn == 'v_min_i32_dpp', (name, mn)
                regs = _regs(ops)
                dst, dpp_src = regs[0], regs[1]
                # a register read through DPP must not have been written by the two
                # instructions in front of this one (inside the block: no s_nop between them)
                assert dpp_src not in written[-2:], (name, ops, written[-2:])
                written.append(dst)
    assert blocks >= 6                       # two levels per tile row reduction, several kernel forms
