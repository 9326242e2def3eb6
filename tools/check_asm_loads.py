"""Sanity check of a hipcc -save-temps .s file: registers written by (hand-placed, inline-asm) buffer_load_dwordx4 instructions must not
be touched before the next s_waitcnt vmcnt in program text order -- the compiler does not know these loads are asynchronous.
usage: python tools/check_asm_loads.py <file.s> [kernel substring]"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def main(path, flt=''):
    pending, bad, kern = set(), 0, ''
    for ln, line in enumerate(open(path), 1):
        code = line.split(';')[0].strip()
        if code.endswith(':') and not code.startswith('.'):
            kern, pending = code[:-1], set()
        if flt not in kern or not code or code.startswith('.'):
            continue
        if code.startswith('s_waitcnt') and 'vmcnt' in code:
            pending = set()
            continue
        if code.startswith('buffer_load_dwordx4') and ' lds' not in code:
            ops = code.split(None, 1)[1].split(',')
            pending |= regs(ops[0])
            if regs(','.join(ops[1:])) & pending:
                print('%s:%d address uses pending register: %s' % (kern[:50], ln, code)); bad += 1
            continue
        hit = regs(code) & pending
        if hit:
            print('%s:%d touches in-flight v%s: %s' % (kern[-30:], ln, sorted(hit)[:4], code)); bad += 1
    print('check_asm_loads: %d problem(s)' % bad)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(*sys.argv[1:3]) else 0)
