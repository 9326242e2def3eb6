"""Sanity check of a hipcc -save-temps .s file: registers written by (hand-placed, inline-asm) buffer_load_dword[x2|x3|x4] instructions must not
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


VMEM = ('buffer_load', 'buffer_store', 'buffer_atomic', 'global_load', 'global_store', 'global_atomic', 'flat_load', 'flat_store', 'scratch_')


def main(path, flt=''):
    """Text-order model of the vector-memory queue: every VMEM instruction enters it (hand-placed buffer_load_dwordx4: with its destination
    registers), `s_waitcnt vmcnt(N)` retires all but the youngest N (vmcnt retires in order).  A destination register still in the queue must
    not be read or written.  (Round 5: the first version emptied the queue at EVERY vmcnt wait and so missed the one real bug of this kind --
    the dummy request of a tile's last k-step, still in flight behind a vmcnt(2 NT) when the epilogue began.)"""
    queue, bad, kern = [], 0, ''
    for ln, line in enumerate(open(path), 1):
        code = line.split(';')[0].strip()
        if code.endswith(':') and not code.startswith('.'):
            kern, queue = code[:-1], []
        if flt not in kern or not code or code.startswith('.'):
            continue
        if code.startswith('s_waitcnt') and 'vmcnt' in code:
            n = int(re.search(r'vmcnt\((\d+)\)', code).group(1))
            queue = queue[len(queue) - n:] if n and n <= len(queue) else ([] if not n else queue)
            continue
        if code.startswith('s_endpgm'):
            queue = []
            continue
        pending = set().union(*queue) if queue else set()
        if re.match(r'buffer_load_dword(x[234])?\s', code) and ' lds' not in code:
            ops = code.split(None, 1)[1].split(',')
            if regs(','.join(ops[1:])) & pending:
                print('%s:%d address uses pending register: %s' % (kern[:50], ln, code)); bad += 1
            if regs(ops[0]) & pending:
                print('%s:%d second request into in-flight register: %s' % (kern[-30:], ln, code)); bad += 1
            queue.append(regs(ops[0]))
            continue
        hit = regs(code) & pending
        if hit:
            print('%s:%d touches in-flight v%s: %s' % (kern[-30:], ln, sorted(hit)[:4], code)); bad += 1
        if code.startswith(VMEM):
            queue.append(set())
    print('check_asm_loads: %d problem(s)' % bad)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(*sys.argv[1:3]) else 0)
