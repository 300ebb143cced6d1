"""Instruction mix of the MFMA main loop of every kernel in a gfx950 .s file (hipcc -S --cuda-device-only).

usage: python scripts/isa_loop_mix.py file.s [name-substring]
Finds, per kernel, the backward-branch region that contains the most MFMAs in the smallest span and
prints its opcode histogram: the non-MFMA instructions per MFMA is what bounds matrix-pipe utilisation.
"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    starts.append((len(lines), 'end'))
    for (s, name), (e, _) in zip(starts, starts[1:]):
        if want not in name:
            continue
        body = lines[s:e]
        labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\w+:', l)}
        mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
        if not mf:
            continue
        regions = []
        for i, l in enumerate(body):
            t = l.strip()
            if t.startswith('s_cbranch') or t.startswith('s_branch'):
                tgt = t.split()[-1]
                if tgt in labels and labels[tgt] < i:
                    lo = labels[tgt]
                    n = sum(1 for m in mf if lo <= m <= i)
                    if n:
                        regions.append((-n, i - lo, lo, i))
        if not regions:
            continue
        regions.sort()
        _, _, lo, hi = regions[0]
        c = collections.Counter()
        for l in body[lo:hi + 1]:
            t = l.strip()
            if not l.startswith('\t') or t.startswith('.') or t.startswith(';'):
                continue
            c[t.split()[0]] += 1
        tot = sum(c.values())
        nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        salu = sum(v for k, v in c.items() if k.startswith('s_'))
        ds = sum(v for k, v in c.items() if k.startswith('ds_'))
        vm = sum(v for k, v in c.items() if k.startswith('global_') or k.startswith('buffer_'))
        print(f"{name}\n  loop lines {lo}-{hi}: {tot} instrs, mfma {nm}, valu {valu}, salu {salu}, lds {ds}, "
              f"vmem {vm}  -> non-MFMA/MFMA {(tot - nm) / nm:.2f}, VALU/MFMA {valu / nm:.2f}")
        print("   " + ", ".join(f"{k}:{v}" for k, v in c.most_common(24)))


if __name__ == '__main__':
    main()
