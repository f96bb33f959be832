'''Per-kernel ISA report of a .s file from `hipcc -S --cuda-device-only`: registers, scratch (spill) instructions and where
they sit relative to the MFMA main loop, waits and barriers.  usage: isa_loop_report.py file.s [name-substring ...]'''
import re
import sys


def kernels(text):
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', text, re.S | re.M):
        yield m.group(1), m.group(2).split('\n'), m.end()


def main():
    text = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for name, body, end in kernels(text):
        if pats and not any(p in name for p in pats):
            continue
        meta = text[end:end + 6000]
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta)
        sg = re.search(r'\.amdhsa_next_free_sgpr (\d+)', meta)
        mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
        scr = [i for i, l in enumerate(body) if re.search(r'scratch_(load|store)', l)]
        vm0 = [i for i, l in enumerate(body) if re.search(r's_waitcnt.*vmcnt\(0\)', l)]
        # the hot loop: the backward branch with the most MFMAs between label and branch
        labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                a = labels[m.group(1)]
                n = sum(1 for j in mf if a <= j <= i)
                if n > best[0]:
                    best = (n, a, i)
        n, a, b = best
        print(f'{name}\n  vgpr {vg.group(1) if vg else "?"} sgpr {sg.group(1) if sg else "?"}  lines {len(body)}  mfma {len(mf)}  '
              f'hot loop lines {a}-{b} ({n} mfma)  scratch ops {len(scr)} (in loop {sum(1 for j in scr if a <= j <= b)})  '
              f'vmcnt(0) in loop {sum(1 for j in vm0 if a <= j <= b)}  barriers in loop {sum(1 for j, l in enumerate(body) if "s_barrier" in l and a <= j <= b)}')


if __name__ == '__main__':
    main()
