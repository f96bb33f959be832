#!/bin/bash
# Kernel sequence of ONE denoising step inside the headline request (between two k_cfg_ddim launches), from a rocprofv3 kernel trace of the bench command:
# names the launches that are not the library's (torch copies / fills).   bash tools/trace_forward_kernels.sh   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fd_trace && timeout 600 rocprofv3 --kernel-trace -d /tmp/fd_trace -o t --output-format csv -- python3 $R/bench.py --no-parity --no-cpu-baseline --steps 1 --warmup 1 > /tmp/fd_trace.json 2> /tmp/fd_trace.err
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/fd_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_cfg_ddim')]
a, b = idx[-3], idx[-2]
seq = names[a + 1:b + 1]
print(f'{len(seq)} launches between two k_cfg_ddim launches of the last timed pass')
c = collections.Counter(n.split('(')[0][:90] for n in seq if not n.startswith(('void k_', 'k_', '_Z')) or 'rocclr' in n or 'at::' in n)
for k, v in c.items(): print(v, k)
for i, n in enumerate(seq):
    if 'rocclr' in n or 'at::' in n:
        print(i, n[:100], '| after:', seq[i - 1][:60] if i else '-')
PY
