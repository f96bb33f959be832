'''Dump every unique fd_gemm_f16 launch of the bench workload -- one full-size SD1.5 CFG forward at batch 8 (64x64 latents, CFG batch 16) and one
VAE decode of 8 latents -- as data: the scalar fields of fd_gemm_desc, which pointers were set (and their alignment), launches per pass.
Needs an MI355X (the forward has to run to see its launches).  The table of the rule's choices is then made and checked WITHOUT a device:
tests/golden/make_gemm_rule_table.py, tests/test_gemm_rule_table.py.
    python tools/dump_gemm_descs.py > tests/golden/gemm_launches_sd15_b8.json'''
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_recorder
from flexdiffuse_amd import ops
rec, keep = gemm_recorder.record('sd15', 64, 8, vae=True)
out = []
for key, (d, cnt) in rec.items():
    row = {'launches': cnt, 'what': gemm_recorder.describe(key)}
    for name, typ in ops.fd_gemm_desc._fields_:
        v = getattr(d, name)
        if typ is ctypes.c_void_p:
            row[name] = None if not v else int(v) % 16       # set? + alignment
        elif name in ('tile', 'split_k'):
            row[name] = 0
        else:
            row[name] = v
    out.append(row)
json.dump({'workload': 'SD1.5 UNet CFG forward, batch 8 (CFG batch 16), 64x64 latents + VAE decode of 8 latents', 'launches': out}, sys.stdout, indent=0)
