'''Per-kernel VGPR / SGPR / scratch / LDS from a hipcc -S listing's metadata (the .amdhsa YAML):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/x.s file.hip
    python tools/kernel_resources.py /tmp/x.s [name-filter]
Scratch > 0 means register spills inside the kernel.'''
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for blk in txt.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if flt in name:
        print(f"{name:90s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} "
              f"spill_v {g('vgpr_spill_count'):>4s} spill_s {g('sgpr_spill_count'):>4s}")
