'''The tile / split-K choice of fd_gemm_f16's rule for every launch of the bench workload (tests/golden/gemm_launches_sd15_b8.json, dumped on an
MI355X by tools/dump_gemm_descs.py), through fd_gemm_plan -- host logic only, runs without a device.
    python tests/golden/make_gemm_rule_table.py            print the table and compare with the committed one
    python tests/golden/make_gemm_rule_table.py --update   rewrite tests/golden/gemm_rule_table.json (after an INTENDED rule change)'''
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
LAUNCHES = os.path.join(HERE, 'gemm_launches_sd15_b8.json')
TABLE = os.path.join(HERE, 'gemm_rule_table.json')


def plan_all():
    '''-> [(description, M, N, K, K2, launches, tile, split_k, rc)] in the order of the dump'''
    from flexdiffuse_amd import hip, ops
    lib = hip.lib()
    rows = []
    for r in json.load(open(LAUNCHES))['launches']:
        d = ops.fd_gemm_desc()
        for name, typ in ops.fd_gemm_desc._fields_:
            v = r.get(name, None if typ is ctypes.c_void_p else 0)     # (fields newer than the dump: off)
            if typ is ctypes.c_void_p:
                # a fake, never dereferenced address with the recorded alignment
                setattr(d, name, None if v is None else 0x10000000 + 0x1000000 * (len(rows) % 7) + int(v))
            else:
                setattr(d, name, v)
        tile, split = ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile), ctypes.byref(split))
        rows.append((r['what'], r['M'], r['N'], r['K'], r['K2'], r['launches'], tile.value, split.value, rc))
    return rows


def main():
    rows = plan_all()
    table = [{'what': w, 'M': M, 'N': N, 'K': K, 'K2': K2, 'launches': n, 'tile': t, 'split_k': s} for w, M, N, K, K2, n, t, s, rc in rows]
    assert all(rc == 0 for *_, rc in rows)
    if '--update' in sys.argv:
        json.dump(table, open(TABLE, 'w'), indent=0)
        print(f'wrote {TABLE}: {len(table)} launches')
        return
    old = json.load(open(TABLE)) if os.path.exists(TABLE) else []
    for i, t in enumerate(table):
        o = old[i] if i < len(old) else {}
        flag = '' if (o.get('tile'), o.get('split_k')) == (t['tile'], t['split_k']) else f"   <<< committed: tile {o.get('tile')} split {o.get('split_k')}"
        print(f"M={t['M']:7d} N={t['N']:5d} K={t['K']:5d} +{t['K2']:4d} {t['what']:28s} x{t['launches']:2d}: tile {t['tile']:3d} split {t['split_k']}{flag}")


if __name__ == '__main__':
    main()
