'''The C-ABI library loads on a machine without a GPU and exports every symbol declared in
include/flexdiffuse_hip.h; the ctypes table binds exactly that set.  No compute calls.'''
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_in_header():
    text = open(os.path.join(ROOT, 'include', 'flexdiffuse_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fd_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    from flexdiffuse_amd import hip
    assert os.path.exists(hip.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = declared_in_header()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in the header but not exported'


def test_ctypes_table_matches_header():
    from flexdiffuse_amd import hip
    assert sorted(hip.declared_symbols()) == declared_in_header()
    assert hip.lib().fd_abi_version() == hip.ABI_VERSION == 12


def test_struct_layouts_match_header():
    '''Field order of the ctypes descriptors mirrors the C structs.'''
    from flexdiffuse_amd import hip, ops
    text = open(os.path.join(ROOT, 'include', 'flexdiffuse_hip.h')).read()

    def fields(struct):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (struct, struct), text, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        out = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith('const') else decl.split(None, 2)[2]
            out += [n.strip().lstrip('*') for n in names.split(',')]
        return out
    assert fields('fd_gemm_desc') == [f[0] for f in ops.fd_gemm_desc._fields_]
    assert fields('fd_attention_desc') == [f[0] for f in ops.fd_attention_desc._fields_]
    assert fields('fd_tween_params') == [f[0] for f in hip.fd_tween_params._fields_]


def test_argument_errors_without_gpu():
    '''Argument validation happens before any launch, so it can be exercised here.'''
    from flexdiffuse_amd import hip
    with pytest.raises(ValueError):
        hip.call('fd_guidance_map', None, None, None, None, None, 1, 0, 10, 77, 48, 1, 1, None)
    assert b'multiple of 32' in hip.lib().fd_last_error()


def test_launch_plan_lifecycle_without_gpu():
    '''fd_plan_* host logic (no launches): a plan records only between begin / end on the calling
    thread, refuses a second recorder and a replay while recording, and an empty plan replays.'''
    from flexdiffuse_amd import hip
    plan = hip.Plan()
    assert len(plan) == 0
    null = ctypes.c_void_p(0)
    plan.replay(null)                               # empty plan: nothing to launch
    with plan.record():
        with pytest.raises(ValueError, match='already recording'):
            with hip.Plan().record():
                pass
        with pytest.raises(ValueError, match='while this thread records'):
            plan.replay(null)
        # a call that fails its argument check is still recorded by value: replay reports it too
        with pytest.raises(ValueError):
            hip.call('fd_guidance_map', None, None, None, None, None, 1, 0, 10, 77, 48, 1, 1, None)
    assert len(plan) == 1
    with pytest.raises(ValueError, match='multiple of 32'):
        plan.replay(null)
    with plan.record():                             # begin clears the plan
        pass
    assert len(plan) == 0


def test_no_cpu_fallback():
    import torch
    from flexdiffuse_amd import guidance
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        guidance.Tweener().tween(torch.zeros(1, 77, 64), torch.zeros(1, 10, 64))
