'''ctypes binding of libflexdiffuse_hip.so (include/flexdiffuse_hip.h).

There is NO fallback: if the shared object is missing or a tensor is not on a HIP
device, calls raise.  PyTorch is used only for device memory and streams.
'''
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FD_LIB_PATH') or os.path.join(_HERE, 'libflexdiffuse_hip.so')

FD_OK, FD_EINVAL, FD_ESHAPE, FD_EHIP = 0, -1, -2, -3


class FDError(RuntimeError):
    '''HIP runtime failure reported by libflexdiffuse_hip.so (FD_EHIP).'''


class fd_tween_params(ctypes.Structure):
    _fields_ = [('threshold_floor', c_double), ('threshold_mult', c_double),
                ('clustered', c_double), ('max_guidance', c_double),
                ('header_max', c_double), ('order', c_int32), ('reuse', c_int32)]


P = c_void_p
_SIGNATURES = {
    'fd_abi_version': (c_int, []),
    'fd_last_error': (c_char_p, []),
    'fd_device_info': (c_int, [c_int, P, P, P, P, c_int]),
    'fd_prof_enable': (c_int, [c_int]),
    'fd_prof_set_stride': (c_int, [c_int]),
    'fd_prof_collect': (c_int, [c_int, P, P, P]),
    'fd_prof_collect2': (c_int, [c_int, P, P, P, P]),
    'fd_prof_calibrate': (c_int, [c_int, P, P]),
    'fd_prof_drain': (c_int, [P, P, P, P, P, c_int64, P]),
    'fd_guidance_workspace_floats': (c_int64, [c_int, c_int, c_int]),
    'fd_guidance_map': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'fd_guidance_tween': (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int,
                                  ctypes.POINTER(fd_tween_params), P]),
    'fd_guidance_concept_override': (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    'fd_guidance_header_pull': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fd_gemm_f16': (c_int, [P, P]),
    'fd_gemm_can_emit_row_stats': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'fd_gemm_plan': (c_int, [c_void_p, c_void_p, c_void_p]),
    'fd_gemm_can_fuse_groupnorm': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'fd_gemm_gn_parts_chunks': (c_int, [P]),
    'fd_groupnorm_apply_parts_f16': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'fd_groupnorm_fold_linear_parts_f16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, c_int, P, P, P]),
    'fd_attention_f16': (c_int, [P, P]),
    'fd_xattn_image_bytes': (c_int64, [c_int, c_int]),
    'fd_xattn_pack_kv_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, P]),
    'fd_xattn_q_f16': (c_int, [P, P]),
    'fd_groupnorm_workspace_floats': (c_int64, [c_int, c_int]),
    'fd_groupnorm_nhwc_f16': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'fd_groupnorm_nhwc_ld_f16': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'fd_layernorm_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'fd_ln_row_stats_f16': (c_int, [P, P, c_int, c_int, c_int, c_float, P]),
    'fd_ln_finalize_stats_f32': (c_int, [P, P, c_int, c_int, c_int, c_float, P]),
    'fd_groupnorm_fold_linear_f16': (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_float, P, P, c_int, P, P, P]),
    'fd_softmax_rows_f16': (c_int, [P, c_int, c_int, c_int, c_float, P]),
    'fd_nchw_f32_to_nhwc_f16': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'fd_nhwc_f32_to_nchw_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    'fd_im2col_f16': (c_int, [P, P] + [c_int] * 12 + [P]),
    'fd_conv3x3_narrow_f16': (c_int, [P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    'fd_concat_channels_f16': (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    'fd_cfg_ddim_step_f32': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                     c_float, c_float, c_float, c_int, c_int, P]),
    'fd_axpby_f32': (c_int, [P, P, P, c_int64, c_float, c_float, c_int, P]),
    'fd_embed_tokens_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'fd_vit_assemble_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'fd_timestep_embedding_f16': (c_int, [P, c_int, P, c_int, c_int, P]),
    'fd_copy2d_f16': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'fd_repeat_rows_f16': (c_int, [P, c_int, P, c_int, c_int64, c_int, c_int, P]),
    'fd_plan_create': (c_int, [P]),
    'fd_plan_destroy': (c_int, [P]),
    'fd_plan_record_begin': (c_int, [P]),
    'fd_plan_record_end': (c_int, [P]),
    'fd_plan_size': (c_int, [P, P]),
    'fd_plan_replay': (c_int, [P, P]),
    'fd_region_blend_f32': (c_int, [P, P] + [c_int] * 7 + [c_float, P]),
    'fd_cast_f32_to_f16': (c_int, [P, P, c_int64, P]),
    'fd_cast_f16_to_f32': (c_int, [P, P, c_int64, P]),
}

ABI_VERSION = 12  # FD_ABI_VERSION in include/flexdiffuse_hip.h
_lib: Optional[ctypes.CDLL] = None


def declared_symbols():
    return sorted(_SIGNATURES)


def lib() -> ctypes.CDLL:
    '''Load (once) and return the shared object; raises if it has not been built.'''
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP extension has not been built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C '
                'flexdiffuse_amd/csrc`). flexdiffuse_amd has no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.fd_abi_version() != ABI_VERSION:
            raise RuntimeError('libflexdiffuse_hip.so ABI version mismatch; rebuild')
        _lib = handle
    return _lib


def check(rc: int, what: str = ''):
    if rc == FD_OK:
        return
    msg = lib().fd_last_error().decode('utf-8', 'replace')
    if rc in (FD_EINVAL, FD_ESHAPE):
        raise ValueError(f'{what}: {msg}')
    raise FDError(f'{what}: {msg} (code {rc})')


def call(name: str, *args):
    check(getattr(lib(), name)(*args), name)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('flexdiffuse_amd runs on an MI355X HIP device only; got a '
                               f'{t.device} tensor (there is no CPU fallback)')


def ptr(t: Optional[torch.Tensor]) -> c_void_p:
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream() -> c_void_p:
    '''torch's current stream of the current device as a hipStream_t.  Called once per kernel
    launch: the raw-handle query is ~8x cheaper than building a torch.cuda.Stream object
    (a third of the host time of a UNet forward went there).'''
    if _raw_stream is not None and _raw_device is not None:
        return c_void_p(_raw_stream(_raw_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def device_info(device: int = 0) -> dict:
    cu, khz, mem = c_int(0), c_int(0), c_int64(0)
    arch = ctypes.create_string_buffer(64)
    call('fd_device_info', device, ctypes.byref(cu), ctypes.byref(khz), ctypes.byref(mem),
         arch, 64)
    return {'cu_count': cu.value, 'clock_khz': khz.value, 'hbm_bytes': mem.value,
            'arch': arch.value.decode()}


class Plan():
    '''Launch plan (include/flexdiffuse_hip.h fd_plan_*): `with plan.record(): ...` runs the enclosed
    C-ABI calls AND records them; `plan.replay()` issues them again on the current stream without
    the per-op host work.  Every device address used inside must stay alive and unchanged (record
    inside a private torch memory pool and keep the pool), and nothing inside may be a torch
    kernel -- only library launches are recorded.'''

    def __init__(self):
        h = c_void_p()
        call('fd_plan_create', ctypes.byref(h))
        self._h = h
        self._replay = lib().fd_plan_replay

    def record(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            call('fd_plan_record_begin', self._h)
            try:
                yield self
            finally:
                call('fd_plan_record_end', self._h)
        return cm()

    def replay(self, on_stream: Optional[c_void_p] = None):
        rc = self._replay(self._h, stream() if on_stream is None else on_stream)
        if rc != FD_OK:
            check(rc, 'fd_plan_replay')

    def __len__(self):
        n = c_int(0)
        call('fd_plan_size', self._h, ctypes.byref(n))
        return n.value

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h is not None and _lib is not None:
            _lib.fd_plan_destroy(h)


_prof_on = False


def prof_enable(on: bool):
    global _prof_on
    _prof_on = bool(on)
    call('fd_prof_enable', int(bool(on)))


def prof_set_stride(stride: int):
    '''Sample every `stride`-th launch of each kernel family (1 = every launch).'''
    call('fd_prof_set_stride', int(stride))


def prof_is_on() -> bool:
    return _prof_on


def prof_calibrate(pairs: int = 256) -> float:
    '''Mean elapsed ms of an empty event bracket on the current stream.'''
    ms = c_double(0)
    call('fd_prof_calibrate', int(pairs), ctypes.byref(ms), stream())
    return ms.value


def prof_drain(cap: int = 1 << 17):
    '''Every recorded bracket in launch order as (family, tag, ms, work, executed) numpy arrays; forgets the records.'''
    import numpy as np
    fam, tag = np.zeros(cap, np.int32), np.zeros(cap, np.uint32)
    ms, work, ex = np.zeros(cap, np.float32), np.zeros(cap, np.float64), np.zeros(cap, np.float64)
    n = c_int64(0)
    call('fd_prof_drain', fam.ctypes.data, tag.ctypes.data, ms.ctypes.data, work.ctypes.data, ex.ctypes.data, cap,
         ctypes.byref(n))
    k = n.value
    return fam[:k], tag[:k], ms[:k], work[:k], ex[:k]


def prof_collect(family: int):
    ms, work, n = c_double(0), c_double(0), c_int64(0)
    call('fd_prof_collect', family, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n))
    return ms.value, work.value, n.value


def prof_collect2(family: int):
    '''(ms, declared work, executed work, launches) of the sampled launches of `family`.'''
    ms, work, ex, n = c_double(0), c_double(0), c_double(0), c_int64(0)
    call('fd_prof_collect2', family, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(ex), ctypes.byref(n))
    return ms.value, work.value, ex.value, n.value
