"""ctypes binding of libmmtg_hip.so (the C ABI declared in include/mmtg_hip.h).

PyTorch is plumbing here: tensors provide device memory (``data_ptr()``) and the
current HIP stream; every compute step of the hot path is one of the entry
points below.  There is NO fallback: if the library is missing or a call fails
a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ABI_VERSION = 12
EPI_NONE, EPI_GELU, EPI_TANH, EPI_RESID, EPI_DGELU, EPI_DTANH, EPI_ATOMIC, EPI_ROWDOT = range(8)
GEMM_NO_TR, GEMM_REGSTAGE, GEMM_SKINNY, GEMM_NO_SKINNY, GEMM_WIDE, GEMM_NO_WIDE = 1, 2, 4, 8, 16, 32
GEMM_PERSIST, GEMM_NO_PERSIST, GEMM_ROW_ORDER, GEMM_OCC4, GEMM_NO_OCC4, GEMM_COL_BLOCK, GEMM_P256, GEMM_NO_P8, GEMM_P8 = 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384
GEMM_GELU_GRAD = 32768
GEMM_P8_288 = 65536
GEMM_AUX2_BF16 = 131072     # mmtg_gemm_x3: the GELU epilogue stores the pre-activation as bf16 rows (bf16x3f)
PROF_CATS = ["gemm_bf16", "gemm_f32", "attn_fwd", "attn_bwd", "layernorm", "embed", "loss",
             "optim", "encoder", "decode", "misc"]

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmmtg_hip.so")
_lib = None

_vp, _i, _l, _f, _u = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint

_SIGS = {
    "mmtg_abi_version": ([], _i),
    "mmtg_last_error": ([], C.c_char_p),
    "mmtg_build_flags": ([], C.c_char_p),
    "mmtg_prof_enable": ([_i], _i),
    "mmtg_prof_read": ([_vp, _vp, _vp, _vp], _i),
    "mmtg_gemm_trace": ([_vp, _i], _i),
    "mmtg_gemm_cu_budget": ([_i], _i),
    "mmtg_debug_occupy": ([_i, _i, C.c_double, _vp], _i),
    "mmtg_gemm": ([_i, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp, _l, _vp, _i, _f, _i, _u, _u, _i, _vp], _i),
    "mmtg_gemm_gather": ([_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp, _i, _vp, _l, _vp, _i, _vp], _i),
    "mmtg_splitk_finish": ([_i, _vp, _i, _i, _i, _l, _vp, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _f, _vp], _i),
    "mmtg_colsum": ([_i, _vp, _l, _i, _i, _vp, _vp, _l, _vp], _i),
    "mmtg_colsum_ws": ([_i, _i], _l),
    "mmtg_layernorm_fwd": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp], _i),
    "mmtg_layernorm_bwd_ws": ([_i, _i], _l),
    "mmtg_layernorm_bwd": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _u, _u, _vp, _vp, _l, _vp], _i),
    "mmtg_layernorm_bwd_x3": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _l, _u, _u, _vp, _vp, _l, _vp], _i),
    "mmtg_attn_fwd": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _u, _u, _vp], _i),
    "mmtg_attn_fwd_x3": ([_vp, _l, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _u, _u, _vp], _i),
    "mmtg_attn_bwd_x3": ([_vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _u, _u, _vp], _i),
    "mmtg_attn_trace": ([_vp], _i),
    "mmtg_attn_bwd": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _u, _u, _i, _vp], _i),
    "mmtg_embed_condition": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mmtg_segment_sum": ([_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mmtg_embed_add": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u, _u, _vp], _i),
    "mmtg_embed_add_bwd": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _u, _u, _vp, _l, _vp], _i),
    "mmtg_embed_add_bwd_ws": ([_i, _i, _i], _l),
    "mmtg_dropout_apply": ([_i, _vp, _vp, _l, _i, _u, _u, _vp], _i),
    "mmtg_loss_fwd": ([_i, _vp, _l, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "mmtg_loss_bwd": ([_i, _i, _vp, _l, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _l, _i, _vp], _i),
    "mmtg_loss_bwd_x3": ([_vp, _l, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _l, _l, _i, _vp], _i),
    "mmtg_gru_cell_fwd": ([_i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _vp], _i),
    "mmtg_gru_cell_bwd_fused": ([_i, _vp, _l, _vp, _vp, _i, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _vp], _i),
    "mmtg_rnn_cell_fwd": ([_i, _i, _vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _i, _i, _vp], _i),
    "mmtg_rnn_cell_bwd": ([_i, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _i, _vp, _l, _i, _i, _vp], _i),
    "mmtg_gru_cell_bwd": ([_i, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _vp], _i),
    "mmtg_alpha_attn_fwd": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mmtg_alpha_attn_bwd": ([_i, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp], _i),
    "mmtg_beta_fuse_fwd": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "mmtg_beta_fuse_bwd": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _l, _vp], _i),
    "mmtg_beta_fuse_bwd_ws": ([_i, _i, _i], _l),
    "mmtg_prefetch": ([_vp, _l, _i, _vp, _vp], _i),
    "mmtg_zero_ranges": ([_vp, _vp, _i, _vp], _i),
    "mmtg_sumsq": ([_vp, _l, _vp, _vp, _l, _vp], _i),
    "mmtg_sumsq_ws": ([_l], _l),
    "mmtg_adamw": ([_vp, _vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp], _i),
    "mmtg_gemm_x3": ([_i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _vp, _l, _l, _vp, _i, _vp, _l, _vp, _u, _u, _i, _vp], _i),
    "mmtg_split_planes": ([_vp, _l, _i, _i, _vp, _l, _l, _vp], _i),
    "mmtg_layernorm_fwd_x3": ([_vp, _vp, _l, _l, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp], _i),
    "mmtg_cast_f32_to": ([_i, _vp, _vp, _l, _vp], _i),
    "mmtg_cast_pad_rows": ([_i, _vp, _l, _vp, _l, _i, _i, _vp], _i),
    "mmtg_cast_to_f32": ([_i, _vp, _vp, _l, _vp], _i),
    "mmtg_axpy_f32": ([_vp, _vp, _f, _l, _vp], _i),
    "mmtg_slab_sum": ([_vp, _i, _l, _vp, _i, _l, _vp], _i),
    "mmtg_wgrad_group": ([_i, _i, _vp, _i, _i, _vp, _l, _vp, _l, _i, _vp], _i),
    "mmtg_transpose_batch": ([_i, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "mmtg_logits_process_argmax": ([_vp, _l, _i, _vp, _l, _vp, _f, _f, _vp, _i, _vp], _i),
    "mmtg_decode_embed": ([_i, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mmtg_decode_embed_add": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp], _i),
    "mmtg_decode_gemm": ([_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _vp, _i, _f, _i, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l,
                          _vp, _vp, _vp, _vp, _vp], _i),
    "mmtg_ln_fold_weights": ([_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "mmtg_decode_gemm_x3": ([_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _vp, _l, _l, _vp, _vp, _vp, _i, _f, _i, _vp, _l, _vp, _i, _vp, _l,
                             _vp, _l, _vp, _vp, _vp, _vp, _vp], _i),
    "mmtg_ln_fold_weights_x3": ([_vp, _l, _l, _vp, _vp, _vp, _vp, _l, _l, _vp, _vp, _i, _i, _vp], _i),
    "mmtg_decode_attn_split_x3": ([_vp, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _vp], _i),
    "mmtg_decode_embed_x3": ([_vp, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mmtg_decode_attn": ([_i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mmtg_decode_attn_split": ([_i, _vp, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mmtg_logits_process_sample": ([_vp, _l, _i, _vp, _l, _vp, _f, _f, _i, _f, _vp, _vp, _vp, _i, _vp], _i),
    "mmtg_decode_sample": ([_vp, _l, _i, _vp, _l, _vp, _i, _i, _f, _f, _i, _f, _vp, _l, _i, _vp, _vp], _i),
    "mmtg_decode_select": ([_vp, _l, _i, _vp, _l, _vp, _i, _i, _f, _f, _i, _vp, _vp], _i),
    "mmtg_decode_advance": ([_vp, _vp], _i),
    "mmtg_decode_mlp_ws_floats": ([_i], _l),
    "mmtg_decode_mlp_sync_words": ([], _i),
    "mmtg_decode_mlp_census": ([_vp, _vp], _i),
    "mmtg_decode_mlp": ([_i, _i, _vp, _l, _vp, _i, _f, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp, _l, _vp, _i, _vp, _vp], _i),
    "mmtg_colsum_batch": ([_vp, _i, _vp], _i),
    "mmtg_layernorm_bwd_partial": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _u, _u, _i, _vp, _l, _vp, _vp], _i),
    "mmtg_attn_bwd_dbias_rows": ([_i, _i, _i], _i),
    "mmtg_layernorm_bwd_x3_partial": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _l, _u, _u, _i, _vp, _l, _vp, _vp], _i),
    "mmtg_comm_unique_id": ([_vp], _i),
    "mmtg_comm_init": ([_i, _i, _vp], _i),
    "mmtg_comm_info": ([_vp, _vp, _vp, _vp], _i),
    "mmtg_allreduce_bucket": ([_vp, _l, _i, _vp], _i),
    "mmtg_allreduce_bucket_async": ([_vp, _l, _i, _vp], _i),
    "mmtg_comm_join": ([_vp], _i),
    "mmtg_comm_destroy": ([], _i),
}


def lib_path():
    return _LIB_PATH


def source_sha():
    """sha1 over the kernel sources and the ABI header the library is built from (csrc/*.hip, csrc/*.h,
    include/mmtg_hip.h): ties a committed PMC measurement to the kernels it was taken on (bench.py roofline.traffic)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    h = hashlib.sha1()
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(".hip") or f.endswith(".h")):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(here, "..", "include", "mmtg_hip.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """Load the shared library (once).  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "libmmtg_hip.so is not built (%s). Run `python -m mmtg_amd.build` -- the MMTG hot "
                "path has no CPU/PyTorch fallback." % _LIB_PATH)
        L = C.CDLL(_LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = res
        if L.mmtg_abi_version() != ABI_VERSION:
            raise RuntimeError("libmmtg_hip.so ABI version mismatch")
        flags = L.mmtg_build_flags().decode()
        if flags and os.environ.get("MMTG_ALLOW_DIAGNOSTIC_BUILD") != "1":
            raise RuntimeError("libmmtg_hip.so is a diagnostic build (MMTG_EXTRA_DEFS=%r): rebuild with `python -m mmtg_amd.build` "
                               "or set MMTG_ALLOW_DIAGNOSTIC_BUILD=1" % flags)
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().mmtg_last_error().decode()))


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("unsupported storage dtype %s" % t.dtype)


def torch_dtype(code):
    return torch.float32 if code == F32 else torch.bfloat16


def drop_thresh(p):
    """dropout probability -> 32-bit threshold (0 disables)."""
    return 0 if p <= 0.0 else min(int(p * 4294967296.0), 4294967295)


# ------------------------------------------------------------------ profiler
def prof_enable(on=True):
    _check(lib().mmtg_prof_enable(int(on)), "prof_enable")


def prof_read():
    n = len(PROF_CATS)
    la = (C.c_int * n)()
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    by = (C.c_double * n)()
    _check(lib().mmtg_prof_read(C.addressof(la), C.addressof(ms), C.addressof(fl), C.addressof(by)), "prof_read")
    return {PROF_CATS[i]: {"launches": la[i], "ms": ms[i], "flops": fl[i], "bytes": by[i]} for i in range(n)}


# ------------------------------------------------------------------ GEMM
def zero_ranges(base, desc, n):
    """Zero n (first element, count) ranges of the fp32 tensor `base` in one launch; desc: int64 device tensor [n, 2]."""
    _check(lib().mmtg_zero_ranges(_p(base), _p(desc), int(n), _stream()), "zero_ranges")


def gemm_cu_budget(cus):
    """CUs the eight-phase kernel's tile rule may count on (0 = all, > 0 = that many, < 0 = all but that many)."""
    _check(lib().mmtg_gemm_cu_budget(int(cus)), "gemm_cu_budget")


# ------------------------------------------------------------------ the data-parallel exchange (csrc/comm.hip)
COMM_ID_BYTES = 128


def comm_unique_id():
    """Rank 0: the 128-byte id every rank passes to comm_init (ship it over any side channel)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().mmtg_comm_unique_id(C.addressof(buf)), "comm_unique_id")
    return buf.raw


def comm_init(rank, world, unique_id):
    """Collective: the process's RCCL communicator on the current device (one process per GPU)."""
    if len(unique_id) != COMM_ID_BYTES:
        raise ValueError("comm_init: the id is %d bytes, not %d" % (len(unique_id), COMM_ID_BYTES))
    buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
    _check(lib().mmtg_comm_init(int(rank), int(world), C.addressof(buf)), "comm_init")


def comm_info():
    """{"rank", "world", "device", "rccl_version"} of the live communicator (world 0: none)."""
    v = (C.c_int * 4)()
    base = C.addressof(v)
    _check(lib().mmtg_comm_info(base, base + 4, base + 8, base + 12), "comm_info")
    return {"rank": v[0], "world": v[1], "device": v[2], "rccl_version": v[3]}


def allreduce_bucket(t, stream=None):
    """In-place SUM all-reduce of the contiguous fp32 / bf16 tensor `t`, enqueued on `stream` (default: the current one)."""
    assert t.is_contiguous()
    st = _stream() if stream is None else stream.cuda_stream
    _check(lib().mmtg_allreduce_bucket(t.data_ptr(), t.numel(), dt(t), st), "allreduce_bucket")


def allreduce_bucket_async(t):
    """The same on the library's side stream, ordered after what the current stream holds now; comm_join() ends the overlap."""
    assert t.is_contiguous()
    _check(lib().mmtg_allreduce_bucket_async(t.data_ptr(), t.numel(), dt(t), _stream()), "allreduce_bucket_async")


def comm_join():
    """The current stream waits for every bucket enqueued with allreduce_bucket_async so far."""
    _check(lib().mmtg_comm_join(_stream()), "comm_join")


def comm_destroy():
    _check(lib().mmtg_comm_destroy(), "comm_destroy")


def debug_occupy(workgroups, usec, lds_bytes=163840, stream=None):
    """Measurement hook: hold `workgroups` CU-sized slots (lds_bytes of LDS each) for `usec` microseconds on `stream`."""
    st = _stream() if stream is None else stream.cuda_stream
    _check(lib().mmtg_debug_occupy(int(workgroups), int(lds_bytes), float(usec), st), "debug_occupy")


def gemm_trace(buf=None):
    """Switch the in-kernel timeline of the LDS-DMA GEMMs on (buf: int64 CUDA tensor [max_wgs, 6]) or off."""
    _check(lib().mmtg_gemm_trace(_p(buf), 0 if buf is None else buf.shape[0]), "gemm_trace")


def gemm(A, B, C_, M, N, K, transA=False, transB=False, lda=None, ldb=None, ldc=None, bias=None,
         epi=EPI_NONE, aux=None, ldaux=0, aux2=None, out_f32=False, alpha=1.0, splits=1,
         drop_p=0.0, drop_seed=0, flags=0, dtype=None):
    """C[M,N] = epi(opA * opB); see include/mmtg_hip.h (mmtg_gemm)."""
    d = dt(A) if dtype is None else dtype
    if lda is None:
        lda = M if transA else K
    if ldb is None:
        ldb = K if transB else N
    if ldc is None:
        ldc = N
    if aux is not None and not ldaux:
        ldaux = N
    _check(lib().mmtg_gemm(d, int(transA), int(transB), M, N, K, _p(A), lda, _p(B), ldb, _p(C_), ldc,
                           _p(bias), epi, _p(aux), ldaux, _p(aux2), int(out_f32), float(alpha), splits,
                           drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, flags, _stream()), "gemm")


EPI_SPLIT = 8
EPI_TANH_ADD = 9


# ------------------------------------------------------------------ split-precision ("bf16x3") products
class Planes:
    """(hi | lo) bf16 plane pair of an fp32 matrix [rows, cols]: ``t`` is a bf16 tensor [2, rows, ld] (or any tensor whose
    storage holds the two planes ``plane`` elements apart, the hi plane at its data pointer)."""
    __slots__ = ("t", "rows", "cols", "ld", "plane")

    def __init__(self, t, rows, cols, ld=None, plane=None):
        self.t, self.rows, self.cols = t, rows, cols
        self.ld = cols if ld is None else ld
        self.plane = rows * self.ld if plane is None else plane

    @staticmethod
    def empty(rows, cols, device):
        return Planes(torch.empty(2, rows, cols, device=device, dtype=torch.bfloat16), rows, cols)

    def float(self):
        """hi + lo as fp32 [rows, cols] (tests)."""
        flat = self.t.reshape(-1)
        hi = torch.as_strided(flat, (self.rows, self.cols), (self.ld, 1), 0)
        lo = torch.as_strided(flat, (self.rows, self.cols), (self.ld, 1), self.plane)
        return hi.float() + lo.float()


def split_planes(src, rows, cols, out, lds=None):
    """fp32 [rows, cols] -> Planes ``out``."""
    _check(lib().mmtg_split_planes(_p(src), cols if lds is None else lds, rows, cols, _p(out.t), out.ld, out.plane, _stream()), "split_planes")
    return out


def layernorm_fwd_x3(x, out, gamma, beta, mean, rstd, rows, cols, eps=1e-5, xb=None):
    _check(lib().mmtg_layernorm_fwd_x3(_p(x), _p(out.t), out.ld, out.plane, _p(gamma), _p(beta), _p(mean), _p(rstd), rows, cols,
                                       float(eps), _p(xb), _stream()), "layernorm_fwd_x3")


def gemm_x3(A, B, C_, M, N, K, planes=None, ldc=None, bias=None, epi=EPI_NONE, aux=None, ldaux=0, aux2=None,
            drop_p=0.0, drop_seed=0, flags=0):
    """C[M,N] (fp32, nullable) / planes (Planes, nullable) = epi(A B^T + bias), A [M,K] and B [N,K] as Planes
    (include/mmtg_hip.h, mmtg_gemm_x3)."""
    if aux is not None and not ldaux:
        ldaux = N
    _check(lib().mmtg_gemm_x3(M, N, K, _p(A.t), A.ld, A.plane, _p(B.t), B.ld, B.plane, _p(C_), N if ldc is None else ldc,
                              0 if planes is None else _p(planes.t), 0 if planes is None else planes.ld,
                              0 if planes is None else planes.plane, _p(bias), epi, _p(aux), ldaux, _p(aux2),
                              drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, flags, _stream()), "gemm_x3")


def gemm_gather(mode, A, B, C_, M, N, K, rows, table_rows, lda, ldb, ldc=None, bias=None, epi=EPI_NONE, aux=None, ldaux=0,
                aux_rows=None, splits=1):
    """Products with table-gathered operand rows (include/mmtg_hip.h, mmtg_gemm_gather): mode 0 forward, mode 1 weight gradient."""
    _check(lib().mmtg_gemm_gather(int(mode), M, N, K, _p(A), lda, _p(B), ldb, _p(C_), N if ldc is None else ldc, _p(bias), epi,
                                  _p(rows), int(table_rows), _p(aux), ldaux or N, _p(aux_rows), int(splits), _stream()), "gemm_gather")


def splitk_finish(part, splits, M, N, out, bias=None, epi=EPI_NONE, aux=None, ldaux=0, ldp=None, ldo=None,
                  ln_gamma=None, ln_beta=None, ln_out=None, eps=1e-5):
    """out = epi(sum of the `splits` fp32 slabs of an EPI_SPLIT product + bias) [, ln_out = LayerNorm(out)]."""
    _check(lib().mmtg_splitk_finish(dt(out), _p(part), splits, M, N, N if ldp is None else ldp, _p(bias), epi, _p(aux),
                                    ldaux or N, _p(out), N if ldo is None else ldo, _p(ln_gamma), _p(ln_beta), _p(ln_out),
                                    float(eps), _stream()), "splitk_finish")


_colsum_ws = {}


def colsum(X, M, N, out, ldx=None, ws=None):
    """out[n] += sum_m X[m, n], in a fixed order (no atomics).  Tall inputs (mmtg_colsum_ws(M, N) > 0) sum in two stages through
    `ws` (default: a cached workspace per device, size and stream)."""
    need = int(lib().mmtg_colsum_ws(M, N))
    if need and (ws is None or ws.numel() < need):
        key = (X.device, need, _stream())        # per stream: two streams summing the same width must not share slices
        ws = _colsum_ws.get(key)
        if ws is None:
            ws = _colsum_ws[key] = torch.empty(need, device=X.device, dtype=torch.float32)
    _check(lib().mmtg_colsum(dt(X), _p(X), N if ldx is None else ldx, M, N, _p(out), _p(ws) if need else 0, need, _stream()), "colsum")


class ColsumItem(C.Structure):
    """mmtg_colsum_item of include/mmtg_hip.h."""
    _fields_ = [("X", C.c_void_p), ("out", C.c_void_p), ("ldx", C.c_long), ("M", C.c_int), ("N", C.c_int)]


_colsum_batch_cache = {}


def colsum_batch(items):
    """items: list of (X data_ptr, out data_ptr, ldx, M, N) -- fp32 rows, M <= 2048: out[c] += sum_r X[r * ldx + c] for every item in ONE
    launch (64 items per launch), each in mmtg_colsum's order.  The ctypes array is cached per item list (a step repeats its list)."""
    if not items:
        return
    key = tuple(items)
    arr = _colsum_batch_cache.get(key)
    if arr is None:
        if len(_colsum_batch_cache) > 64:
            _colsum_batch_cache.clear()
        arr = _colsum_batch_cache[key] = (ColsumItem * len(items))(*[ColsumItem(*it) for it in items])
    _check(lib().mmtg_colsum_batch(C.addressof(arr), len(items), _stream()), "colsum_batch")


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, y, gamma, beta, mean, rstd, rows, cols, eps=1e-5):
    _check(lib().mmtg_layernorm_fwd(dt(x), _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), rows, cols,
                                    float(eps), _stream()), "layernorm_fwd")


_ln_ws = {}


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, cols, dx_masked=None, drop_p=0.0,
                  drop_seed=0, dcolsum=None, ws=None):
    if ws is None:   # convenience for tests; the engine passes its own workspace
        need = lib().mmtg_layernorm_bwd_ws(rows, cols)
        key = (x.device, need)
        ws = _ln_ws.get(key)
        if ws is None:
            ws = _ln_ws[key] = torch.empty(need, device=x.device, dtype=torch.float32)
    _check(lib().mmtg_layernorm_bwd(dt(x), _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
                                    _p(dgamma), _p(dbeta), rows, cols, _p(dx_masked), drop_thresh(drop_p),
                                    drop_seed & 0xFFFFFFFF, _p(dcolsum), _p(ws), ws.numel(), _stream()), "layernorm_bwd")


def layernorm_bwd_partial(dy, x, gamma, mean, rstd, dres, dx, rows, cols, ws, dx_masked=None, drop_p=0.0, drop_seed=0, want_colsum=False):
    """First stage of layernorm_bwd only; returns the number of partial rows left in ws ([rows][3][cols]) for colsum_batch."""
    n = C.c_int(0)
    _check(lib().mmtg_layernorm_bwd_partial(dt(x), _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), rows, cols,
                                            _p(dx_masked), drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, int(want_colsum),
                                            _p(ws), ws.numel(), C.addressof(n), _stream()), "layernorm_bwd_partial")
    return n.value


def layernorm_bwd_x3(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, cols, dx_planes, drop_p=0.0, drop_seed=0, dcolsum=None, ws=None):
    """fp32 LayerNorm backward whose (masked) input gradient goes to the Planes ``dx_planes`` (x3 mode)."""
    _check(lib().mmtg_layernorm_bwd_x3(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma), _p(dbeta), rows, cols,
                                       _p(dx_planes.t), dx_planes.plane, drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _p(dcolsum),
                                       _p(ws), ws.numel(), _stream()), "layernorm_bwd_x3")


def layernorm_bwd_x3_partial(dy, x, gamma, mean, rstd, dres, dx, rows, cols, dx_planes, ws, drop_p=0.0, drop_seed=0, want_colsum=False):
    """First stage of layernorm_bwd_x3 only; returns the number of partial rows left in ws ([rows][3][cols]) for colsum_batch."""
    n = C.c_int(0)
    _check(lib().mmtg_layernorm_bwd_x3_partial(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), rows, cols,
                                               _p(dx_planes.t), dx_planes.plane, drop_thresh(drop_p), drop_seed & 0xFFFFFFFF,
                                               int(want_colsum), _p(ws), ws.numel(), C.addressof(n), _stream()), "layernorm_bwd_x3_partial")
    return n.value


# ------------------------------------------------------------------ attention
def attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=0.0, drop_seed=0):
    _check(lib().mmtg_attn_fwd(dt(qkv), _p(qkv), _p(keep), _p(out), _p(lse), B, T, nH, dh,
                               drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _stream()), "attn_fwd")


def attn_fwd_x3(qkv_planes, keep, out, out_planes, lse, B, T, nH, dh, drop_p=0.0, drop_seed=0):
    """Split-precision attention forward; qkv_planes: Planes [B*T, 3D] (what the c_attn product writes), out fp32, out_planes:
    Planes or None (include/mmtg_hip.h)."""
    if qkv_planes.ld != 3 * nH * dh:
        raise ValueError("attn_fwd_x3: the qkv plane pair must be dense ([B*T, 3D], ld = 3D)")
    _check(lib().mmtg_attn_fwd_x3(_p(qkv_planes.t), qkv_planes.plane, _p(keep), _p(out), 0 if out_planes is None else _p(out_planes.t),
                                  0 if out_planes is None else out_planes.plane, _p(lse), B, T, nH, dh,
                                  drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _stream()), "attn_fwd_x3")


def attn_bwd_x3_dq_floats(B, T, D):
    """Floats of attn_bwd_x3's dq32 scratch: one [B*T, D] buffer per block of 128 keys."""
    return (-(-T // 128)) * B * T * D


def attn_bwd_x3_ws(B, T, D):
    """Floats of the dbias workspace of attn_bwd_x3: the partial rows + the workspaces of their ordered column sums (mmtg_colsum_ws:
    128 x N floats once a reduction has more than 2048 rows)."""
    nkv, nq = B * (-(-T // 128)), -(-(B * T) // 16)
    return (nkv + nq) * 3 * D + (128 * 3 * D if nkv > 2048 else 0) + (128 * D if nq > 2048 else 0)


def attn_bwd_x3(qkv_planes, keep, out, dout_planes, lse, delta, dq32, dqkv_planes, B, T, nH, dh, drop_p=0.0, drop_seed=0, dbias=None,
                dbias_ws=None, delta_ready=False):
    """Split-precision attention backward: qkv [B*T, 3D] and d(ctx) [B*T, D] as Planes, out fp32; d(qkv) as the Planes
    ``dqkv_planes`` [B*T, 3D]; dq32: fp32 scratch of attn_bwd_x3_dq_floats(B, T, D) elements."""
    if qkv_planes.ld != 3 * nH * dh or dout_planes.ld != nH * dh or dqkv_planes.ld != 3 * nH * dh:
        raise ValueError("attn_bwd_x3: the plane pairs must be dense (ld = columns)")
    _check(lib().mmtg_attn_bwd_x3(_p(qkv_planes.t), qkv_planes.plane, _p(keep), _p(out), _p(dout_planes.t), dout_planes.plane, _p(lse), _p(delta),
                                  int(delta_ready), _p(dq32), dq32.numel(), _p(dqkv_planes.t), dqkv_planes.plane,
                                  _p(dbias), _p(dbias_ws), 0 if dbias_ws is None else dbias_ws.numel(), B, T, nH, dh,
                                  drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _stream()), "attn_bwd_x3")


ATTN_ELEM_MASK = 1      # attn_bwd flags: the tiled / split-precision kernels' per-element dropout mask in the whole-head kernels


def attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=0.0, drop_seed=0, delta_ready=False,
             dbias=None, dbias_ws=None, flags=0):
    _check(lib().mmtg_attn_bwd(dt(qkv), _p(qkv), _p(keep), _p(out), _p(dout), _p(lse), _p(delta), int(delta_ready), _p(dq32),
                               _p(dqkv), _p(dbias), _p(dbias_ws), B, T, nH, dh, drop_thresh(drop_p), drop_seed & 0xFFFFFFFF,
                               int(flags), _stream()), "attn_bwd")


def attn_trace(buf):
    _check(lib().mmtg_attn_trace(_p(buf)), "attn_trace")


ATTN_DBIAS_ROWS = 2


def attn_bwd_dbias_rows(dtype_code, B, T):
    """Partial bias rows attn_bwd(flags=ATTN_DBIAS_ROWS) leaves unsummed at the head of dbias_ws for this shape (0: it sums inside the call)."""
    return int(lib().mmtg_attn_bwd_dbias_rows(dtype_code, B, T))


def attn_bwd_bias_rows(B, T, dtype_code):
    """Rows of the dbias_ws scratch of attn_bwd ([rows, 3*D] f32): one partial bias row per (batch row, key block) + 44 rows the
    tiled kernels' ordered column sum of dQ uses as its workspace."""
    return B * (-(-T // (128 if dtype_code == F32 else 256))) + 44


# ------------------------------------------------------------------ conditioning front end
def embed_condition(table, topic_ids, targets, c, x, B, P, L, S, E, two_sents, V):
    _check(lib().mmtg_embed_condition(dt(table), _p(table), _p(topic_ids), _p(targets), _p(c), _p(x),
                                      B, P, L, S, E, two_sents, V, _stream()), "embed_condition")


def segment_sum(g, out, B, P, L, S, H, two_sents):
    _check(lib().mmtg_segment_sum(dt(g), _p(g), _p(out), B, P, L, S, H, two_sents, _stream()), "segment_sum")


def embed_add(g, wpe, wte, type_ids, h, M, T, D, drop_p=0.0, drop_seed=0):
    _check(lib().mmtg_embed_add(dt(g), _p(g), _p(wpe), _p(wte), _p(type_ids), _p(h), M, T, D,
                                drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _stream()), "embed_add")


_embed_ws = {}


def embed_add_bwd(dh, type_ids, dwpe, dwte, M, T, D, ntypes, drop_p=0.0, drop_seed=0, ws=None):
    need = int(lib().mmtg_embed_add_bwd_ws(M, D, ntypes))
    if ws is None or ws.numel() < need:     # (the engine passes its own workspace; tests get a cached one)
        key = (dh.device, need)
        ws = _embed_ws.get(key)
        if ws is None:
            ws = _embed_ws[key] = torch.empty(need, device=dh.device, dtype=torch.float32)
    _check(lib().mmtg_embed_add_bwd(dt(dh), _p(dh), _p(type_ids), _p(dwpe), _p(dwte), M, T, D, ntypes,
                                    drop_thresh(drop_p), drop_seed & 0xFFFFFFFF, _p(ws), ws.numel(), _stream()), "embed_add_bwd")


def dropout_apply(x, y, n, drop_p, drop_seed):
    _check(lib().mmtg_dropout_apply(dt(x), _p(x), _p(y), n, 0, drop_thresh(drop_p), drop_seed & 0xFFFFFFFF,
                                    _stream()), "dropout_apply")


# ------------------------------------------------------------------ loss
def loss_fwd(logits, ldl, V, topic_ids, targets, ratings, stage, label_zero, B, P, L, batch_den,
             nll, lse, sample_ce, coef, scalars):
    _check(lib().mmtg_loss_fwd(dt(logits), _p(logits), ldl, V, _p(topic_ids), _p(targets), _p(ratings), stage,
                               int(label_zero), B, P, L, float(batch_den), _p(nll), _p(lse), _p(sample_ce),
                               _p(coef), _p(scalars), _stream()), "loss_fwd")


def loss_bwd(logits, ldl, V, topic_ids, targets, lse, coef, gscale, B, P, L, dlogits, ldd, Vpad, lm_coef=0.0):
    _check(lib().mmtg_loss_bwd(dt(dlogits), dt(logits), _p(logits), ldl, V, _p(topic_ids), _p(targets), _p(lse), _p(coef),
                               float(gscale), float(lm_coef), B, P, L, _p(dlogits), ldd, Vpad, _stream()), "loss_bwd")


def loss_bwd_x3(logits, ldl, V, topic_ids, targets, lse, coef, gscale, B, P, L, planes, Vpad, lm_coef=0.0):
    """d(logits) of fp32 logits as a Planes pair (x3 mode)."""
    _check(lib().mmtg_loss_bwd_x3(_p(logits), ldl, V, _p(topic_ids), _p(targets), _p(lse), _p(coef), float(gscale), float(lm_coef), B, P, L,
                                  _p(planes.t), planes.ld, planes.plane, Vpad, _stream()), "loss_bwd_x3")


# ------------------------------------------------------------------ encoder pieces
def gru_cell_bwd_fused(rows, ld_rows, carry, part, splits, save, h_prev, dgi, dgh, dh_prev, B, H, ld_hp=None, ld_dgi=None):
    """gru_cell_bwd with dh_t = rows + carry + sum of the carry product's slabs assembled in the kernel."""
    _check(lib().mmtg_gru_cell_bwd_fused(dt(dgi), _p(rows), ld_rows, _p(carry), _p(part), splits, _p(save), _p(h_prev),
                                         H if ld_hp is None else ld_hp, _p(dgi), 3 * H if ld_dgi is None else ld_dgi,
                                         _p(dgh), _p(dh_prev), B, H, _stream()), "gru_cell_bwd_fused")



def gru_cell_fwd(gi, gh, h_prev, h, save, B, H, ld_gi=None, ld_hp=None, ld_h=None, ld_gh=None):
    _check(lib().mmtg_gru_cell_fwd(dt(gi), _p(gi), 3 * H if ld_gi is None else ld_gi, _p(gh), 3 * H if ld_gh is None else ld_gh, _p(h_prev),
                                   H if ld_hp is None else ld_hp, _p(h), H if ld_h is None else ld_h, _p(save),
                                   B, H, _stream()), "gru_cell_fwd")


def gru_cell_bwd(dh, save, h_prev, dgi, dgh, dh_prev, B, H, ld_hp=None, ld_dgi=None):
    _check(lib().mmtg_gru_cell_bwd(dt(dgi), _p(dh), _p(save), _p(h_prev), H if ld_hp is None else ld_hp, _p(dgi),
                                   3 * H if ld_dgi is None else ld_dgi, _p(dgh), _p(dh_prev), B, H, _stream()),
           "gru_cell_bwd")


RNN_RELU, RNN_LSTM = 0, 1
RNN_GATES = {"GRU": 3, "LSTM": 4, "RNN": 1}


def rnn_cell_fwd(kind, gi, gh, c_prev, h, c, save, B, H, ld_gi, ld_h, ld_gh=None):
    """ReLU-RNN / LSTM cell of one step (reference model.py:41-59 encoder types)."""
    G = 4 if kind == RNN_LSTM else 1
    _check(lib().mmtg_rnn_cell_fwd(dt(gi), kind, _p(gi), ld_gi, _p(gh), G * H if ld_gh is None else ld_gh, _p(c_prev), _p(h), ld_h,
                                   _p(c), _p(save), B, H, _stream()), "rnn_cell_fwd")


def rnn_cell_bwd(kind, rows, ld_rows, part, save, c_prev, h, ld_h, dc, dc_in, da, ld_da, B, H):
    _check(lib().mmtg_rnn_cell_bwd(dt(da), kind, _p(rows), ld_rows, _p(part), _p(save), _p(c_prev), _p(h), ld_h, _p(dc), int(dc_in),
                                   _p(da), ld_da, B, H, _stream()), "rnn_cell_bwd")


def alpha_attn_fwd(qkv, prior, ctx, probs, kl, B, S, H, heads):
    _check(lib().mmtg_alpha_attn_fwd(dt(qkv), _p(qkv), _p(prior), _p(ctx), _p(probs), _p(kl), B, S, H, heads,
                                     _stream()), "alpha_attn_fwd")


def alpha_attn_bwd(qkv, prior, probs, dctx, dkl, dqkv, B, S, H, heads):
    _check(lib().mmtg_alpha_attn_bwd(dt(qkv), _p(qkv), _p(prior), _p(probs), _p(dctx), float(dkl), _p(dqkv),
                                     B, S, H, heads, _stream()), "alpha_attn_bwd")


def beta_fuse_fwd(topic, img, txt, att_w, att_b, o, a, B, S, H):
    _check(lib().mmtg_beta_fuse_fwd(dt(topic), _p(topic), _p(img), _p(txt), _p(att_w), _p(att_b), _p(o), _p(a),
                                    B, S, H, _stream()), "beta_fuse_fwd")


_beta_ws = {}


def beta_fuse_bwd(topic, img, txt, att_w, a, d_o, dtopic, dimg, dtxt, datt_w, datt_b, B, S, H):
    need = int(lib().mmtg_beta_fuse_bwd_ws(B, S, H))             # 0.7 MB at the released sizes
    key = (d_o.device, need, _stream())                          # one workspace per device, size and stream (not one per step)
    ws = _beta_ws.get(key)
    if ws is None:
        if len(_beta_ws) > 64:
            _beta_ws.clear()
        ws = _beta_ws[key] = torch.empty(need, device=d_o.device, dtype=torch.float32)
    _check(lib().mmtg_beta_fuse_bwd(dt(topic), _p(topic), _p(img), _p(txt), _p(att_w), _p(a), _p(d_o), _p(dtopic),
                                    _p(dimg), _p(dtxt), _p(datt_w), _p(datt_b), B, S, H, _p(ws), ws.numel(), _stream()), "beta_fuse_bwd")


# ------------------------------------------------------------------ optimizer / casts
def prefetch(t, sink, workgroups=256, stream=None):
    """Pull tensor `t` into the Infinity Cache on `stream` (default: the current one)."""
    _check(lib().mmtg_prefetch(_p(t), t.numel() * t.element_size(), int(workgroups), _p(sink),
                               _stream() if stream is None else stream.cuda_stream), "prefetch")


_sumsq_ws = {}


def sumsq(x, n, out):
    """out[0] = sum of squares of x[:n] (written, not accumulated), in a fixed summation order."""
    need = int(lib().mmtg_sumsq_ws(n))
    key = (x.device, need, _stream())
    ws = _sumsq_ws.get(key)
    if ws is None:
        ws = _sumsq_ws[key] = torch.empty(need, device=x.device, dtype=torch.float32)
    _check(lib().mmtg_sumsq(_p(x), n, _p(out), _p(ws), need, _stream()), "sumsq")


def adamw(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, wd, step, normsq, max_norm, grad_scale=1.0, count=None, p_lo=None):
    """count: optional device scalar (float32) = global row count; g is then a SUM over rows (see the header).
    p_lo: the x3 mode's lo plane (p_bf16 is then the hi plane)."""
    _check(lib().mmtg_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), _p(p_lo), n, float(lr), float(beta1), float(beta2),
                            float(eps), float(wd), int(step), _p(normsq), float(max_norm), float(grad_scale),
                            _p(count), _stream()), "adamw")


def cast_f32_to(src, dst, n):
    _check(lib().mmtg_cast_f32_to(dt(dst), _p(src), _p(dst), n, _stream()), "cast_f32_to")


def cast_pad_rows(src, lds, dst, ldd, rows, cols):
    _check(lib().mmtg_cast_pad_rows(dt(dst), _p(src), lds, _p(dst), ldd, rows, cols, _stream()), "cast_pad_rows")


def cast_to_f32(src, dst, n):
    _check(lib().mmtg_cast_to_f32(dt(src), _p(src), _p(dst), n, _stream()), "cast_to_f32")


def axpy_f32(y, x, a, n):
    _check(lib().mmtg_axpy_f32(_p(y), _p(x), float(a), n, _stream()), "axpy_f32")


def slab_sum(part, splits, stride, dst, n, accumulate=True):
    """dst[0:n] (+)= sum of the `splits` fp32 slabs (stride floats apart) of a weight-gradient EPI_SPLIT product."""
    _check(lib().mmtg_slab_sum(_p(part), splits, stride, _p(dst), int(accumulate), n, _stream()), "slab_sum")


class WgradProblem(C.Structure):
    """mmtg_wgrad_problem of include/mmtg_hip.h."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_long), ("B", C.c_void_p), ("ldb", C.c_long),
                ("C", C.c_void_p), ("ldc", C.c_long), ("M", C.c_int), ("N", C.c_int),
                ("planeA", C.c_long), ("planeB", C.c_long)]


def wgrad_group_sizes(shapes, splits, config=0):
    """(tiles, workspace floats, counters) of a grouped launch over problems of the given (M, N) shapes."""
    tb, nw = (256, 8) if config else (128, 4)
    tiles = sum(((M + tb - 1) // tb) * ((N + tb - 1) // tb) for (M, N) in shapes)
    return tiles, tiles * splits * tb * tb, tiles * nw


def wgrad_group(problems, K, splits, ws, counters, accumulate=False, config=0):
    """Grouped weight gradients C (+)= A^T B (include/mmtg_hip.h, mmtg_wgrad_group).
    problems: list of (A [K, lda] bf16, B [K, ldb] bf16, C [M, ldc] f32, M, N[, lda, ldb, ldc]).
    config 2 (x3): A and B are Planes (hi | lo plane pairs)."""
    arr = (WgradProblem * len(problems))()
    for i, pr in enumerate(problems):
        A, B, C_, M, N = pr[:5]
        lda, ldb, ldc = (pr[5:8] if len(pr) >= 8 else (M, N, N))
        if config & 2:
            arr[i] = WgradProblem(_p(A.t), lda, _p(B.t), ldb, _p(C_), ldc, M, N, A.plane, B.plane)
        else:
            arr[i] = WgradProblem(_p(A), lda, _p(B), ldb, _p(C_), ldc, M, N, 0, 0)
    _check(lib().mmtg_wgrad_group(int(config), len(problems), C.addressof(arr), int(K), int(splits), _p(ws), 0 if ws is None else ws.numel(),
                                  _p(counters), 0 if counters is None else counters.numel(), int(accumulate), _stream()), "wgrad_group")


def transpose_batch(src, dst, desc, n, max_rows, max_cols):
    """Matrix i ([rows, cols] at src + desc[i,0]) -> [cols, rows] at dst + desc[i,3]; desc: int64 CUDA [n,4]."""
    _check(lib().mmtg_transpose_batch(dt(src), _p(src), _p(dst), _p(desc), n, max_rows, max_cols, _stream()), "transpose_batch")


# ------------------------------------------------------------------ generation
def logits_process_argmax(logits, ldl, V, generated, ldg, gen_len, temperature, rep_penalty, nxt, B):
    _check(lib().mmtg_logits_process_argmax(_p(logits), ldl, V, _p(generated), ldg, _p(gen_len),
                                            float(temperature), float(rep_penalty), _p(nxt), B, _stream()),
           "logits_process_argmax")


def logits_process_sample(logits, ldl, V, generated, ldg, gen_len, temperature, rep_penalty, top_k, top_p, uniforms, nxt, B,
                          filtered=None):
    """One draw per row from the reference's filtered distribution (include/mmtg_hip.h); uniforms: f32 [B] in [0,1)."""
    _check(lib().mmtg_logits_process_sample(_p(logits), ldl, V, _p(generated), ldg, _p(gen_len), float(temperature),
                                            float(rep_penalty), int(top_k), float(top_p), _p(uniforms), _p(nxt), _p(filtered),
                                            B, _stream()), "logits_process_sample")


# ------------------------------------------------------------------ KV-cached decode step
def decode_embed(table, seq, c, x, pos, tpw_type, tpw_mask, type_out, keep, B, P, S, E, two_sents, V, sent, max_sent_num):
    _check(lib().mmtg_decode_embed(dt(table), _p(table), _p(seq), seq.stride(0), _p(c), _p(x), _p(pos), _p(tpw_type),
                                   _p(tpw_mask), _p(type_out), _p(keep), keep.stride(0), B, P, S, E, two_sents, V, sent,
                                   max_sent_num, _stream()), "decode_embed")


def decode_embed_add(g, wpe, wte, type_ids, pos, h, B, D, stats=None):
    _check(lib().mmtg_decode_embed_add(dt(g), _p(g), _p(wpe), _p(wte), _p(type_ids), _p(pos), _p(h), B, D, _p(stats), _stream()),
           "decode_embed_add")


DG_NP = 32      # statistics partials per row (include/mmtg_hip.h, mmtg_decode_gemm)


def decode_gemm(mode, A, W, C_, M, N, K, bias=None, colsum=None, stats_in=None, np_in=0, eps=1e-5, act=EPI_NONE, out_f32=False,
                resid=None, stats_out=None, splits=1, ws=None, counters=None, lda=None, ldw=None, ldc=None, ldr=None,
                emb_pos=None, emb_type=None, type_ids=None, pos=None):
    """The fused products of the decode step (include/mmtg_hip.h, mmtg_decode_gemm)."""
    _check(lib().mmtg_decode_gemm(int(mode), M, N, K, _p(A), K if lda is None else lda, _p(W), K if ldw is None else ldw, _p(C_),
                                  N if ldc is None else ldc, _p(bias), _p(colsum), _p(stats_in), int(np_in), float(eps), int(act),
                                  int(out_f32), _p(resid), N if ldr is None else ldr, _p(stats_out), int(splits), _p(ws),
                                  0 if ws is None else ws.numel(), _p(counters), 0 if counters is None else counters.numel(),
                                  _p(emb_pos), _p(emb_type), _p(type_ids), _p(pos), _stream()), "decode_gemm")


def decode_gemm_x3(mode, A, W, M, N, K, C_=None, Cp=None, ldc=None, bias=None, colsum=None, stats_in=None, np_in=0, eps=1e-5, act=EPI_NONE,
                   resid=None, ldr=None, stats_out=None, splits=1, ws=None, counters=None, emb_pos=None, emb_type=None, type_ids=None, pos=None):
    """The fused decode products on plane pairs (include/mmtg_hip.h, mmtg_decode_gemm_x3); A, W, Cp: Planes."""
    _check(lib().mmtg_decode_gemm_x3(int(mode), M, N, K, _p(A.t), A.ld, A.plane, _p(W.t), W.ld, W.plane, _p(C_), N if ldc is None else ldc,
                                     0 if Cp is None else _p(Cp.t), 0 if Cp is None else Cp.ld, 0 if Cp is None else Cp.plane,
                                     _p(bias), _p(colsum), _p(stats_in), int(np_in), float(eps), int(act), _p(resid), N if ldr is None else ldr,
                                     _p(stats_out), int(splits), _p(ws), 0 if ws is None else ws.numel(), _p(counters),
                                     0 if counters is None else counters.numel(), _p(emb_pos), _p(emb_type), _p(type_ids), _p(pos), _stream()),
           "decode_gemm_x3")


def ln_fold_weights_x3(W, gamma, beta, bias, Wf, colsum, bias_f, N, K):
    _check(lib().mmtg_ln_fold_weights_x3(_p(W.t), W.ld, W.plane, _p(gamma), _p(beta), _p(bias), _p(Wf.t), Wf.ld, Wf.plane, _p(colsum), _p(bias_f),
                                         N, K, _stream()), "ln_fold_weights_x3")


def decode_attn_split_x3(part, splits, bias, kcache, vcache, keep, pos, out, B, nH, dh, Tmax):
    _check(lib().mmtg_decode_attn_split_x3(_p(part), splits, _p(bias), _p(kcache), _p(vcache), _p(keep), keep.stride(0), _p(pos), _p(out.t),
                                           out.plane, B, nH, dh, Tmax, _stream()), "decode_attn_split_x3")


def decode_embed_x3(table, seq, c, x, pos, tpw_type, tpw_mask, type_out, keep, B, P, S, E, two_sents, V, sent, max_sent_num):
    _check(lib().mmtg_decode_embed_x3(_p(table), _p(seq), seq.stride(0), _p(c), _p(x.t), x.plane, _p(pos), _p(tpw_type), _p(tpw_mask),
                                      _p(type_out), _p(keep), keep.stride(0), B, P, S, E, two_sents, V, sent, max_sent_num, _stream()),
           "decode_embed_x3")


def ln_fold_weights(W, gamma, beta, bias, Wf, colsum, bias_f, N, K, ldw=None):
    _check(lib().mmtg_ln_fold_weights(_p(W), K if ldw is None else ldw, _p(gamma), _p(beta), _p(bias), _p(Wf), _p(colsum), _p(bias_f),
                                      N, K, _stream()), "ln_fold_weights")


def decode_attn(qkv, kcache, vcache, keep, pos, out, B, nH, dh, Tmax):
    _check(lib().mmtg_decode_attn(dt(qkv), _p(qkv), _p(kcache), _p(vcache), _p(keep), keep.stride(0), _p(pos), _p(out),
                                  B, nH, dh, Tmax, _stream()), "decode_attn")


def decode_attn_split(part, splits, bias, kcache, vcache, keep, pos, out, B, nH, dh, Tmax):
    _check(lib().mmtg_decode_attn_split(dt(out), _p(part), splits, _p(bias), _p(kcache), _p(vcache), _p(keep), keep.stride(0),
                                        _p(pos), _p(out), B, nH, dh, Tmax, _stream()), "decode_attn_split")


def decode_select(logits, ldl, V, seq, pos, P, sent, temperature, rep_penalty, B, pos_next=None):
    _check(lib().mmtg_decode_select(_p(logits), ldl, V, _p(seq), seq.stride(0), _p(pos), P, sent, float(temperature),
                                    float(rep_penalty), B, _p(pos_next), _stream()), "decode_select")


def decode_sample(logits, ldl, V, seq, pos, P, sent, temperature, rep_penalty, top_k, top_p, uniforms, B, pos_next=None):
    _check(lib().mmtg_decode_sample(_p(logits), ldl, V, _p(seq), seq.stride(0), _p(pos), P, sent, float(temperature),
                                    float(rep_penalty), int(top_k), float(top_p), _p(uniforms), uniforms.stride(0), B, _p(pos_next),
                                    _stream()), "decode_sample")


def decode_advance(pos):
    _check(lib().mmtg_decode_advance(_p(pos), _stream()), "decode_advance")


def decode_mlp_ws_floats(M):
    return int(lib().mmtg_decode_mlp_ws_floats(int(M)))


def decode_mlp_sync_words():
    return int(lib().mmtg_decode_mlp_sync_words())


def decode_mlp_census(device):
    """Workgroup placement of a 256-workgroup, one-per-CU launch: an [8, 8] int tensor, [v, x] = workgroups with id % 8 == v found on
    physical XCD x (include/mmtg_hip.h, mmtg_decode_mlp_census).  Synchronises."""
    import torch
    out = torch.zeros(64, dtype=torch.int32, device=device)
    _check(lib().mmtg_decode_mlp_census(_p(out), _stream()), "decode_mlp_census")
    return out.view(8, 8).cpu()


def decode_mlp(X, stats_in, np_in, eps, W1f, colsum1, bias1f, W2t, bias2, G, C_, stats_out, ws, sync, M, D, plain=False, trace=None):
    """c_fc (LN-fold + GELU) -> mlp.c_proj (+ bias + residual + statistics) of a GPT-2 block in one launch (include/mmtg_hip.h, mmtg_decode_mlp)."""
    _check(lib().mmtg_decode_mlp(int(M), int(D), _p(X), X.stride(0), _p(stats_in), int(np_in), float(eps), _p(W1f), W1f.stride(0), _p(colsum1),
                                 _p(bias1f), _p(W2t), W2t.stride(0), _p(bias2), _p(G), G.stride(0), _p(C_), C_.stride(0), _p(stats_out),
                                 _p(ws), ws.numel(), _p(sync), int(bool(plain)), _p(trace), _stream()), "decode_mlp")
