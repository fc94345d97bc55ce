"""ctypes binding of libsnmf_hip.so (the C ABI in include/snmf.h).

This is the Python analogue of the MEX shim in integration/sparse_nmf_mex.cpp: plain pointers
and sizes across the boundary, no torch types.  There is no CPU fallback: if the shared library
is missing the import of the compute entry points fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libsnmf_hip.so")
import glob as _glob
# The library is several translation units compiled in parallel and linked once (csrc/snmf_internal.h says which is which).
SRC = sorted(_glob.glob(os.path.join(_HERE, "csrc", "*.hip")))
# experiment kernels (csrc/experiments/: k_hstep_m, the merged-role H step -- 15 % slower than the shipped kernel, kept for its
# counters): only in builds that ask for them, SNMF_EXPERIMENTS=1 python scripts/build_variant.py exp -> -DSNMF_EXPERIMENTS
SRC_EXPERIMENTS = sorted(_glob.glob(os.path.join(_HERE, "csrc", "experiments", "*.hip")))
OBJ_DIR = os.path.join(_ROOT, "build", "obj")
# headers each translation unit includes beyond the ones every unit does (an edit to a header rebuilds only its users)
_COMMON_HDRS = ["snmf_internal.h", "snmf_kernels.h", os.path.join(_ROOT, "include", "snmf.h")]
# int (*snmf_allreduce_fn)(double* stats_dev, int64_t len, void* user)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)

_TU_HDRS = {
    "snmf_api.hip": ["snmf_frontend.h", "snmf_generic.h", "snmf_prof.h"],
    "snmf_tu_wstats.hip": ["snmf_generic.h", "snmf_wstats_dispatch.h"],
    "snmf_tu_wstats4.hip": ["snmf_generic.h", "snmf_wstats_dispatch.h"],
    "snmf_tu_wstats8.hip": ["snmf_generic.h", "snmf_wstats_dispatch.h"],
    "snmf_tu_online.hip": ["snmf_online.h"],
    "snmf_tu_multi.hip": ["snmf_multi.h"],
    "snmf_tu_dnmf.hip": ["snmf_frontend.h"],
    "snmf_tu_smallf.hip": ["snmf_smallf.h"],
    "snmf_tu_itersf.hip": ["snmf_smallf.h"],
    "snmf_tu_smallr.hip": ["snmf_smallf.h", "snmf_smallr.h"],
    "snmf_tu_hstep_m.hip": ["experiments/snmf_hstep_m.h"],
}
HDRS = sorted(_glob.glob(os.path.join(_HERE, "csrc", "*.h"))) + [os.path.join(_ROOT, "include", "snmf.h")]

# every symbol include/snmf.h declares
SYMBOLS = [
    "snmf_abi_version", "snmf_last_error", "snmf_device_count",
    "snmf_ctx_create", "snmf_ctx_set_stream", "snmf_ctx_sync", "snmf_ctx_destroy",
    "snmf_sparse_nmf_f64", "snmf_sparse_nmf_f32",
    "snmf_plan_create", "snmf_plan_destroy",
    "snmf_plan_set_v_f64", "snmf_plan_set_v_f32", "snmf_plan_set_w_f64", "snmf_plan_set_w_f32",
    "snmf_plan_set_h_f64", "snmf_plan_set_h_f32", "snmf_plan_set_sparsity_f64", "snmf_plan_set_sparsity_f32",
    "snmf_plan_init", "snmf_plan_run", "snmf_plan_stats_len",
    "snmf_plan_hstep", "snmf_plan_wstats", "snmf_plan_wapply", "snmf_plan_objstats", "snmf_plan_objapply",
    "snmf_plan_stopped", "snmf_plan_run_sharded",
    "snmf_plan_get_w_f64", "snmf_plan_get_w_f32", "snmf_plan_get_h_f64", "snmf_plan_get_h_f32",
    "snmf_plan_get_objective", "snmf_plan_solve_frames_f64", "snmf_plan_solve_frames_f32",
    "snmf_ctx_timing", "snmf_ctx_timing_get", "snmf_plan_describe",
    "snmf_stft_num_frames", "snmf_stft_features_f32", "snmf_plan_set_v_from_audio_f32", "snmf_mel_features_f32", "snmf_tf_dd_f32",
    "snmf_plan_set_mask_f64", "snmf_plan_set_mask_f32", "snmf_plan_get_v_mdi_f64", "snmf_plan_get_v_mdi_f32",
    "snmf_online_create", "snmf_online_set_mel", "snmf_online_get_mel_basis_f32", "snmf_online_process_f32", "snmf_online_get_basis_f32", "snmf_online_trace",
    "snmf_online_destroy",
    "snmf_multi_create", "snmf_multi_destroy", "snmf_multi_set_v_f64", "snmf_multi_set_v_f32", "snmf_multi_set_w_f64",
    "snmf_multi_set_w_f32", "snmf_multi_set_h_f64", "snmf_multi_set_h_f32", "snmf_multi_set_sparsity_f64",
    "snmf_multi_set_sparsity_f32", "snmf_multi_init", "snmf_multi_run", "snmf_multi_get_w_f64", "snmf_multi_get_w_f32",
    "snmf_multi_get_w_rank_f64", "snmf_multi_get_h_f64", "snmf_multi_get_h_f32", "snmf_multi_get_objective",
    "snmf_sparse_nmf_multi_f64", "snmf_sparse_nmf_multi_f32", "snmf_multi_set_exchange",
    "snmf_plan_set_h_random", "snmf_run_basis_dnmf_f64", "snmf_run_basis_dnmf_f32", "snmf_run_basis_dnmf_audio_f64",
    "snmf_run_basis_train_audio_f64", "snmf_ctx_xfer_stats", "snmf_sparse_nmf_oop_f64", "snmf_sparse_nmf_oop_f32",
    "snmf_run_basis_dnmf_multi_f64", "snmf_run_basis_dnmf_multi_f32",
    "snmf_multi_release_cache", "snmf_multi_cached_teams",
    "snmf_rccl_available", "snmf_rccl_get_unique_id", "snmf_rccl_comm_create", "snmf_rccl_comm_destroy", "snmf_plan_run_sharded_rccl",
]
ABI_VERSION = 5  # include/snmf.h: SNMF_ABI_VERSION this binding was written against
EXCHANGE_AUTO, EXCHANGE_FLAGS, EXCHANGE_EVENTS = 0, 1, 2

SNMF_OK = 0
STATUS_NAMES = {
    0: "SNMF_OK", 1: "SNMF_ERR_INVALID", 2: "SNMF_ERR_NO_INIT", 3: "SNMF_ERR_DIM", 4: "SNMF_ERR_NO_FIELD",
    5: "SNMF_ERR_NO_DEVICE", 6: "SNMF_ERR_NOMEM", 7: "SNMF_ERR_STATE", 8: "SNMF_ERR_UNSUPPORTED", 9: "SNMF_ERR_INTERNAL",
}


class SnmfParams(C.Structure):
    _fields_ = [
        ("F", C.c_int32), ("T", C.c_int32), ("r", C.c_int32),
        ("beta", C.c_double), ("max_iter", C.c_int32), ("conv_eps", C.c_double),
        ("cost_check", C.c_int32), ("floor_v", C.c_int32),
        ("sparsity_kind", C.c_int32), ("sparsity_scalar", C.c_double),
        ("w_update_ind", C.c_void_p), ("h_update_ind", C.c_void_p),
    ]


class SnmfStftParams(C.Structure):
    _fields_ = [
        ("framelength", C.c_int32), ("frameshift", C.c_int32), ("fftlength", C.c_int32), ("dcbin", C.c_int32),
        ("splice", C.c_int32), ("preemph", C.c_double), ("pow", C.c_double), ("nonzerofloor", C.c_double),
        ("window", C.c_void_p),
    ]


class SnmfOnlineParams(C.Structure):
    _fields_ = [
        ("fftlength", C.c_int32), ("framelength", C.c_int32), ("frameshift", C.c_int32),
        ("dcbin", C.c_int32), ("dcbin_back", C.c_int32), ("delay", C.c_int32),
        ("preemph", C.c_double), ("pow", C.c_double), ("nonzerofloor", C.c_double), ("overlapscale", C.c_double),
        ("R_x", C.c_int32), ("R_d", C.c_int32),
        ("beta_div", C.c_double), ("sparsity", C.c_double), ("max_iter", C.c_int32), ("cost_check", C.c_int32),
        ("conv_eps", C.c_double),
        ("enhance_method", C.c_int32), ("init_N_len", C.c_int32),
        ("alpha_eta", C.c_double), ("alpha_d", C.c_double), ("beta", C.c_double), ("beta_max", C.c_double),
        ("blk_sparse", C.c_int32), ("P_len_k", C.c_int32), ("P_len_l", C.c_int32), ("blk_gap", C.c_int32),
        ("alpha_p", C.c_double),
        ("adapt_train_N", C.c_int32), ("R_a", C.c_int32), ("m_a", C.c_int32),
        ("overlap_m_a", C.c_double), ("Ar_up", C.c_double),
        ("class_outputs", C.c_int32), ("basis_update_N", C.c_int32), ("basis_update_E", C.c_int32),
    ]


class SnmfOnlineFrame(C.Structure):
    _fields_ = [
        ("n_iter", C.c_int32), ("trig", C.c_int32), ("solved", C.c_int32), ("n_up", C.c_int32),
        ("adapt_iters", C.c_int32),
        ("beta", C.c_float), ("A_x_mag", C.c_float), ("A_d_mag", C.c_float), ("Q_control", C.c_float),
    ]


class SnmfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


def _tu_deps(src):
    hs = _COMMON_HDRS + _TU_HDRS.get(os.path.basename(src), [])
    return [src] + [h if os.path.isabs(h) else os.path.join(_HERE, "csrc", h) for h in hs]


def build(force=False, verbose=False, jobs=None, extra_flags=(), lib_path=None, obj_dir=None):
    """Compile libsnmf_hip.so for gfx950 with hipcc (cross-compiles without a GPU): every csrc/*.hip to an object file
    (in parallel, only the ones whose sources changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    lib_path = lib_path or LIB_PATH
    obj_dir = obj_dir or OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value",
             "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc")] + list(extra_flags)
    experiments = os.environ.get("SNMF_EXPERIMENTS", "0") == "1" or "-DSNMF_EXPERIMENTS" in flags
    if experiments and "-DSNMF_EXPERIMENTS" not in flags:
        flags.append("-DSNMF_EXPERIMENTS")
    todo, objs = [], []
    for src in SRC + (SRC_EXPERIMENTS if experiments else []):
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        deps = [d for d in _tu_deps(src) if os.path.exists(d)]
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps):
            todo.append((src, obj))
    if not todo and os.path.exists(lib_path) and os.path.getmtime(lib_path) >= max(os.path.getmtime(o) for o in objs):
        return lib_path

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    jobs = jobs or max(1, min(len(todo), os.cpu_count() or 1, 8))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", lib_path] + objs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return lib_path


_lib = None


def _preload_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  The PyTorch wheel ships its own libamdhip64.so.7
    (same soname as /opt/rocm's); whichever is loaded first wins for the whole process, and torch
    cannot initialise its devices on top of the other copy.  torch.distributed (RCCL) is the
    multi-GPU transport of this package, so when the wheel is present its copy is loaded first --
    without importing torch (slow).  Without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared library (building nothing).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SNMF_LIB_PATH") or LIB_PATH  # SNMF_LIB_PATH: a diagnostic build (scripts/phase_prof.sh)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The engine has no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    PP = C.POINTER(SnmfParams)
    sig = {
        "snmf_abi_version": (C.c_int, []),
        "snmf_last_error": (C.c_char_p, []),
        "snmf_device_count": (C.c_int, []),
        "snmf_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "snmf_ctx_set_stream": (C.c_int, [vp, vp]),
        "snmf_ctx_sync": (C.c_int, [vp]),
        "snmf_ctx_destroy": (None, [vp]),
        "snmf_sparse_nmf_f64": (C.c_int, [vp, PP, vp, i64, vp, vp, vp, vp, vp, C.POINTER(i32)]),
        "snmf_sparse_nmf_f32": (C.c_int, [vp, PP, vp, i64, vp, vp, vp, vp, vp, C.POINTER(i32)]),
        "snmf_plan_create": (C.c_int, [vp, PP, C.POINTER(vp)]),
        "snmf_plan_destroy": (None, [vp]),
        "snmf_plan_init": (C.c_int, [vp]),
        "snmf_plan_run": (C.c_int, [vp, i32, C.POINTER(i32)]),
        "snmf_plan_stats_len": (i64, [vp]),
        "snmf_plan_hstep": (C.c_int, [vp]),
        "snmf_plan_wstats": (C.c_int, [vp, vp]),
        "snmf_plan_wapply": (C.c_int, [vp, vp]),
        "snmf_plan_objstats": (C.c_int, [vp, vp]),
        "snmf_plan_objapply": (C.c_int, [vp, vp]),
        "snmf_plan_stopped": (C.c_int, [vp, C.POINTER(i32)]),
        "snmf_plan_get_objective": (C.c_int, [vp, vp, vp, C.POINTER(i32)]),
        "snmf_ctx_timing": (C.c_int, [vp, C.c_int]),
        "snmf_ctx_timing_get": (C.c_int, [vp, C.c_char_p, C.POINTER(dbl), C.POINTER(i64)]),
        "snmf_plan_describe": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    }
    SP = C.POINTER(SnmfStftParams)
    sig["snmf_stft_num_frames"] = (i64, [SP, i64])
    sig["snmf_stft_features_f32"] = (C.c_int, [vp, SP, vp, i64, C.c_int, vp, i64, C.c_int, C.POINTER(i32)])
    sig["snmf_plan_set_v_from_audio_f32"] = (C.c_int, [vp, SP, vp, i64, C.c_int])
    sig["snmf_mel_features_f32"] = (C.c_int, [vp, vp, i32, i32, i32, vp, i64, i32, vp, i64, C.c_int])
    sig["snmf_tf_dd_f32"] = (C.c_int, [vp, dbl, i32, i32, vp, i64, vp, i64, C.c_int])
    OP = C.POINTER(SnmfOnlineParams)
    sig["snmf_online_create"] = (C.c_int, [vp, OP, vp, vp, vp, vp, vp, vp, C.POINTER(vp)])
    sig["snmf_online_set_mel"] = (C.c_int, [vp, i32, i32, vp, vp, vp])
    sig["snmf_online_get_mel_basis_f32"] = (C.c_int, [vp, vp, i64])
    sig["snmf_online_process_f32"] = (C.c_int, [vp, vp, i64, C.c_int, vp, vp, vp, vp, i64, C.POINTER(i64)])
    sig["snmf_online_get_basis_f32"] = (C.c_int, [vp, vp, i64])
    sig["snmf_online_trace"] = (C.c_int, [vp, vp, i64, C.POINTER(i64)])
    sig["snmf_online_destroy"] = (None, [vp])
    for ty in ("f64", "f32"):
        sig[f"snmf_plan_solve_frames_{ty}"] = (C.c_int, [vp, i32, vp, i64, i32, vp, vp, vp, vp])
    for nm in ("v", "w", "h", "mask"):
        for ty in ("f64", "f32"):
            sig[f"snmf_plan_set_{nm}_{ty}"] = (C.c_int, [vp, vp, i64, C.c_int])
    for ty in ("f64", "f32"):
        sig[f"snmf_plan_get_v_mdi_{ty}"] = (C.c_int, [vp, vp, i64, C.c_int])
    for nm in ("w", "h"):
        for ty in ("f64", "f32"):
            sig[f"snmf_plan_get_{nm}_{ty}"] = (C.c_int, [vp, vp, i64, C.c_int])
    for ty in ("f64", "f32"):
        sig[f"snmf_plan_set_sparsity_{ty}"] = (C.c_int, [vp, vp, C.c_int])
    sig["snmf_multi_create"] = (C.c_int, [vp, i32, PP, vp, C.POINTER(vp)])
    sig["snmf_multi_destroy"] = (None, [vp])
    sig["snmf_multi_release_cache"] = (i32, [])
    sig["snmf_multi_cached_teams"] = (i32, [])
    for nm in ("v", "w", "h"):
        for ty in ("f64", "f32"):
            sig[f"snmf_multi_set_{nm}_{ty}"] = (C.c_int, [vp, vp, i64])
    for ty in ("f64", "f32"):
        sig[f"snmf_multi_set_sparsity_{ty}"] = (C.c_int, [vp, vp])
        sig[f"snmf_multi_get_w_{ty}"] = (C.c_int, [vp, vp, i64])
        sig[f"snmf_multi_get_h_{ty}"] = (C.c_int, [vp, vp, i64])
        sig[f"snmf_sparse_nmf_multi_{ty}"] = (C.c_int, [vp, i32, PP, vp, i64, vp, vp, vp, vp, vp, C.POINTER(i32)])
    sig["snmf_multi_init"] = (C.c_int, [vp])
    sig["snmf_multi_set_exchange"] = (C.c_int, [vp, i32])
    sig["snmf_multi_run"] = (C.c_int, [vp, i32, C.POINTER(i32)])
    sig["snmf_multi_get_w_rank_f64"] = (C.c_int, [vp, i32, vp, i64])
    sig["snmf_multi_get_objective"] = (C.c_int, [vp, vp, vp, C.POINTER(i32)])
    u64 = C.c_uint64
    sig["snmf_plan_set_h_random"] = (C.c_int, [vp, u64])
    for ty in ("f64", "f32"):
        sig[f"snmf_run_basis_dnmf_{ty}"] = (C.c_int, [vp, PP, i32, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, u64, vp, i64, vp, i64, vp])
        sig[f"snmf_run_basis_dnmf_multi_{ty}"] = (C.c_int, [vp, i32, PP, i32, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, u64, vp, i64, vp, i64, vp])
    sig["snmf_run_basis_dnmf_audio_f64"] = (C.c_int, [vp, PP, SP, i32, i32, vp, i64, vp, i64, vp, i32, vp, i64, vp, u64, vp, i64, vp, i64, vp])
    sig["snmf_run_basis_train_audio_f64"] = (C.c_int, [vp, PP, SP, dbl, vp, i32, vp, i64, vp, i32, vp, u64, vp, vp, vp, vp, vp])
    sig["snmf_ctx_xfer_stats"] = (C.c_int, [vp, vp, C.c_int])
    sig["snmf_plan_run_sharded"] = (C.c_int, [vp, i32, vp, ALLREDUCE_FN, vp, i32, i32, C.POINTER(i32)])
    sig["snmf_rccl_available"] = (C.c_int, [])
    sig["snmf_rccl_get_unique_id"] = (C.c_int, [vp, i64])
    sig["snmf_rccl_comm_create"] = (C.c_int, [i32, vp, i32, i32, C.POINTER(vp)])
    sig["snmf_rccl_comm_destroy"] = (None, [vp])
    sig["snmf_plan_run_sharded_rccl"] = (C.c_int, [vp, i32, vp, vp, i32, i32, C.POINTER(i32)])
    for ty in ("f64", "f32"):
        sig[f"snmf_sparse_nmf_oop_{ty}"] = (C.c_int, [vp, PP, vp, i64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)])
    lib.snmf_abi_version.restype = C.c_int
    if lib.snmf_abi_version() != ABI_VERSION:  # a stale library must not be driven through newer prototypes
        raise ImportError(f"{path} has ABI version {lib.snmf_abi_version()}, this binding needs {ABI_VERSION}: rebuild the library")
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != SNMF_OK:
        msg = load().snmf_last_error()
        raise SnmfError(status, msg.decode() if msg else "")
