"""Loads libstark_hip.so (the HIP kernels + C ABI of include/stark_hip.h) with
ctypes.  There is no fallback: if the library is missing or has no GPU to run
on, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STARK_HIP_LIBRARY names another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("STARK_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libstark_hip.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
voidpp = C.POINTER(C.c_void_p)
szp = C.POINTER(C.c_size_t)
sz = C.c_size_t
i32 = C.c_int32

TRANSCRIPT_CB = C.CFUNCTYPE(i32, C.c_void_p, C.c_uint32, i32, u8p, u64p)

# name -> (restype, argtypes); mirrors include/stark_hip.h one to one
SIGNATURES = {
    "stark_last_error": (C.c_char_p, []),
    "stark_ctx_create": (i32, [i32, voidpp]),
    "stark_ctx_destroy": (None, [C.c_void_p]),
    "stark_ctx_set_stream": (i32, [C.c_void_p, C.c_void_p]),
    "stark_ctx_reset_stream": (i32, [C.c_void_p]),
    "stark_ctx_synchronize": (i32, [C.c_void_p]),
    "stark_ctx_trim": (i32, [C.c_void_p]),
    "stark_dev_alloc": (i32, [C.c_void_p, sz, voidpp]),
    "stark_dev_free": (i32, [C.c_void_p, C.c_void_p]),
    "stark_copy_h2d": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_copy_d2h": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_copy_d2d": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_ntt": (i32, [C.c_void_p, C.c_void_p, sz, i32]),
    "stark_ntt_dev": (i32, [C.c_void_p, C.c_void_p, sz, i32]),
    "stark_ntt_batch": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, i32]),
    "stark_ntt_batch_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, i32]),
    "stark_coset_lde_batch": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, sz, u64p, C.c_void_p, sz]),
    "stark_coset_lde_batch_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, sz, u64p, C.c_void_p, sz]),
    "stark_vec_read_dev": (i32, [C.c_void_p, C.c_void_p, sz, C.c_void_p, sz, C.c_void_p]),
    "stark_poly_compose_linear_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, u64p]),
    "stark_coset_interpolate_batch_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, u64p]),
    "stark_vec_inv_dev": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_poly_lincomb_dev": (i32, [C.c_void_p, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_mpoly_eval_dev": (i32, [C.c_void_p, sz, C.c_void_p, C.c_void_p, sz, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, sz]),
    "stark_poly_degree_dev": (i32, [C.c_void_p, C.c_void_p, sz, szp, C.POINTER(i32)]),
    "stark_poly_degree_post_dev": (i32, [C.c_void_p, C.c_void_p, sz, C.c_void_p]),
    "stark_sample_bytes_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, i32, C.c_void_p]),
    "stark_fourstep_twiddle_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, sz, i32]),
    "stark_fourstep_pack_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, sz, sz, i32, C.c_void_p]),
    "stark_fourstep_unpack_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, C.c_void_p]),
    "stark_vec_mul": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_vec_mul_dev": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz]),
    "stark_merkle_commit_fe": (i32, [C.c_void_p, C.c_void_p, sz, u8p, voidpp]),
    "stark_merkle_commit_fe_dev": (i32, [C.c_void_p, C.c_void_p, sz, u8p, voidpp]),
    "stark_merkle_roots_fe_batch_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, C.c_void_p]),
    "stark_merkle_root_fe_static_dev": (i32, [C.c_void_p, C.c_void_p, sz, C.c_void_p]),
    "stark_merkle_subtree_roots_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, C.c_void_p, voidpp]),
    "stark_merkle_root_of_digests_dev": (i32, [C.c_void_p, C.c_void_p, sz, u8p, voidpp]),
    "stark_merkle_commit_bytes": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, sz, u8p, voidpp]),
    "stark_merkle_open": (i32, [C.c_void_p, sz, u8p, szp]),
    "stark_merkle_open_many": (i32, [C.c_void_p, C.c_void_p, sz, C.c_void_p, szp]),
    "stark_tree_num_leaves": (sz, [C.c_void_p]),
    "stark_tree_download": (i32, [C.c_void_p, C.c_void_p]),
    "stark_tree_free": (None, [C.c_void_p]),
    "stark_fri_fold": (i32, [C.c_void_p, C.c_void_p, sz, u64p, u64p, u64p, C.c_void_p]),
    "stark_fri_fold_dev": (i32, [C.c_void_p, C.c_void_p, sz, u64p, u64p, u64p, C.c_void_p]),
    "stark_fri_fold_shard_dev": (i32, [C.c_void_p, C.c_void_p, sz, u64p, u64p, u64p, sz, sz, sz, C.c_void_p,
                                       C.c_void_p, voidpp]),
    "stark_fri_num_rounds": (sz, [sz, sz, sz]),
    "stark_fri_commit": (i32, [C.c_void_p, C.c_void_p, sz, u64p, u64p, sz, sz, TRANSCRIPT_CB, C.c_void_p, voidpp]),
    "stark_fri_commit_dev": (i32, [C.c_void_p, C.c_void_p, sz, u64p, u64p, sz, sz, TRANSCRIPT_CB, C.c_void_p, voidpp]),
    "stark_fri_layers_count": (sz, [C.c_void_p]),
    "stark_fri_layer_len": (sz, [C.c_void_p, sz]),
    "stark_fri_layer_root": (i32, [C.c_void_p, sz, u8p]),
    "stark_fri_layer_download": (i32, [C.c_void_p, sz, C.c_void_p]),
    "stark_fri_layer_read": (i32, [C.c_void_p, sz, u64p, sz, C.c_void_p]),
    "stark_fri_layer_open": (i32, [C.c_void_p, sz, sz, u8p, szp]),
    "stark_fri_layer_open_many": (i32, [C.c_void_p, sz, C.c_void_p, sz, C.c_void_p, szp]),
    "stark_fri_layer_query": (i32, [C.c_void_p, sz, C.c_void_p, sz, C.c_void_p, C.c_void_p, szp, szp]),
    "stark_poly_eval_points_dev": (i32, [C.c_void_p, C.c_void_p, sz, sz, sz, C.c_void_p, sz, C.c_void_p]),
    "stark_fri_query_all": (i32, [C.c_void_p, C.c_void_p, sz, C.c_void_p, C.c_void_p]),
    "stark_fri_layers_free": (None, [C.c_void_p]),
}

_lib = None


def kernel_source_stamp() -> str:
    """SHA-256 over the kernel and C-ABI sources (stark_amd/csrc/*.hip, *.hpp, in name order).  Profiles
    taken for one state of the kernels are stamped with it, and a counter figure is only quoted next
    to a measurement when the stamp still matches the sources the library was built from."""
    import hashlib
    d = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def preload_hip_runtime() -> None:
    """A process must hold ONE HIP runtime.  A PyTorch-ROCm wheel carries its own
    libamdhip64.so (and HSA runtime) and asks for it by a name that does not
    match the soname of the system copy, so if this library pulled the system
    copy in first, a later torch.cuda initialisation would start a second
    runtime and find "No HIP GPUs".  When such a wheel is installed, load its
    copy first; libstark_hip.so's dependency then resolves to it by soname."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Returns the loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C stark_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
