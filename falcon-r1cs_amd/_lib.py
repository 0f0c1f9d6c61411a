"""ctypes binding of libfrw.so (the C ABI declared in include/frw.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C falcon-r1cs_amd/csrc``.
There is no fallback: if the shared object is missing or no HIP device is usable, callers get
an exception, never a CPU-computed result.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FrwError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__("%s failed: %d %s" % (where, code, detail))


def lib_path():
    return os.path.join(_HERE, "libfrw.so")


class LayoutStruct(C.Structure):
    _fields_ = [("logn", C.c_int32), ("n", C.c_int32), ("num_witness", C.c_int32),
                ("num_instance", C.c_int32), ("num_constraints", C.c_int32),
                ("seg_off", C.c_int32 * 8), ("seg_len", C.c_int32 * 8)]


class LayoutDualStruct(C.Structure):
    _fields_ = [("logn", C.c_int32), ("n", C.c_int32), ("num_witness", C.c_int32),
                ("num_instance", C.c_int32), ("num_constraints", C.c_int32),
                ("seg_off", C.c_int32 * 15), ("seg_len", C.c_int32 * 15)]


class CompactLayoutStruct(C.Structure):
    _fields_ = [("logn", C.c_int32), ("n", C.c_int32), ("bytes_per_signature", C.c_uint64),
                ("small_off", C.c_uint64), ("num_small", C.c_uint64), ("t_off", C.c_uint64), ("num_t", C.c_uint64),
                ("bits_off", C.c_uint64),
                ("num_bit_words", C.c_uint64), ("bit_seg_off", C.c_uint64 * 6), ("instance_off", C.c_uint64),
                ("num_instance_values", C.c_uint64), ("status_off", C.c_uint64)]


class MsmInfoStruct(C.Structure):
    _fields_ = [("num_points", C.c_uint64), ("window_bits", C.c_int32), ("num_windows", C.c_int32), ("table_bytes", C.c_uint64),
                ("workspace_bytes_per_signature", C.c_uint64)]


class Groth16PkDesc(C.Structure):
    _fields_ = [("num_instance", C.c_uint64), ("num_witness", C.c_uint64), ("domain_size", C.c_uint64),
                ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p), ("beta_g2", C.c_void_p),
                ("delta_g2", C.c_void_p), ("a_query", C.c_void_p), ("b_g1_query", C.c_void_p), ("b_g2_query", C.c_void_p),
                ("h_query", C.c_void_p), ("l_query", C.c_void_p)]


class Groth16KeyOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32)]


class Groth16PkInfoStruct(C.Structure):
    _fields_ = [("mode", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32), ("z_lo", C.c_uint64), ("z_hi", C.c_uint64),
                ("h_lo", C.c_uint64), ("h_hi", C.c_uint64), ("key_bytes", C.c_uint64)]


class R1csInfoStruct(C.Structure):
    _fields_ = [("num_statements", C.c_uint64), ("count_logn9", C.c_uint64), ("count_logn10", C.c_uint64),
                ("num_instance", C.c_uint64), ("num_witness", C.c_uint64), ("num_constraints", C.c_uint64),
                ("log_domain_size", C.c_int32), ("witness_map_on_device", C.c_int32)]


class QapInfoStruct(C.Structure):
    _fields_ = [("log_domain_size", C.c_int32), ("domain_size", C.c_uint64), ("num_constraints", C.c_uint64),
                ("num_instance", C.c_uint64), ("workspace_bytes_per_signature", C.c_uint64)]


# name -> (restype, argtypes); must list every symbol include/frw.h declares
PROTOTYPES = {
    "frw_layout": (C.c_int, [C.c_int, C.POINTER(LayoutStruct)]),
    "frw_strerror": (C.c_char_p, [C.c_int]),
    "frw_last_error": (C.c_char_p, []),
    "frw_device_count": (C.c_int, []),
    "frw_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "frw_ctx_destroy": (None, [C.c_void_p]),
    "frw_witness_ntt_verify_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_ntt_modq_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "frw_witness_ntt_verify": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "frw_ntt_modq": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    "frw_diag_host_allocations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "frw_ctx_trim": (C.c_int, [C.c_void_p]),
    "frw_diag_valu_rates": (C.c_int, [C.c_void_p, C.POINTER(C.c_double * 4)]),
    "frw_r1cs_diag_host_allocations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "frw_msm_g1_load": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "frw_msm_free": (None, [C.c_void_p]),
    "frw_g1_fixed_base": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "frw_g2_fixed_base": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "frw_msm_g2_load": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "frw_msm_g1_load_narrow": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "frw_msm_g2_load_narrow": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "frw_msm_g1_load_bare": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "frw_msm_g2_load_bare": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "frw_groth16_pk_load_opts": (C.c_int, [C.c_int, C.POINTER(Groth16PkDesc), C.POINTER(Groth16KeyOpts), C.POINTER(C.c_void_p)]),
    "frw_groth16_pk_query": (C.c_void_p, [C.c_void_p, C.c_int]),
    "frw_groth16_pk_info": (C.c_int, [C.c_void_p, C.POINTER(Groth16PkInfoStruct)]),
    "frw_groth16_setup_r1cs_opts": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Groth16KeyOpts), C.POINTER(C.c_void_p), C.c_void_p]),
    "frw_groth16_prove_partial_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "frw_groth16_prove_combine_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "frw_groth16_vk_load_opts": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "frw_diag_poly_eval_dev": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_diag_groth16_side_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "frw_msm_g2_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "frw_msm_info": (C.c_int, [C.c_void_p, C.POINTER(MsmInfoStruct)]),
    "frw_msm_g1_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "frw_groth16_msm_h_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "frw_groth16_pk_load": (C.c_int, [C.c_int, C.POINTER(Groth16PkDesc), C.POINTER(C.c_void_p)]),
    "frw_groth16_pk_free": (None, [C.c_void_p]),
    "frw_groth16_setup": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "frw_groth16_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "frw_groth16_prove_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "frw_groth16_prove_rs_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "frw_groth16_vk_load": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "frw_groth16_vk_free": (None, [C.c_void_p]),
    "frw_groth16_verify": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "frw_diag_pairing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_compact_layout": (C.c_int, [C.c_int, C.POINTER(CompactLayoutStruct)]),
    "frw_witness_ntt_verify_compact_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_expand_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_expand_host": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_diag_launch_shape": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_int32 * 4)]),
    "frw_layout_dual": (C.c_int, [C.c_int, C.POINTER(LayoutDualStruct)]),
    "frw_witness_dual_ntt_verify_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_witness_dual_ntt_verify": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "frw_r1cs_export": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_void_p]),
    "frw_r1cs_load": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "frw_r1cs_free": (None, [C.c_void_p]),
    "frw_r1cs_load_aggregate": (C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "frw_r1cs_info": (C.c_int, [C.c_void_p, C.POINTER(R1csInfoStruct)]),
    "frw_aggregate_assign_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_groth16_setup_r1cs": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "frw_r1cs_check_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_r1cs_eval_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_r1cs_eval_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_int]),
    "frw_r1cs_eval_scratch_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "frw_qap_info": (C.c_int, [C.c_void_p, C.POINTER(QapInfoStruct)]),
    "frw_qap_witness_map": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_qap_quotient_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "frw_qap_witness_map_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "frw_hash_to_point_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "frw_decode_public_keys_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "frw_decode_signatures_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_prepare_inputs": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_gadget_block_len": (C.c_int, [C.c_int]),
    "frw_gadget_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "frw_gadget": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                             C.c_void_p]),
    "frw_digest_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "frw_diag_write_stream_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "frw_synth_triples": (C.c_int, [C.c_int, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "frw_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "frw_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frw_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "frw_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frw_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "frw_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "frw_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
}


def load_library():
    """Load libfrw.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FrwError(-2, "load_library", "%s not built; run __graft_entry__.build()" % path)
        # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7.  If torch is going to
        # be used for device memory / streams / torch.distributed in this process, its runtime must be the one
        # already mapped when libfrw.so (NEEDED libamdhip64.so.7) is loaded, or torch later finds no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(code, where):
    if code != 0:
        lib = load_library()
        detail = "%s; %s" % (lib.frw_strerror(code).decode(), lib.frw_last_error().decode())
        raise FrwError(code, where, detail)
