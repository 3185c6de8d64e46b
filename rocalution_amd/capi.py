"""ctypes declaration of the C ABI in include/rocalution_amd.h (one entry per exported symbol).

The library is the product: importing it never falls back to anything else.  If
librocalution_amd.so is missing it is built (hipcc); if there is no GPU, ``ramd_init`` fails
loudly -- there is no host compute path in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librocalution_amd.so")
# RAMD_LIB: another build of the same library (the ASAN flavour of rocalution_amd/build.py)
LIB_PATH = os.environ.get("RAMD_LIB", LIB_PATH)
if os.environ.get("RAMD_LIB"):  # A/B runs of differently built libraries (tools/)
    LIB_PATH = os.environ["RAMD_LIB"]

OK, ERR_HIP, ERR_ARG, ERR_UNSUPPORTED, ERR_REFUSED, ERR_NO_DEVICE, ERR_STATE = range(7)
SOLVER_CG, SOLVER_GMRES, SOLVER_BICGSTAB = 0, 1, 2
SOLVER_FCG, SOLVER_CR, SOLVER_FGMRES, SOLVER_BICGSTABL, SOLVER_QMRCGSTAB, SOLVER_IDR = 3, 4, 5, 6, 7, 8
SOLVER_FIXEDPOINT = 9
PC_NONE, PC_JACOBI, PC_ILU0, PC_MCSGS, PC_MCGS, PC_MCILU, PC_GS, PC_SGS, PC_IC, PC_UAAMG, PC_SAAMG = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
PC_GLOBAL_UAAMG, PC_GLOBAL_SAAMG = 11, 12
F64, F32, I32 = 0, 1, 2
CSR, COO, ELL, HYB = 1, 4, 6, 7

vec_t = C.c_void_p
mat_t = C.c_void_p
i32, i64, f64, ptr = C.c_int, C.c_int64, C.c_double, C.c_void_p
pi32, pi64, pf64 = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)

# name -> (restype, [argtypes]); every function returning a status uses i32
SIGNATURES = {
    "ramd_init": (i32, [i32]),
    "ramd_stop": (i32, []),
    "ramd_is_initialized": (i32, []),
    "ramd_info": (i32, [C.c_char_p, i32]),
    "ramd_get_arch": (C.c_char_p, []),
    "ramd_last_error": (C.c_char_p, []),
    "ramd_set_last_error": (None, [C.c_char_p]),
    "ramd_device_count": (i32, [pi32]),
    "ramd_sync": (i32, []),
    "ramd_sync_default": (i32, []),
    "ramd_sync_interior": (i32, []),
    "ramd_sync_ghost": (i32, []),
    "ramd_compute_default": (i32, []),
    "ramd_compute_interior": (i32, []),
    "ramd_compute_ghost": (i32, []),
    "ramd_current_stream": (ptr, []),
    "ramd_alloc_pinned": (i32, [C.POINTER(ptr), i64]),
    "ramd_free_pinned": (i32, [ptr]),
    # vectors
    "ramd_vec_create": (i32, [i32, C.POINTER(vec_t)]),
    "ramd_vec_destroy": (i32, [vec_t]),
    "ramd_vec_allocate": (i32, [vec_t, i64]),
    "ramd_vec_allocate_apart": (i32, [vec_t, i64, vec_t]),
    "ramd_vec_placement_class": (i32, [vec_t, pi32]),
    "ramd_vec_place_apart": (i32, [vec_t, vec_t, pi32]),
    "ramd_vec_place_by_trial": (i32, [vec_t, C.c_void_p, C.c_void_p, i32, C.c_double, vec_t, pi32]),
    "ramd_placement_seconds": (i32, [C.POINTER(C.c_double), i32]),
    "ramd_placement_room": (i32, [i64, i32, pi32]),
    "ramd_vec_clear": (i32, [vec_t]),
    "ramd_vec_size": (i32, [vec_t, pi64]),
    "ramd_vec_dtype": (i32, [vec_t, pi32]),
    "ramd_vec_data": (ptr, [vec_t]),
    "ramd_vec_zeros": (i32, [vec_t]),
    "ramd_vec_ones": (i32, [vec_t]),
    "ramd_vec_set_values": (i32, [vec_t, f64]),
    "ramd_vec_copy_from_host": (i32, [vec_t, ptr]),
    "ramd_vec_copy_to_host": (i32, [vec_t, ptr]),
    "ramd_vec_copy_from": (i32, [vec_t, vec_t]),
    "ramd_vec_copy_from_offset": (i32, [vec_t, vec_t, i64, i64, i64]),
    "ramd_vec_copy_from_float": (i32, [vec_t, vec_t]),
    "ramd_vec_copy_from_double": (i32, [vec_t, vec_t]),
    "ramd_vec_copy_from_permute": (i32, [vec_t, vec_t, vec_t]),
    "ramd_vec_copy_from_permute_backward": (i32, [vec_t, vec_t, vec_t]),
    "ramd_vec_add_scale": (i32, [vec_t, vec_t, f64]),
    "ramd_vec_scale_add": (i32, [vec_t, f64, vec_t]),
    "ramd_vec_scale_add_scale_offset": (i32, [vec_t, f64, vec_t, f64, i64, i64, i64]),
    "ramd_vec_scale_add_scale": (i32, [vec_t, f64, vec_t, f64]),
    "ramd_vec_scale_add2": (i32, [vec_t, f64, vec_t, f64, vec_t, f64]),
    "ramd_vec_scale": (i32, [vec_t, f64]),
    "ramd_vec_dot": (i32, [vec_t, vec_t, pf64]),
    "ramd_vec_norm": (i32, [vec_t, pf64]),
    "ramd_vec_reduce": (i32, [vec_t, pf64]),
    "ramd_vec_asum": (i32, [vec_t, pf64]),
    "ramd_vec_amax": (i32, [vec_t, pf64, pi64]),
    "ramd_vec_pointwise_mult": (i32, [vec_t, vec_t]),
    "ramd_vec_pointwise_mult2": (i32, [vec_t, vec_t, vec_t]),
    "ramd_vec_get_index_values": (i32, [vec_t, vec_t, vec_t]),
    # matrices
    "ramd_mat_create": (i32, [i32, C.POINTER(mat_t)]),
    "ramd_mat_destroy": (i32, [mat_t]),
    "ramd_mat_clear": (i32, [mat_t]),
    "ramd_mat_info": (i32, [mat_t, pi32, pi32, pi64, pi32, pi32]),
    "ramd_mat_set_csr_from_host": (i32, [mat_t, i32, i32, i64, ptr, ptr, ptr]),
    "ramd_mat_copy_csr_to_host": (i32, [mat_t, ptr, ptr, ptr]),
    "ramd_mat_clone": (i32, [mat_t, C.POINTER(mat_t)]),
    "ramd_mat_cast": (i32, [mat_t, C.POINTER(mat_t)]),
    "ramd_mat_convert": (i32, [mat_t, i32]),
    "ramd_mat_ell_info": (i32, [mat_t, pi32, pi64]),
    "ramd_mat_copy_ell_to_host": (i32, [mat_t, ptr, ptr]),
    "ramd_mat_copy_coo_to_host": (i32, [mat_t, ptr, ptr, ptr]),
    "ramd_mat_apply": (i32, [mat_t, vec_t, vec_t]),
    "ramd_mat_pattern_info": (i32, [mat_t, pi32, pi32, pi32]),
    "ramd_mat_pattern_use": (i32, [mat_t, i32]),
    "ramd_mat_apply_add": (i32, [mat_t, vec_t, f64, vec_t]),
    "ramd_mat_extract_diag": (i32, [mat_t, vec_t]),
    "ramd_mat_extract_inv_diag": (i32, [mat_t, vec_t]),
    "ramd_mat_extract_submatrix": (i32, [mat_t, i32, i32, i32, i32, mat_t]),
    "ramd_mat_permute": (i32, [mat_t, vec_t]),
    "ramd_mat_multicoloring": (i32, [mat_t, pi32, ptr, vec_t]),
    "ramd_mat_ilu0_factorize": (i32, [mat_t]),
    "ramd_mat_ilup_factorize": (i32, [mat_t, i32, i32]),
    "ramd_mat_lu_analyse": (i32, [mat_t]),
    "ramd_mat_lu_analyse_clear": (i32, [mat_t]),
    "ramd_mat_lu_solve": (i32, [mat_t, vec_t, vec_t]),
    "ramd_tri_plan_stats": (i32, [i32, pi64]),
    "ramd_selftest_sf_div": (i32, [i64, ptr, ptr, ptr, ptr, ptr]),
    "ramd_mat_l_analyse": (i32, [mat_t, i32]),
    "ramd_mat_l_analyse_clear": (i32, [mat_t]),
    "ramd_mat_l_solve": (i32, [mat_t, vec_t, vec_t]),
    "ramd_mat_u_analyse": (i32, [mat_t, i32]),
    "ramd_mat_u_analyse_clear": (i32, [mat_t]),
    "ramd_mat_u_solve": (i32, [mat_t, vec_t, vec_t]),
    "ramd_mat_gen_poisson7": (i32, [mat_t, i32]),
    "ramd_mat_gen_laplace27": (i32, [mat_t, i32, i32, i32]),
    "ramd_mat_gen_laplace27_slab": (i32, [ptr, ptr, i32, i32, i32, i32, i32]),
    "ramd_mat_gen_poisson7_slab": (i32, [mat_t, mat_t, i32, i64, i64]),
    # fused ops / scalar records
    "ramd_mat_ic_factorize": (i32, [mat_t, vec_t]),
    "ramd_mat_ll_analyse": (i32, [mat_t]),
    "ramd_mat_ll_analyse_clear": (i32, [mat_t]),
    "ramd_mat_ll_solve": (i32, [mat_t, vec_t, vec_t, vec_t]),
    "ramd_mat_it_lu_analyse": (i32, [mat_t]),
    "ramd_mat_it_lu_analyse_clear": (i32, [mat_t]),
    "ramd_mat_it_lu_solve": (i32, [mat_t, i32, f64, i32, vec_t, vec_t]),
    "ramd_mat_it_ll_analyse": (i32, [mat_t]),
    "ramd_mat_it_ll_analyse_clear": (i32, [mat_t]),
    "ramd_mat_it_ll_solve": (i32, [mat_t, i32, f64, i32, vec_t, vec_t]),
    "ramd_mat_it_l_analyse": (i32, [mat_t, i32]),
    "ramd_mat_it_l_analyse_clear": (i32, [mat_t]),
    "ramd_mat_it_l_solve": (i32, [mat_t, i32, f64, i32, vec_t, vec_t]),
    "ramd_mat_it_u_analyse": (i32, [mat_t, i32]),
    "ramd_mat_it_u_analyse_clear": (i32, [mat_t]),
    "ramd_mat_it_u_solve": (i32, [mat_t, i32, f64, i32, vec_t, vec_t]),
    "ramd_mat_amg_pmis_aggregate": (i32, [mat_t, f64, vec_t, vec_t, vec_t]),
    "ramd_mat_amg_greedy_aggregate": (i32, [mat_t, f64, vec_t, vec_t, vec_t]),
    "ramd_mat_amg_unsmoothed_prolong": (i32, [mat_t, vec_t, vec_t, mat_t]),
    "ramd_mat_amg_smoothed_prolong": (i32, [mat_t, f64, i32, vec_t, vec_t, vec_t, mat_t]),
    "ramd_mat_merge_columns": (i32, [mat_t, mat_t, i32, mat_t]),
    "ramd_mat_amg_pmis_aggregate_global": (i32, [mat_t, f64, ptr, i32, i32, pi32, pi64, pi64, vec_t, i64, vec_t, vec_t,
                                                 vec_t, vec_t, pi64, pi64, pi64]),
    "ramd_mat_amg_prolong_global": (i32, [mat_t, i32, f64, i32, vec_t, vec_t, vec_t, i64, mat_t]),
    "ramd_mat_diag_mult": (i32, [mat_t, vec_t, i32]),
    "ramd_mat_sort": (i32, [mat_t]),
    "ramd_mat_transpose": (i32, [mat_t, mat_t]),
    "ramd_mat_matrix_add": (i32, [mat_t, mat_t, f64, f64, i32]),
    "ramd_mat_mat_mult": (i32, [mat_t, mat_t, mat_t]),
    "ramd_mat_extract_tri": (i32, [mat_t, mat_t, i32, i32]),
    "ramd_mat_scale_values": (i32, [mat_t, f64, i32]),
    "ramd_mat_add_scalar_values": (i32, [mat_t, f64, i32]),
    "ramd_mat_update_values": (i32, [mat_t, ptr]),
    "ramd_mem_info": (i32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ramd_scalars_set": (i32, [i32, f64]),
    "ramd_scalars_fetch": (i32, [pf64, i32, i32]),
    "ramd_scalars_fetch_async_begin": (i32, [i32, i32, i32]),
    "ramd_scalars_fetch_async_end": (i32, [i32, pf64, i32]),
    "ramd_fused_apply_dot": (i32, [mat_t, vec_t, vec_t, i32]),
    "ramd_fused_apply_add_dot": (i32, [mat_t, vec_t, f64, vec_t, vec_t, i32]),
    "ramd_fused_jacobi_sweep": (i32, [mat_t, vec_t, vec_t, vec_t, vec_t, f64]),
    "ramd_fused_apply_dotv": (i32, [mat_t, vec_t, vec_t, vec_t, i32]),
    "ramd_fused_bicg_r_update": (i32, [vec_t, vec_t, i32, i32]),
    "ramd_fused_bicg_xr_update": (i32, [vec_t, vec_t, vec_t, vec_t, vec_t, vec_t, vec_t, i32, i32, i32, i32, i32, i32]),
    "ramd_fused_bicg_direction": (i32, [vec_t, vec_t, vec_t, i32, i32, i32, i32]),
    "ramd_fused_cg_update": (i32, [vec_t, vec_t, vec_t, vec_t, i32, i32, i32, i32]),
    "ramd_fused_cg_direction": (i32, [vec_t, vec_t, vec_t, i32, i32, i32]),
    "ramd_mcsgs_build": (i32, [mat_t, i32, pi32, vec_t, C.POINTER(ptr)]),
    "ramd_mcsgs_apply": (i32, [ptr, vec_t, vec_t]),
    "ramd_mcsgs_info": (i32, [ptr, pi64]),
    "ramd_mcsgs_apply_kind": (i32, [ptr, i32, vec_t, vec_t]),
    "ramd_mcsgs_destroy": (i32, [ptr]),
    "ramd_scalars_eval": (i32, [ptr, i32, i32]),
    "ramd_vec_combine_s": (i32, [vec_t, i32, ptr, pi32, ptr, i32]),
    "ramd_fused_multi_dot": (i32, [C.POINTER(vec_t), i32, vec_t, i32]),
    "ramd_fused_multi_axpy": (i32, [vec_t, C.POINTER(vec_t), pf64, i32]),
    "ramd_fused_mgs_step": (i32, [vec_t, vec_t, i32, vec_t, i32]),
    "ramd_fused_mgs_block_max": (i32, []),
    "ramd_fused_mgs_block": (i32, [vec_t, C.POINTER(vec_t), i32, i32, i32, C.POINTER(vec_t), i32, i32]),
    "ramd_fused_normalize": (i32, [vec_t, i32, i32]),
    # measurement hooks
    "ramd_timer_start": (i32, []),
    "ramd_timer_stop": (i32, [pf64]),
    "ramd_prof_spmv_enable": (i32, [i32]),
    "ramd_prof_spmv_result": (i32, [pi32, pf64, pf64, pf64]),
    "ramd_prof_enable": (i32, [i32, i32]),
    "ramd_prof_result": (i32, [i32, pi32, pf64, pf64, pf64]),
    "ramd_prof_count": (i32, [i32, pi64]),
    "ramd_comm_rccl_count": (i32, [ptr, pi32]),
    # communicator
    "ramd_comm_unique_id": (i32, [C.c_char_p]),
    "ramd_comm_init_rccl": (i32, [i32, i32, C.c_char_p, C.POINTER(ptr)]),
    "ramd_comm_init_callback": (i32, [i32, i32, ptr, ptr, ptr, C.POINTER(ptr)]),
    "ramd_comm_destroy": (i32, [ptr]),
    "ramd_comm_rank": (i32, [ptr, pi32]),
    "ramd_comm_size": (i32, [ptr, pi32]),
    "ramd_comm_allgather_i64": (i32, [ptr, pi64, i32, pi64]),
    "ramd_comm_allreduce_scalars": (i32, [ptr, i32, i32]),
    "ramd_comm_halo_select": (i32, [ptr, i32, pi32, pi64, pi64, pi32]),
    "ramd_comm_halo_begin": (i32, [ptr, vec_t, vec_t, i32, pi32, pi64, pi64]),
    "ramd_comm_halo_begin_plan": (i32, [ptr, i32, vec_t, vec_t, i32, pi32, pi64, pi64]),
    "ramd_comm_halo_release": (i32, [ptr, i32, i64]),
    "ramd_comm_generation": (i32, [ptr, pi64]),
    "ramd_comm_halo_end": (i32, [ptr]),
    # solver layer
    "ramd_solver_create": (i32, [i32, i32, i32, C.POINTER(ptr)]),
    "ramd_solver_create_mixed": (i32, [i32, i32, C.POINTER(ptr)]),
    "ramd_solver_destroy": (i32, [ptr]),
    "ramd_solver_init": (i32, [ptr, f64, f64, f64, i32, i32]),
    "ramd_solver_init_inner": (i32, [ptr, f64, f64, f64, i32]),
    "ramd_solver_set_basis": (i32, [ptr, i32]),
    "ramd_solver_rebuild_numeric": (i32, [ptr]),
    "ramd_solver_set_seed": (i32, [ptr, C.c_ulonglong]),
    "ramd_solver_set_precond_params": (i32, [ptr, f64, f64, f64]),
    "ramd_solver_set_tri_solver": (i32, [ptr, i32, i32, f64, i32]),
    "ramd_solver_set_params": (i32, [ptr, f64, f64]),
    "ramd_solver_set_fused": (i32, [ptr, i32]),
    "ramd_solver_set_verbose": (i32, [ptr, i32]),
    "ramd_solver_set_precond_format": (i32, [ptr, i32]),
    "ramd_solver_set_decomposition": (i32, [ptr, i32]),
    "ramd_solver_set_fused_sweeps": (i32, [ptr, i32]),
    "ramd_mat_read_mtx": (i32, [C.c_char_p, i32, C.POINTER(mat_t)]),
    "ramd_mat_read_file": (i32, [C.c_char_p, i32, i32, C.POINTER(mat_t)]),
    "ramd_mat_write_file": (i32, [mat_t, C.c_char_p, i32]),
    "ramd_vec_read_file": (i32, [vec_t, C.c_char_p, i32]),
    "ramd_vec_write_file": (i32, [vec_t, C.c_char_p, i32]),
    "ramd_solver_build": (i32, [ptr, mat_t]),
    "ramd_solver_solve": (i32, [ptr, vec_t, vec_t]),
    "ramd_solver_precond_apply": (i32, [ptr, vec_t, vec_t]),
    "ramd_solver_result": (i32, [ptr, pi32, pi32, pf64]),
    "ramd_solver_set_time_mark": (i32, [ptr, i32]),
    "ramd_solver_seconds_since_time_mark": (i32, [ptr, pf64]),
    "ramd_solver_history": (i32, [ptr, pf64, i32, pi32]),
    "ramd_solver_num_colors": (i32, [ptr, pi32]),
    "ramd_solver_clear": (i32, [ptr]),
    # distributed driver
    "ramd_gsolver_create": (i32, [ptr, i32, i32, C.POINTER(ptr)]),
    "ramd_gsolver_create_mixed": (i32, [ptr, i32, i32, C.POINTER(ptr)]),
    "ramd_gsolver_init_inner": (i32, [ptr, f64, f64, f64, i32]),
    "ramd_gsolver_destroy": (i32, [ptr]),
    "ramd_gsolver_setup_poisson": (i32, [ptr, i32, i32, i32]),
    "ramd_gsolver_setup_laplace27": (i32, [ptr, i32, i32, i32]),
    "ramd_gsolver_setup_csr": (i32, [ptr, i64, i32, i64, ptr, ptr, ptr, i64, ptr, ptr, ptr, i32, ptr, ptr,
                                     ptr, ptr]),
    "ramd_gsolver_convert": (i32, [ptr, i32]),
    "ramd_gsolver_init": (i32, [ptr, f64, f64, f64, i32, i32]),
    "ramd_gsolver_set_basis": (i32, [ptr, i32]),
    "ramd_gsolver_set_verbose": (i32, [ptr, i32]),
    "ramd_gsolver_build": (i32, [ptr]),
    "ramd_gsolver_apply": (i32, [ptr, ptr, ptr]),
    "ramd_gsolver_amg_info": (i32, [ptr, ptr, ptr, ptr]),
    "ramd_gsolver_amg_level": (i32, [ptr, i32, pi64, pi64, pf64]),
    "ramd_gsolver_solve": (i32, [ptr, ptr, ptr]),
    "ramd_gsolver_solve_ones": (i32, [ptr]),
    "ramd_gsolver_prepare_ones": (i32, [ptr]),
    "ramd_gsolver_solve_device": (i32, [ptr]),
    "ramd_gsolver_result": (i32, [ptr, pi32, pi32, pf64]),
    "ramd_gsolver_set_time_mark": (i32, [ptr, i32]),
    "ramd_gsolver_seconds_since_time_mark": (i32, [ptr, pf64]),
    "ramd_gsolver_dot_check": (i32, [ptr, pf64]),
}

# entry points outside SURVEY.md's scope (SPAI / FSAI / Ruge-Stueben AMG / Gershgorin): only in a library built with
# RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE (include/rocalution_amd.h keeps them behind the same macro); attached when exported
OPTIONAL = {
    "ramd_mat_rs_pmis_coarsening": (i32, [mat_t, C.c_float, vec_t, vec_t]),
    "ramd_mat_rs_direct_interpolation": (i32, [mat_t, vec_t, vec_t, mat_t]),
    "ramd_mat_fsai": (i32, [mat_t, i32]),
    "ramd_mat_fsai_pattern": (i32, [mat_t, mat_t]),
    "ramd_mat_spai": (i32, [mat_t]),
    "ramd_mat_gershgorin": (i32, [mat_t, pf64, pf64]),
}

_lib = None


class RamdError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("rocalution_amd status %d: %s" % (status, msg))
        self.status = status


def load(build_if_missing=True):
    """dlopen librocalution_amd.so and attach the prototypes of SIGNATURES (missing symbols are an
    error: the header, this table and the library must agree)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise FileNotFoundError(LIB_PATH)
        from . import build as _build
        _build.build()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in OPTIONAL.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def has(name):
    """is this (optional) entry point in the library?"""
    return hasattr(load(), name)


def check(status):
    if status != OK:
        raise RamdError(status, (load().ramd_last_error() or b"").decode())
    return status
