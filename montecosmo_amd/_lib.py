"""ctypes binding of libmcpm.so (the C ABI declared in include/mcpm.h).

The HIP library is the product: there is NO CPU fallback.  Importing this module without a built
`libmcpm.so` next to it raises ImportError with the build command.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: libmcpm.so binds to the HIP runtime / rocFFT torch loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCPM_LIB") or os.path.join(_HERE, "libmcpm.so")      # MCPM_LIB: another build of the same ABI (A/B runs on one box)

OK = 0
POS_ABSOLUTE, POS_LATTICE = 0, 1
FD_INF, FD_2, FD_4 = 0, 2, 4

_f32p = C.c_void_p  # device pointers travel as integers
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/mcpm.h one to one
SIGNATURES = {
    "mcpm_plan_create": (C.c_int, [C.c_int] * 6 + [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mcpm_plan_create_slab": (C.c_int, [C.c_int] * 6 + [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mcpm_plan_slab_oob": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_plan_destroy": (C.c_int, [C.c_void_p]),
    "mcpm_last_error": (C.c_char_p, [C.c_void_p]),
    "mcpm_version": (C.c_char_p, []),
    "mcpm_plan_last_outliers": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_plan_last_bucketed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_plan_last_paint_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_plan_set_centre": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_plan_set_lattice_patch": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_plan_chained_fb": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _f32p, _f32p, C.POINTER(C.c_void_p)]),
    "mcpm_plan_set_halo": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_plan_set_paint3_fixed": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_plan_last_redo": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_fft_r2c": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int]),
    "mcpm_fft_c2r": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int]),
    "mcpm_cell_index": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mcpm_paint_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_float, C.c_int, _f32p, C.c_int]),
    "mcpm_paint_kb_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_float, C.c_int, C.c_float, _f32p, C.c_int]),
    "mcpm_read_kb_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_float, _f32p, _f32p, C.c_int64, C.c_float, _f32p]),
    "mcpm_paint3_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, _f32p, C.c_int]),
    "mcpm_read_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int, _f32p]),
    "mcpm_paint_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_float, C.c_int, _f32p, _f32p, _f32p]),
    "mcpm_read_vjp_pos_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int, _f32p, _f32p]),
    "mcpm_kspace_force_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int]),
    "mcpm_kspace_force_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mcpm_kspace_hessian_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int, C.c_int]),
    "mcpm_kspace_hessian_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mcpm_kspace_phase_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mcpm_hessian_combine_f32": (C.c_int, [C.c_void_p, _f32p, _f32p]),
    "mcpm_hessian_combine_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p]),
    "mcpm_force_meshes_f32": (C.c_int, [C.c_void_p, _f32p, _f32p]),
    "mcpm_force_meshes_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p]),
    "mcpm_bias_fields_f32": (C.c_int, [C.c_void_p, _f32p, C.c_float, C.c_float, C.c_float, _f32p]),
    "mcpm_bias_fields_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_float, C.c_float, C.c_float, _f32p, _f32p]),
    "mcpm_bias_fields_save_f32": (C.c_int, [C.c_void_p, _f32p, C.c_float, C.c_float, C.c_float, _f32p, _f32p]),
    "mcpm_bias_fields_vjp_saved_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, _f32p, _f32p, _f32p]),
    "mcpm_bias_weights_f32": (C.c_int, [C.c_void_p, C.c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, _f32p, C.c_float,
                                        C.POINTER(C.c_float), _f32p, _f32p, C.c_void_p]),
    "mcpm_bias_weights_vjp_f32": (C.c_int, [C.c_void_p, C.c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, _f32p, C.c_float,
                                            C.POINTER(C.c_float), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "mcpm_power_mult_f32": (C.c_int, [C.c_void_p, _f32p, C.c_float, C.c_float, C.c_float, C.c_double, C.c_void_p, C.c_void_p, C.c_int, _f32p]),
    "mcpm_interp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_float, _f32p]),
    "mcpm_lpt_combine_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, _f32p, _f32p]),
    "mcpm_lpt_combine_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, _f32p, _f32p, _f32p]),
    "mcpm_observe_pos_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int,
                                       C.c_void_p, C.c_int, C.c_int, _f32p]),
    "mcpm_observe_pos_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int,
                                           C.c_void_p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "mcpm_lightcone_tables_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_void_p, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_void_p]),
    "mcpm_observe_pos_tables_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int,
                                                  C.c_void_p, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "mcpm_rg2cgh_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "mcpm_rg2cgh_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "mcpm_cgh2rg_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "mcpm_cgh2rg_amp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "mcpm_chreshape_c64": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int]),
    "mcpm_chreshape_vjp_c64": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int]),
    "mcpm_slab_spec_elems": (C.c_int64, [C.c_void_p]),
    "mcpm_slab_zfwd": (C.c_int, [C.c_void_p, _f32p, C.c_int64, _f32p, C.c_int]),
    "mcpm_slab_ycol": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mcpm_slab_ycol2": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mcpm_slab_set_window": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mcpm_slab_set_chunks": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_slab_xfused": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int]),
    "mcpm_slab_zinv": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int]),
    "mcpm_pm_forces_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p]),
    "mcpm_pm_forces_spec_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p]),
    "mcpm_pm_forces_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    "mcpm_pm_forces2_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "mcpm_plan_force_meshes": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mcpm_drift_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_float, _f32p]),
    "mcpm_kick_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_float, C.c_float, _f32p]),
    "mcpm_kick_drift_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _f32p]),
    "mcpm_kick_drift_il_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _f32p]),
    "mcpm_plan_track_dmax": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcpm_slab_zinv3_il": (C.c_int, [C.c_void_p, _f32p, _f32p]),
    "mcpm_plan_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpm_plan_profile_read": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p, C.POINTER(C.c_int64)]),
    "mcpm_stage_name": (C.c_char_p, [C.c_int]),
    "mcpm_bullfrog_step_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, _f32p, _f32p, _f32p]),
    "mcpm_bullfrog_step_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, _f32p, _f32p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "mcpm_bullfrog_step_vjp_from_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "mcpm_plan_hint_next_adjoint": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "mcpm_step_adjoint_particles_il_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, _f32p, _f32p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "mcpm_step_adjoint_particles_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, _f32p, _f32p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "mcpm_lpt_accum_f32": (C.c_int, [C.c_void_p, _f32p, C.c_float, C.c_float, C.c_int, _f32p, _f32p]),
    "mcpm_lattice_scatter_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_float, _f32p]),
    "mcpm_lattice_dot_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p]),
    "mcpm_lpt_vjp_opts_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f64p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f64p]),
    "mcpm_pm_forces_vjp_opts_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]),
    "mcpm_lpt_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f64p, _f32p, _f32p, _f32p, _f64p]),
    "mcpm_lpt_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _f32p, _f32p]),
    "mcpm_lpt_save_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _f32p, _f32p]),
    "mcpm_lpt_vjp_saved_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f64p, _f32p, _f32p, _f32p, _f32p, _f64p]),
    "mcpm_nbody_bf_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f64p, _f64p, C.c_double, _f64p, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    "mcpm_nbody_ckpt_floats": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "mcpm_plan_probe_particle_pitch": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.POINTER(C.c_int64)]),
    "mcpm_plan_set_particle_pitch": (C.c_int, [C.c_void_p, C.c_int64]),
    "mcpm_plan_particle_pitch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcpm_nbody_bf_vjp_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f64p, _f64p, C.c_double, _f64p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f64p]),
    "mcpm_slab_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "mcpm_slab_comm_init_local": (C.c_int, [C.c_void_p]),
    "mcpm_slab_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcpm_slab_comm_init_ops": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcpm_slab_comm_selftest": (C.c_int, [C.c_void_p]),
    "mcpm_slab_comm_shutdown": (C.c_int, [C.c_void_p]),
    "mcpm_slab_bind_workspace": (C.c_int, [C.c_void_p] + [_f32p] * 8),
    "mcpm_slab_step_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    "mcpm_slab_step_vjp_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _f32p, _f32p,
                                        C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_double, C.c_double]),
    "mcpm_slab_dmax_seq": (C.c_int64, [C.c_void_p]),
    "mcpm_slab_dmax_read": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mcpm_selftest_store3_nt": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int]),
    "mcpm_axpby_f32": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int64, C.c_float, C.c_float, _f32p]),
    "mcpm_growth_table": (C.c_int, [C.c_double] * 6 + [C.c_int] + [_f64p] * 7),
    "mcpm_distance_table": (C.c_int, [C.c_double] * 6 + [C.c_int] + [_f64p] * 2),
}


ABI_VERSION = "mcpm 0.6 (gfx950)"   # must equal mcpm_version() of the loaded library (include/mcpm.h MCPM_ABI_VERSION)


def _load():
    build_log = ""
    if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        # a fresh checkout (the .so is git-ignored): build the HIP library in-tree once; there is still no fallback
        import subprocess
        r = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], check=False, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            build_log = "\n--- tail of the failed `make -C montecosmo_amd/csrc` ---\n" + "\n".join(r.stdout.splitlines()[-30:])
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product and there is no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C montecosmo_amd/csrc`." + build_log)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.mcpm_version().decode()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI '{got}' but this package was written against '{ABI_VERSION}': "
                          "stale library, rebuild it with `make -C montecosmo_amd/csrc`.")
    return lib


lib = _load()


class McpmError(RuntimeError):
    pass


def check(rc, plan=None, what=""):
    if rc != OK:
        msg = lib.mcpm_last_error(plan)
        raise McpmError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
